// Timing experiments and profiling switches of the kernels, in ONE place.
//
// The product kernels contain no "#ifdef VR_EXP_*": they test the constexpr flags below, which are all false in a product
// build, so the compiler drops the experiment branches.  A switch that changes the rendered image (every VR_EXP_* does: they
// isolate what one part of the tile pass costs by leaving it out) or adds instrumentation (VR_RASTER_PROFILE,
// VR_SELECT_PROFILE) can only be turned on together with -DVR_EXPERIMENT_BUILD; alone it is a compile error, so that a
// stray -D can never ship a wrong G-buffer.  vr_build_experiments() (vr_host.hip) returns kExpMask: tests/test_abi_cpu.py
// asserts 0 for the library it loads.  tools/build_variant.py adds -DVR_EXPERIMENT_BUILD for experiment variants.
#pragma once

#if !defined(VR_EXPERIMENT_BUILD)
#if defined(VR_EXP_FAST_SAMEADDR) || defined(VR_EXP_FAST_ALB1) || defined(VR_EXP_FAST_HGT1) || defined(VR_EXP_FAST_NOLEVEL1) || \
    defined(VR_EXP_TILED_STORES) || defined(VR_EXP_NOSTORE) || defined(VR_EXP_NOEMISSIVE) || defined(VR_EXP_NORECORD) || \
    defined(VR_EXP_NOENCODE) || defined(VR_EXP_NOTABLES) || defined(VR_EXP_REC_ON_CHANGE) || defined(VR_EXP_REC_MASKED) || defined(VR_RASTER_PROFILE) || defined(VR_SELECT_PROFILE)
#error "VR_EXP_* / VR_*_PROFILE switches produce wrong images or instrumented kernels: they need -DVR_EXPERIMENT_BUILD (tools/build_variant.py)"
#endif
#endif

#ifdef VR_EXP_FAST_SAMEADDR     // every texel fetch of the fast variant inside the tables' first 256 bytes: what does the memory system cost?
constexpr bool kExpSameAddr = true;
#else
constexpr bool kExpSameAddr = false;
#endif
#ifdef VR_EXP_FAST_ALB1         // one albedo fetch per level instead of four
constexpr bool kExpOneAlbedo = true;
#else
constexpr bool kExpOneAlbedo = false;
#endif
#ifdef VR_EXP_FAST_HGT1         // one height fetch and one albedo fetch per level
constexpr bool kExpOneHeight = true;
#else
constexpr bool kExpOneHeight = false;
#endif
#ifdef VR_EXP_FAST_NOLEVEL1     // no coarser mip level
constexpr bool kExpNoLevel1 = true;
#else
constexpr bool kExpNoLevel1 = false;
#endif
#ifdef VR_EXP_TILED_STORES      // every tile's pixels contiguous in each plane (scrambled image)
constexpr bool kExpTiledStores = true;
#else
constexpr bool kExpTiledStores = false;
#endif
#ifdef VR_EXP_NOSTORE           // nothing leaves the tile pass (a dependent dummy keeps the shading alive)
constexpr bool kExpNoStore = true;
#else
constexpr bool kExpNoStore = false;
#endif
#ifdef VR_EXP_NOEMISSIVE        // the emissive plane is not written whatever its state (what the clean-plane tracking can gain at most)
constexpr bool kExpNoEmissive = true;
#else
constexpr bool kExpNoEmissive = false;
#endif
#ifdef VR_EXP_NORECORD          // the resolve fetches no triangle records (planes of a constant)
constexpr bool kExpNoRecord = true;
#else
constexpr bool kExpNoRecord = false;
#endif
#ifdef VR_EXP_NOENCODE          // no sRGB-encode look-ups
constexpr bool kExpNoEncode = true;
#else
constexpr bool kExpNoEncode = false;
#endif
#ifdef VR_EXP_NOTABLES          // the tile pass's small tables (sRGB encode, level table) are not fetched from memory per workgroup (zeros instead)
constexpr bool kExpNoTables = true;
#else
constexpr bool kExpNoTables = false;
#endif
#ifdef VR_EXP_REC_ON_CHANGE      // the resolve's record fetch goes out of range (zeros, no memory access) for a lane whose next pixel has the same triangle
constexpr bool kExpRecOnChange = true;
#else
constexpr bool kExpRecOnChange = false;
#endif
#ifdef VR_EXP_REC_MASKED         // ... or is not issued at all for such a lane (exec-masked fetch)
constexpr bool kExpRecMasked = true;
#else
constexpr bool kExpRecMasked = false;
#endif
#ifdef VR_RASTER_PROFILE
constexpr bool kExpRasterProfile = true;
#else
constexpr bool kExpRasterProfile = false;
#endif
#ifdef VR_SELECT_PROFILE
constexpr bool kExpSelectProfile = true;
#else
constexpr bool kExpSelectProfile = false;
#endif

constexpr unsigned kExpMask = (kExpSameAddr ? 1u : 0u) | (kExpOneAlbedo ? 2u : 0u) | (kExpOneHeight ? 4u : 0u) | (kExpNoLevel1 ? 8u : 0u)
                            | (kExpTiledStores ? 16u : 0u) | (kExpNoStore ? 32u : 0u) | (kExpNoEmissive ? 64u : 0u) | (kExpNoRecord ? 128u : 0u)
                            | (kExpNoEncode ? 256u : 0u) | (kExpNoTables ? 512u : 0u) | (kExpRecOnChange ? 1024u : 0u) | (kExpRecMasked ? 2048u : 0u) | (kExpRasterProfile ? 0x10000u : 0u) | (kExpSelectProfile ? 0x20000u : 0u)
#ifdef VR_EXPERIMENT_BUILD
                            | 0x80000000u
#endif
    ;
