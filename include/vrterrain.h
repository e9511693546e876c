/*
 * vrterrain.h — C ABI of libvrterrain.so: the MI355X-native (gfx950) terrain +
 * deferred-shading hot path of Viictor/vrenderer.
 *
 * Every entry point names the reference interface it replaces (file:line relative
 * to the reference tree).  Conventions:
 *   - POD structs, opaque handles, `int` status (0 = VR_OK), no exceptions cross
 *     the boundary;
 *   - host memory is caller-owned, device memory is library-owned unless an entry
 *     point says "device pointer";
 *   - all GPU work is stream-ordered on the context's stream (vr_context_set_stream
 *     takes a caller hipStream_t); nothing synchronises unless it returns host data;
 *   - one context per device, not thread-safe per context (the reference drives the
 *     path from one thread and one command list, Renderer.cpp:321-454).
 *
 * Matrices are row-major float4x4 used with ROW vectors (v' = v * M), exactly the
 * layout of Donut's PlanarViewConstants consumed by terrain_vs.hlsl:60-61.
 */
#ifndef VRTERRAIN_H
#define VRTERRAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VR_API __attribute__((visibility("default")))

/* ---- status codes -------------------------------------------------------- */
enum {
    VR_OK = 0,
    VR_ERR_INVALID_ARGUMENT = 1,
    VR_ERR_NO_DEVICE = 2,        /* no HIP device / HIP runtime error            */
    VR_ERR_OUT_OF_MEMORY = 3,
    VR_ERR_TOO_MANY_INSTANCES = 4, /* select produced > max_instances nodes
                                      (reference: assert, TerrainPass.cpp:238)   */
    VR_ERR_OVERFLOW = 5,         /* an internal work list overflowed             */
    VR_ERR_HIP = 6
};

/* ---- compile-time settings of the reference, as runtime fields ------------ */
/* TerrainSettings (TerrainPass.h:23-30), QuadTree::MAX_LODS (QuadTree.h:67),
 * minLodDistance (QuadTree.cpp:236), morph start (terrain_vs.hlsl:20). */
#define VR_MAX_LODS 12

typedef struct vr_terrain_params {
    int32_t max_instances;    /* MAX_INSTANCES = 4096                          */
    float   surface_size;     /* SURFACE_SIZE  = 2048 (quadtree width=height)  */
    float   world_size;       /* WORLD_SIZE    = 2048                          */
    int32_t grid_size;        /* GRID_SIZE     = 32 (must be 32 in this build) */
    float   min_lod_distance; /* 4.0                                           */
    float   morph_start;      /* 0.85                                          */
    float   location[3];      /* quadtree centre (TerrainPass.cpp:108)         */
    int32_t reserved;
} vr_terrain_params;

/* What the path reads from donut::engine::IView / PlanarViewConstants
 * (terrain_vs.hlsl:46,60-61; TerrainPass.cpp:181,280-282,301-302). */
typedef struct vr_view {
    float   world_to_view[16];
    float   view_to_clip[16];
    float   world_to_clip[16];
    float   clip_to_world[16];
    float   camera_pos[4];      /* matViewToWorld[3] = IView::GetViewOrigin()     */
    float   planes[6][4];       /* dm::frustum: outward normal xyz, distance w;
                                   order NEAR, FAR, LEFT, RIGHT, TOP, BOTTOM;
                                   a point is outside when dot(n,p) - w > 0      */
    int32_t viewport_x, viewport_y, viewport_w, viewport_h;
    int32_t mirrored;           /* IView::IsMirrored() -> frontCounterClockwise   */
    int32_t reverse_depth;      /* IView::IsReverseDepth() (must be 0 here)       */
    int32_t reserved[2];
} vr_view;

/* Donut InstanceData as filled by TerrainPass::UpdateTransforms
 * (TerrainPass.cpp:243-253): 16-byte header + two float3x4 (row-major rows). */
typedef struct vr_instance {
    uint32_t padding;
    uint32_t first_geometry_instance_index;
    uint32_t first_geometry_index;
    uint32_t num_geometries;
    float    transform[12];
    float    prev_transform[12];
} vr_instance;

/* TerrainPass::RenderParams (TerrainPass.h:68-73) + EditorParams::m_MaxHeight
 * (Renderer.h:40, read at TerrainPass.cpp:155). */
typedef struct vr_render_params {
    int32_t wireframe;    /* EditorParams::m_Wireframe -> RasterFillMode::Wireframe (TerrainPass.cpp:476): triangle edges as aliased lines */
    int32_t lock_view;    /* reuse the previous selection (TerrainPass.cpp:173,191)  */
    int32_t depth_only;   /* PS = null (TerrainPass.cpp:465)                         */
    int32_t assume_cleared; /* extension: 1 = caller guarantees the G-buffer holds its
                               clear values, so the pass need not read depth back and
                               writes clear values itself (fuses RenderTargets::Clear,
                               Renderer.cpp:382)                                     */
    float   max_height;   /* EditorParams::m_MaxHeight = 400                         */
    int32_t depth_ranges; /* extension: 1 = leave the depth range of every 32x32 light tile with the G-buffer
                             (needs assume_cleared, shaded fill mode); vr_deferred_light_tiled's culling
                             stage then takes them instead of reading the depth plane again.  Any other
                             write to the G-buffer in between (clear, upload, another render) drops them,
                             and a G-buffer whose pointers were handed out (vr_gbuffer_describe) never
                             gets them; results are identical either way                      */
    int32_t reserved[2];
} vr_render_params;

/* Donut LightConstants subset consumed by the deferred pass (ShadeSurface): directional lights
 * (direction, irradiance, angular size), point and spot lights (position, 1/range, optional source
 * radius; spot: cone axis = direction, inner/outer half angles in radians). */
enum { VR_LIGHT_DIRECTIONAL = 1, VR_LIGHT_SPOT = 2, VR_LIGHT_POINT = 3 };
typedef struct vr_light {
    float   direction[3];  int32_t type;
    float   position[3];   float   radius;            /* source radius (0 = punctual) */
    float   color[3];      float   intensity;         /* irradiance (directional)     */
    float   angular_size_or_inv_range;                /* radians | 1/range            */
    float   inner_angle, outer_angle;
    float   out_of_bounds_shadow;
} vr_light;

/* G-buffer planes (Donut GBufferRenderTargets formats; 28 B/pixel, row-major):
 *   depth    float32                 (cleared to 1.0)
 *   diffuse  SRGBA8_UNORM   (u32)    rgb albedo, a opacity
 *   specular SRGBA8_UNORM   (u32)    rgb F0,     a occlusion
 *   normals  RGBA16_SNORM   (2xu32)  xyz normal, w roughness
 *   emissive RGBA16_FLOAT   (2xu32)  rgb emissive
 * HDR colour: RGBA16_FLOAT (8 B/pixel, Renderer.h:69-79). */
typedef struct vr_gbuffer_desc {
    int32_t width, height;
    void*   depth;     /* device pointers */
    void*   diffuse;
    void*   specular;
    void*   normals;
    void*   emissive;
} vr_gbuffer_desc;

/* Screen-tile partition of one frame over the GPUs of a node. */
#define VR_OWNER_TILE 128
typedef struct vr_partition {
    int32_t rank, world_size;   /* owner(tx,ty) = (tx + ty) mod world_size; 1 <= world_size <= 64 */
} vr_partition;

typedef struct vr_context  vr_context;
typedef struct vr_terrain  vr_terrain;
typedef struct vr_gbuffer  vr_gbuffer;
typedef struct vr_image    vr_image;    /* RGBA16F image in device memory        */

/* ---- context --------------------------------------------------------------- */
/* nvrhi::IDevice + the single command list (main.cpp:57-61, Renderer.cpp:48). */
VR_API int  vr_context_create(int device_ordinal, vr_context** out);
VR_API void vr_context_destroy(vr_context* ctx);
VR_API int  vr_context_set_stream(vr_context* ctx, void* hip_stream);
VR_API int  vr_context_synchronize(vr_context* ctx);          /* Renderer::Submit + wait */
/* Options.  VR_OPT_ASYNC_GEOMETRY (default 1): vr_terrain_render runs its view-dependent geometry
 * stages (select, vertex, setup, bins) on a second stream, so that they overlap whatever the caller
 * queued on the context's stream after the previous vr_terrain_render (typically vr_deferred_light
 * of the previous frame); the tile pass stays on the context's stream.  0 = everything on one stream. */
/* VR_OPT_DISPATCH_EVENTS (default 1): the two big kernels of a frame (tile pass, lighting pass) are launched with
 * hipExtLaunchKernelGGL, whose start/stop events are stamped by the dispatch itself, and the stop events double as the
 * cross-stream dependencies - no event-record packets sit between the two kernels.  0 = explicit hipEventRecord. */
/* VR_OPT_RASTER_TILE (default 0): edge of the tile pass's raster tiles - 0 = chosen by frame size and split (32 pixels below
 * ~13000 64-pixel tiles per rank, i.e. up to about 9.6K x 5.4K; 64 above), 32 or 64 = pinned.  The rendered frame does not depend on it. */
/* VR_OPT_PLANE_TRACKING (default 1): main_ps writes 0 to the emissive target for every pixel (terrain_ps.hlsl:80) and Clear writes 0
 * everywhere, so on this path the emissive plane - 8 of the G-buffer's 28 B/pixel - only ever holds zeros.  While the library knows
 * the plane is all zero (it cleared it - vr_gbuffer_create / vr_gbuffer_clear - or a tile pass that writes every pixel ran since the
 * last foreign write) the tile pass does not rewrite it; the plane's contents are the same either way.  vr_gbuffer_upload of the
 * plane ends that knowledge until the next clear or whole-target pass; vr_gbuffer_describe ends it for good (the pointers have left
 * the library), and likewise the depth ranges of vr_render_params::depth_ranges.  0 = every pass writes all five planes. */
/* VR_OPT_SCRATCH_WORST_CASE (default 0): a terrain's per-frame scratch (vertices, triangle records, bins: three rotating sets) is sized
 * by the high-water mark of the node counts its frames select - 1024 nodes to begin with (1.3 GB for the three sets), doubled before a
 * frame can exceed it - instead of for max_instances (4096: 5 GB).  1 (set BEFORE vr_terrain_create): worst case up front, no growth,
 * no frame can ever be truncated by the scratch. */
enum { VR_OPT_ASYNC_GEOMETRY = 1, VR_OPT_DISPATCH_EVENTS = 2, VR_OPT_RASTER_TILE = 3, VR_OPT_PLANE_TRACKING = 4, VR_OPT_SCRATCH_WORST_CASE = 5 };
VR_API int  vr_context_set_option(vr_context* ctx, int option, int value);
VR_API const char* vr_last_error(void);
VR_API const char* vr_version(void);
/* 0 for a product build.  Non-zero: the library was built with timing-experiment or profiling switches
 * (csrc/vr_experiments.h: -DVR_EXPERIMENT_BUILD + VR_EXP_* / VR_*_PROFILE) and may render wrong images on purpose. */
VR_API uint32_t vr_build_experiments(void);

/* Per-kernel timing with HIP events recorded on the context's stream, the analogue of
 * the reference's PROFILE_GPU_SCOPE timestamp queries (Profiler.h:55-125,
 * Renderer.cpp:326-437).  Kernel ids: */
enum { VR_K_SELECT = 0, VR_K_VERTEX, VR_K_SETUP, VR_K_CLIP, VR_K_SCAN, VR_K_FILL, VR_K_RASTER,
       VR_K_DEFERRED, VR_K_DETILE, VR_K_CLEAR, VR_K_DEFERRED_TILED, VR_K_NODE_HEIGHTS,
       VR_K_TM_HISTOGRAM, VR_K_TM_EXPOSURE, VR_K_TONEMAP, VR_K_DETILE_LDR, VR_K_RASTER_DEPTH, VR_K_LIGHT_CULL, VR_K_RASTER_LIT, VR_K_COUNT };
/* (VR_K_COUNT grows when kernels are added - 16 in round 2, 18 in round 3, 19 now: size vr_timing_collect's arrays with the
 * constant of the header the host is compiled against AND check vr_timing_kernel_count() at run time) */
VR_API int  vr_timing_kernel_count(void);
VR_API int  vr_timing_enable(vr_context* ctx, int enable);     /* also resets the samples; 1 = every kernel (two event records per
                                                                * launch), 2 = only the tile pass and the lighting passes, whose
                                                                * events the dispatch stamps at no host cost */
/* Synchronises the stream; per kernel id: summed milliseconds and launch count since
 * the last enable/collect; resets the samples. */
VR_API int  vr_timing_collect(vr_context* ctx, float ms_sum[VR_K_COUNT], int32_t launches[VR_K_COUNT]);
VR_API const char* vr_kernel_name(int id);

/* ---- host helper: what FirstPersonCamera::LookAt + perspProjD3DStyle +
 * PlanarView::UpdateCache produce (Renderer.cpp:97,312-319).  Pure host code. -- */
VR_API int vr_view_from_camera(const float eye[3], const float target[3], const float up[3],
                               float vertical_fov_radians, float z_near, float z_far,
                               int32_t width, int32_t height, vr_view* out);
VR_API void vr_terrain_default_params(vr_terrain_params* out);
VR_API void vr_render_default_params(vr_render_params* out);

/* ---- terrain --------------------------------------------------------------- */
/* TerrainPass::Init + QuadTree::QuadTree/Init (TerrainPass.cpp:34-141,
 * QuadTree.cpp:10-52).  height_r8: hm_w*hm_h bytes; albedo_srgba8: al_w*al_h*4
 * bytes (decoded with sRGB=true, Renderer.cpp:55).  Both are copied to the device
 * and mip chains are generated there (Donut TextureCache mip generation).
 * Limits: each texture at most 16384 texels on a side and 2^26 texels in all (8192 x 8192): the tile pass
 * reads decoded copies (16 B per texel / footprint, x 4/3 for the mips) through 32-bit offsets.  Device memory
 * held by a terrain: vr_terrain_memory_bytes(). */
VR_API int  vr_terrain_create(vr_context* ctx, const vr_terrain_params* params,
                              const uint8_t* height_r8, int32_t hm_w, int32_t hm_h,
                              const uint8_t* albedo_srgba8, int32_t al_w, int32_t al_h,
                              vr_terrain** out);
VR_API void vr_terrain_destroy(vr_terrain* t);
VR_API int  vr_terrain_num_lods(const vr_terrain* t);                 /* QuadTree::GetNumLods  */
VR_API int  vr_terrain_lod_ranges(const vr_terrain* t, float out[VR_MAX_LODS]); /* GetLodRanges */
/* test/IO helper: read back one level of the device mip chain (which: 0 heightmap R8,
 * 1 albedo SRGBA8); *w,*h receive the level size; host may be NULL to query sizes. */
VR_API int  vr_terrain_download_mip(vr_terrain* t, int which, int level, void* host, size_t bytes,
                                    int32_t* w, int32_t* h, int32_t* levels);

/* QuadTree::SetHeight for every node (QuadTree.cpp:164-208) as a device reduction over the heightmap,
 * and m_HeightLoaded (QuadTree.h:70).  The reference wrote this and left its launch commented out
 * (QuadTree.cpp:46-51), so the default is "not loaded": cull boxes span y in [0, camera.y].  With
 * enable != 0 the per-node (position.y, extents.y) are computed and NodeSelect culls with
 * [min.y, max.y] * max_height (QuadTree.cpp:87-91); UpdateTransforms then carries them too.
 * enable == 0 returns to the tree as built (y = location.y, extents.y = 0, not loaded). */
VR_API int  vr_terrain_update_heights(vr_terrain* t, int enable);
/* test helper: (position.y, extents.y) of node ids [first, first+count) */
VR_API int  vr_terrain_download_node_heights(vr_terrain* t, uint32_t first, uint32_t count, float* out);

/* QuadTree::ClearSelectedNodes + NodeSelect + TerrainPass::UpdateTransforms +
 * EditorParams::m_NumChunks (TerrainPass.cpp:173-198, QuadTree.cpp:80-131).
 * Runs on the device; optional host outputs (may be NULL) force a stream sync.
 * node_ids: id = (4^d-1)/3 + iz*2^d + ix, in the reference's m_SelectedNodes order. */
VR_API int  vr_terrain_select(vr_terrain* t, const vr_view* view, float max_height,
                              uint32_t* node_ids, vr_instance* instances, uint32_t* count);

/* TerrainPass::Render into the G-buffer framebuffer (TerrainPass.cpp:143-232;
 * terrain_vs.hlsl, terrain_ps.hlsl; raster state TerrainPass.cpp:460-485).
 * `part` may be NULL (= whole frame on this device).
 * The call is asynchronous.  Conditions only the device can detect - more than max_instances nodes selected
 * (VR_ERR_TOO_MANY_INSTANCES, the reference's assert at TerrainPass.cpp:238), a full bin / clipper work list or a frame that
 * outgrew the scratch (VR_ERR_OVERFLOW: triangles or nodes were dropped) - are STICKY: every geometry chain leaves its counters
 * in pinned host memory, and the next vr_terrain_render / vr_terrain_prepare after such a chain has completed returns the code
 * once (vr_last_error() names the frame's node count).  The call that returns it has still queued its own frame.
 * vr_terrain_num_chunks() waits for the current frame's geometry and returns its condition directly. */
VR_API int  vr_terrain_render(vr_terrain* t, const vr_view* view, const vr_view* view_prev,
                              vr_gbuffer* gb, const vr_render_params* rp,
                              const vr_partition* part);
/* Opt-in fused variant: TerrainPass::Render + DeferredLightingPass::Render in ONE pass over the pixels.  The tile
 * pass's resolve encodes each pixel to the G-buffer's formats in registers, decodes and shades it with the lighting pass's own
 * arithmetic and writes depth + hdr_out only (the other four G-buffer planes keep what they held): 12 bytes per pixel reach memory
 * instead of 28 + 36.  depth and hdr_out (HdrColor, or this rank's packed RGB16F tiles with a partition) are bit-identical to
 * vr_terrain_render(assume_cleared = 1) + vr_deferred_light on the same inputs.  Needs render->assume_cleared = 1, shaded fill
 * mode, <= 16 lights.  The fused kernel covers the reference's case (heightmap and albedo of one size, power-of-two world size,
 * directional / punctual point lights, no shadow term); for anything else the call runs the two passes one after the other.
 * vr_terrain_prepare works as for vr_terrain_render.  The unfused pair stays the default path (and the measured one). */
VR_API int  vr_terrain_render_lit(vr_terrain* t, const vr_view* view, vr_gbuffer* gb, const vr_render_params* rp,
                                  const vr_partition* part, const vr_light* lights, int32_t num_lights,
                                  const float ambient_top[3], const float ambient_bottom[3], vr_image* hdr_out);
/* Optional: build the view-dependent geometry (select .. bins) of an upcoming vr_terrain_render ahead of
 * time, on one of the terrain's two geometry streams.  Called right after vr_terrain_render of frame N with frame
 * N+1's view, it overlaps frame N's tile pass (and leaves its lighting pass alone).  Up to two frames may be prepared
 * ahead (N+1 and N+2: three geometry sets rotate, two chains are in flight - what keeps small frames and a rank's share
 * of a split frame from waiting for the latency-bound chain); naming a frame that is already prepared is a no-op, a
 * third distinct frame replaces the oldest.  A later vr_terrain_render uses a prepared set when view, max_height,
 * target size and partition are identical; sets that are never used are simply overwritten.
 * Extension; the reference has no counterpart (its frames are strictly serial).
 * Note for hosts with many streams of their own: HIP maps streams onto 4 hardware queues by default and streams that
 * share a queue serialise; the library uses the context's stream + 2.  GPU_MAX_HW_QUEUES raises the limit. */
VR_API int  vr_terrain_prepare(vr_terrain* t, const vr_view* view, vr_gbuffer* gb, const vr_render_params* rp,
                               const vr_partition* part);
/* EditorParams::m_NumChunks of the last render/select (syncs the stream). */
VR_API int  vr_terrain_num_chunks(vr_terrain* t, uint32_t* count);

/* ---- render targets ---------------------------------------------------------- */
/* RenderTargets::Init / Clear (Renderer.h:60-101, Renderer.cpp:382).
 * Under VR_OPT_PLANE_TRACKING (default) the clear is lazy: the next vr_terrain_render that shades the whole frame runs as
 * "over a cleared target" (what vr_render_params::assume_cleared asks for explicitly) and writes each pixel once; anything else
 * that looks at the planes first - a lighting pass, a rank's share, a depth-only pass, vr_gbuffer_download / _upload /
 * _describe - has the clear values written then.  What the planes read as never differs from an eager clear. */
VR_API int  vr_gbuffer_create(vr_context* ctx, int32_t width, int32_t height, vr_gbuffer** out);
VR_API void vr_gbuffer_destroy(vr_gbuffer* gb);
VR_API int  vr_gbuffer_clear(vr_gbuffer* gb);
/* The device pointers of the planes, for interop.  From this call on the library assumes nothing about the planes' contents
 * (a host may write through the pointers at any time): vr_render_params::depth_ranges is ignored and every tile pass writes
 * all five planes (VR_OPT_PLANE_TRACKING) for the rest of this G-buffer's life. */
VR_API int  vr_gbuffer_describe(vr_gbuffer* gb, vr_gbuffer_desc* out);
/* 1 when the library knows `plane` (4 = emissive; others: 0) holds only zeros and the next tile pass will not rewrite it. */
VR_API int  vr_gbuffer_plane_known_zero(vr_gbuffer* gb, int plane);
/* Plane-state tracking per region (VR_OPT_PLANE_TRACKING; a region = 8 rows x 32 pixels, what one wave of a 32-pixel raster
 * tile resolves): counts[0] regions nothing is known about, counts[1] regions whose specular plane holds the terrain
 * shader's one constant in every pixel (terrain_ps.hlsl:76: the next tile pass that covers such a region completely does not
 * rewrite that plane), counts[2] regions that hold the clear values in all planes (a tile pass over a cleared target that
 * draws nothing there writes nothing), counts[3] their total.  The planes' contents never depend on the tracking; this is
 * what a bench needs to say how many bytes a tile pass really wrote.  Waits for the context's stream. */
VR_API int  vr_gbuffer_region_census(vr_gbuffer* gb, uint32_t counts[4]);
/* test/IO helpers: copy planes host<->device (synchronous). plane: 0 depth,
 * 1 diffuse, 2 specular, 3 normals, 4 emissive. */
VR_API int  vr_gbuffer_download(vr_gbuffer* gb, int plane, void* host, size_t bytes);
VR_API int  vr_gbuffer_upload(vr_gbuffer* gb, int plane, const void* host, size_t bytes);

/* HdrColor (Renderer.h:69-79).  external_device_mem may be NULL (library allocates)
 * or a caller-owned device buffer of width*height*8 bytes. */
VR_API int  vr_image_create(vr_context* ctx, int32_t width, int32_t height,
                            void* external_device_mem, vr_image** out);
VR_API void vr_image_destroy(vr_image* img);
VR_API void* vr_image_device_ptr(vr_image* img);
VR_API int  vr_image_download(vr_image* img, void* host, size_t bytes);
VR_API int  vr_image_upload(vr_image* img, const void* host, size_t bytes);

/* ---- deferred lighting --------------------------------------------------------- */
/* DeferredLightingPass::Render(cmd, view, Inputs{GBuffer, ambientColorTop/Bottom,
 * lights, output}) (Renderer.cpp:417-428).  `part` NULL = whole frame, output
 * row-major; otherwise only owned tiles are shaded and written to `hdr_out` as a
 * packed tile-major buffer (vr_partition_packed_bytes). */
VR_API int  vr_deferred_light(vr_context* ctx, const vr_view* view, vr_gbuffer* gb,
                              const vr_light* lights, int32_t num_lights,
                              const float ambient_top[3], const float ambient_bottom[3],
                              vr_image* hdr_out, const vr_partition* part);

/* The same pass for many lights (BASELINE config 5: 1024 point lights): a culling kernel tests the
 * lights against the world-space bounds of every 128x128 macro tile and of its sixteen 32x32 light
 * tiles (frustum cells over the tiles' depth ranges) and leaves one list per light tile; the shading
 * kernel walks only that list.  Same inputs/outputs as vr_deferred_light; up to 65536 lights in all and
 * VR_TILE_LIGHT_CAP lights per tile after culling.  A tile that keeps more drops the excess (in
 * light order) and raises a device-side flag; the launch stays asynchronous, so the condition is
 * reported by vr_deferred_tiled_status.  The light array is uploaded only when it differs from the
 * one the previous call left on the device.  Device memory: (w/32) x (h/32) lists of
 * min(num_lights, VR_TILE_LIGHT_CAP) + 1 words, kept by the context. */
#define VR_TILE_LIGHT_CAP 1024
VR_API int  vr_deferred_light_tiled(vr_context* ctx, const vr_view* view, vr_gbuffer* gb,
                                    const vr_light* lights, int32_t num_lights,
                                    const float ambient_top[3], const float ambient_bottom[3],
                                    vr_image* hdr_out, const vr_partition* part);
/* Waits for the tiled passes queued so far on this context and returns VR_ERR_OVERFLOW if any of their
 * tiles kept more than VR_TILE_LIGHT_CAP lights since the last call (the flag is cleared), else VR_OK. */
VR_API int  vr_deferred_tiled_status(vr_context* ctx);

/* ---- terrain shadows ---------------------------------------- */
/* The reference renders the terrain depth-only from the sun into a 2048^2 one-cascade shadow map every
 * frame and the deferred pass consumes it (Renderer.cpp:83-93, 333-367, 427).  CascadedShadowMap and the
 * shadow lookup are Donut code (absent): [DONUT-RECOLLECTION] "stable" cascade = bounding sphere of the
 * camera frustum slice [0, maxShadowDistance], centre snapped to shadow texels in light space,
 * orthographic D3D projection; lookup = 4x4 GatherCmp tent PCF (weights [1-f,1,1,f]^2 / 9), LessEqual. */
typedef struct vr_shadow_params {
    int32_t resolution;            /* 2048 (Renderer.cpp:83) */
    float   max_shadow_distance;   /* WORLD_SIZE (:349) */
    float   light_space_z_up;      /* WORLD_SIZE (:350-352) */
    float   light_space_z_down;    /* WORLD_SIZE */
    float   depth_bias;            /* in shadow-map depth units, subtracted from the receiver; 0: the terrain
                                      depth pass has no raster bias (TerrainPass.cpp:467-471) */
    int32_t reserved[3];
} vr_shadow_params;
VR_API void vr_shadow_default_params(vr_shadow_params* out, float world_size);
/* CascadedShadowMap::SetupForPlanarViewStable(light, projectionFrustum, inverseViewMatrix, maxShadowDistance,
 * zUp, zDown, exponent, 0, 1) for the single cascade (Renderer.cpp:345-352): the light's view, to be
 * rendered with vr_terrain_render(depth_only = 1) into a resolution^2 vr_gbuffer (m_ShadowFramebuffer). */
VR_API int  vr_shadow_view_setup(const vr_light* light, const vr_view* camera_view, const vr_shadow_params* p,
                                 vr_view* out_light_view);
/* what DirectionalLight::shadowMap hands to the lighting pass */
typedef struct vr_shadow_binding {
    const vr_view* light_view;
    vr_gbuffer*    shadow_map;     /* its depth plane is sampled */
    int32_t        light_index;    /* which entry of `lights` casts it */
    float          depth_bias;
} vr_shadow_binding;
/* vr_deferred_light with the shadow term on one light (vr_light.out_of_bounds_shadow outside the map). */
VR_API int  vr_deferred_light_shadowed(vr_context* ctx, const vr_view* view, vr_gbuffer* gb,
                                       const vr_light* lights, int32_t num_lights,
                                       const float ambient_top[3], const float ambient_bottom[3],
                                       vr_image* hdr_out, const vr_partition* part,
                                       const vr_shadow_binding* shadow);

/* LdrColor: SRGBA8_UNORM render target (Renderer.h:81-92), width*height*4 bytes of device memory, or - as a
 * multi-GPU send buffer - the packed RGB8 tiles of one rank (capacity_bytes >= vr_partition_packed_bytes_ldr).
 * capacity_bytes 0 = width*height*4.  external != NULL wraps caller-owned device memory of that capacity. */
typedef struct vr_ldr_image vr_ldr_image;
VR_API int    vr_ldr_image_create(vr_context* ctx, int32_t width, int32_t height, size_t capacity_bytes, void* external_device_ptr,
                                  vr_ldr_image** out);
VR_API void   vr_ldr_image_destroy(vr_ldr_image* im);
VR_API void*  vr_ldr_image_device_ptr(vr_ldr_image* im);
VR_API size_t vr_ldr_image_capacity(vr_ldr_image* im);
VR_API int    vr_ldr_image_download(vr_ldr_image* im, void* host, size_t bytes);   /* synchronous; bytes <= capacity */
VR_API int    vr_ldr_image_upload(vr_ldr_image* im, const void* host, size_t bytes);

/* ---- multi-GPU frame assembly (new) -------------------------------- */
VR_API int    vr_partition_num_tiles(int32_t width, int32_t height, const vr_partition* part,
                                     int32_t* tiles_x, int32_t* tiles_y, int32_t* owned,
                                     int32_t* max_owned);
VR_API size_t vr_partition_packed_bytes(int32_t width, int32_t height, int32_t world_size);
/* Builds the partition tables of a context up front (they are otherwise built on first use by
 * vr_terrain_render / vr_deferred_light); needed when vr_frame_detile runs on its own context/stream. */
VR_API int    vr_partition_prepare(vr_context* ctx, int32_t width, int32_t height, const vr_partition* part);
/* After the all-gather: gathered = world_size consecutive packed buffers (device pointer; packed
 * tiles are RGB16F, 6 B/pixel - HdrColor's alpha is always 0 on this path and is not exchanged);
 * rebuilds the row-major RGBA16F frame. */
VR_API int    vr_frame_detile(vr_context* ctx, const void* gathered_device, int32_t world_size,
                              vr_image* frame_out);

/* The exchange itself (vr_frame_allgather).  `nccl_comm` is the caller's ncclComm_t for the ranks of the
 * split (one process per GPU; rank = vr_partition.rank, nranks = world_size).  ncclAllGather of this rank's packed
 * tiles (vr_partition_packed_bytes of RGB16F out of vr_deferred_light(part)) into `gathered_device`
 * (world_size x that), queued on the context's stream, followed by vr_frame_detile into frame_out - every rank ends
 * with the whole row-major RGBA16F frame.  RCCL is resolved at first use from the copy already in the process (the
 * one that made the communicator), else from librccl.so; libvrterrain.so itself does not link it. */
VR_API int    vr_frame_allgather(vr_context* ctx, void* nccl_comm, const void* packed_device, void* gathered_device,
                                 int32_t world_size, vr_image* frame_out);

/* The all-gather alone, for a host that de-tiles on another stream (vr_frame_detile[_ldr] on a second context, so that the
 * de-tile of frame i runs under the all-gather of frame i+1): ncclAllGather of bytes_per_rank bytes (vr_partition_packed_bytes
 * or vr_partition_packed_bytes_ldr) from packed_device into gathered_device (world_size x that) on the context's stream. */
VR_API int    vr_frame_allgather_tiles(vr_context* ctx, void* nccl_comm, const void* packed_device, void* gathered_device,
                                       int32_t world_size, size_t bytes_per_rank);

/* ---- tone mapping to LdrColor ------------------------------- */
/* donut::render::ToneMappingPass as used by the reference: created with default CreateParameters
 * (Renderer.cpp:256-257), AdvanceFrame(seconds) (:188-189), SimpleRender(cmd, ToneMappingParameters(),
 * view, HdrColor) into LdrColor SRGBA8 (:430-431, Renderer.h:81-95).  [DONUT-RECOLLECTION: luminance
 * histogram (256 bins over log2 luminance, 6-bit fixed-point weights split over two bins) -> average
 * log luminance between two percentiles -> eye adaptation -> extended Reinhard on luminance.]
 * Integer results (histogram, SRGBA8 pixels) are bit-exact against the oracle. */
typedef struct vr_tonemap_params {          /* ToneMappingParameters defaults + CreateParameters' log range */
    float histogram_low_percentile;         /* 0.8  */
    float histogram_high_percentile;        /* 0.95 */
    float eye_adaptation_speed_up;          /* 1.0  */
    float eye_adaptation_speed_down;        /* 0.5  */
    float min_adapted_luminance;            /* 0.02 */
    float max_adapted_luminance;            /* 0.5  */
    float exposure_bias;                    /* -0.5 */
    float white_point;                      /* 3.0  */
    float min_log_luminance;                /* -10  */
    float max_log_luminance;                /*  4   */
} vr_tonemap_params;
#define VR_TONEMAP_BINS 256
typedef struct vr_tonemap vr_tonemap;       /* the pass object: histogram + adapted-luminance (exposure) buffer */
VR_API void vr_tonemap_default_params(vr_tonemap_params* out);
VR_API int  vr_tonemap_create(vr_context* ctx, vr_tonemap** out);
VR_API void vr_tonemap_destroy(vr_tonemap* tm);
/* ResetExposure(cmd, initialExposure): adapted luminance := value (0 = "unset": the next ComputeExposure jumps to its target) */
VR_API int  vr_tonemap_reset_exposure(vr_tonemap* tm, float adapted_luminance);
VR_API int  vr_tonemap_reset_histogram(vr_tonemap* tm);
/* AddFrameToHistogram: `hdr` is a row-major RGBA16F frame (part NULL) or the packed RGB16F tiles that
 * vr_deferred_light wrote for `part` (only this rank's pixels are counted). */
VR_API int  vr_tonemap_add_frame_to_histogram(vr_tonemap* tm, const vr_tonemap_params* p, vr_image* hdr,
                                              int32_t width, int32_t height, const vr_partition* part);
/* Device pointer to the VR_TONEMAP_BINS uint32 bins: with a partition, sum it over the ranks
 * (ncclAllReduce, ncclUint32, ncclSum) between add_frame_to_histogram and compute_exposure. */
VR_API void* vr_tonemap_histogram_device_ptr(vr_tonemap* tm);
VR_API int  vr_tonemap_compute_exposure(vr_tonemap* tm, const vr_tonemap_params* p, float frame_time_seconds);
/* Render: HdrColor -> LdrColor.  part NULL: row-major SRGBA8 (width*height*4 bytes at ldr_device).
 * Otherwise packed tiles in, packed RGB8 tiles out (3 B/pixel, alpha is always 255 and is not
 * exchanged; vr_partition_packed_bytes_ldr); rebuild the frame with vr_frame_detile_ldr. */
VR_API int  vr_tonemap_render(vr_tonemap* tm, const vr_tonemap_params* p, vr_image* hdr, int32_t width, int32_t height,
                              void* ldr_device, size_t ldr_capacity_bytes, const vr_partition* part);
/* SimpleRender = ResetHistogram + AddFrameToHistogram + ComputeExposure + Render on one GPU. */
VR_API int  vr_tonemap_simple_render(vr_tonemap* tm, const vr_tonemap_params* p, float frame_time_seconds, vr_image* hdr,
                                     void* ldr_device, size_t ldr_capacity_bytes);
/* test/IO helpers (synchronous) */
VR_API int  vr_tonemap_download(vr_tonemap* tm, uint32_t histogram[VR_TONEMAP_BINS], float* adapted_luminance);
VR_API size_t vr_partition_packed_bytes_ldr(int32_t width, int32_t height, int32_t world_size);
/* gathered = world_size consecutive packed RGB8 buffers -> row-major SRGBA8 frame (device pointers) */
VR_API int  vr_frame_detile_ldr(vr_context* ctx, const void* gathered_device, int32_t world_size,
                                int32_t width, int32_t height, void* ldr_frame_device);

/* The same exchange for tone-mapped frames (3 B/pixel on the wire): ncclAllGather of the packed RGB8 tiles out of
 * vr_tonemap_render(part) + vr_frame_detile_ldr; and the tone mapper's one real exchange step, the sum of the 256
 * histogram bins over the ranks (ncclAllReduce, in place, on the tone mapper's context stream) between
 * vr_tonemap_add_frame_to_histogram(part) and vr_tonemap_compute_exposure. */
VR_API int  vr_frame_allgather_ldr(vr_context* ctx, void* nccl_comm, const void* packed_ldr_device, void* gathered_device,
                                   int32_t world_size, int32_t width, int32_t height, void* ldr_frame_device);
VR_API int  vr_tonemap_allreduce_histogram(vr_tonemap* tm, void* nccl_comm);

/* ---- a whole frame in one call ---------------------------------------------------- */
/* The terrain part of Renderer::RecordCommand (Renderer.cpp:321-446: ONE command list per frame) as one entry point:
 * vr_terrain_render(view) [RenderTargets::Clear fused when render->assume_cleared] -> vr_terrain_prepare for up to two upcoming
 * frames -> vr_deferred_light / _tiled / _shadowed -> optionally ToneMappingPass::SimpleRender on the tone mapper's own context
 * (reset + add_frame_to_histogram [+ vr_tonemap_allreduce_histogram] + compute_exposure + render) [+ vr_frame_allgather_ldr].
 * It queues what those calls queue, in that order, on the same streams; it exists because a frame is otherwise about nine calls
 * and a rank of an 8-way split of the 8K frame has a period near 0.1 ms.  When the tone mapper's context uses another stream than
 * the terrain's, the library orders the two: the stage waits for the lighting pass, and a later lighting pass into the same
 * hdr_out waits for the stage that still reads it (rotate two images to keep both streams busy). */
typedef struct vr_frame_desc {
    const vr_view*          view;               /* this frame's view (IView of Renderer::RenderScene) */
    const vr_view*          prepare_views[2];   /* geometry of upcoming frames to build ahead; NULL = none */
    const vr_render_params* render;
    const vr_partition*     part;               /* NULL = whole frame */
    const vr_light*         lights;
    int32_t                 num_lights;
    int32_t                 tiled;              /* 1 = vr_deferred_light_tiled (many lights) */
    float                   ambient_top[3], ambient_bottom[3];
    const vr_shadow_binding* shadow;            /* optional: vr_deferred_light_shadowed */
    vr_image*               hdr_out;            /* HdrColor, or this rank's packed RGB16F tiles */
    /* optional tone-map stage: tonemap NULL = none */
    struct vr_tonemap*      tonemap;
    const vr_tonemap_params* tonemap_params;
    float                   frame_time_seconds;
    int32_t                 reserved0;
    void*                   ldr_out;            /* device: LdrColor SRGBA8, or this rank's packed RGB8 tiles */
    size_t                  ldr_capacity;
    /* optional exchange behind it (N ranks; all NULL = none): the host's ncclComm_t, world x packed bytes, the whole SRGBA8 frame */
    void*                   nccl_comm;
    void*                   gathered;
    void*                   ldr_frame;
} vr_frame_desc;
VR_API int vr_frame_submit(vr_terrain* t, vr_gbuffer* gb, const vr_frame_desc* frame);

/* ---- synthetic inputs (media/ is absent from the reference checkout;
 * SURVEY §8d): seeded integer-hash fBm heightmap and banded albedo, generated on
 * the device and copied to host buffers. ------------------------------------------ */
VR_API int vr_synth_heightmap(vr_context* ctx, int32_t size, uint32_t seed, uint8_t* out_r8);
VR_API int vr_synth_albedo(vr_context* ctx, int32_t size, uint32_t seed,
                           const uint8_t* height_r8, uint8_t* out_srgba8);

/* diagnostics of the last vr_terrain_render (synchronises): out[0] selected nodes, [1] status flags,
 * [2] clipper sub-triangles, [3] clipper vertices, [4] triangles sent to the clipper, [5] bin entries,
 * [6] largest bin, [7] non-empty bins */
VR_API int vr_debug_render_stats(vr_terrain* t, uint32_t out[8]);
/* Test helper: the launch order of the last vr_terrain_render's tile pass - the raster tiles the frame (or this rank) covers,
 * longest bins first in eight classes (k_scan) - and each of those tiles' bin lengths, in that order.  `capacity` entries per
 * array; *out_count = number of tiles (0 if nothing was rendered).  Synchronises. */
VR_API int vr_debug_tile_order(vr_terrain* t, int32_t* out_tiles, uint32_t* out_bin_lengths, int32_t capacity, int32_t* out_count);
/* Device memory a terrain holds, in bytes: out[0] textures (chains + decoded tables), out[1] per-frame geometry scratch
 * (three rotating sets: instances, vertices, triangle records, bins - sized for params->max_instances), out[2] node
 * heights (after vr_terrain_update_heights), out[3] the sum. */
VR_API int vr_terrain_memory_bytes(const vr_terrain* t, uint64_t out[4]);
/* Test helper: the vertex stage's output (main_vs, terrain_vs.hlsl:35-62) of the last vr_terrain_render for `count`
 * vertices starting at vertex `first` of the instanced draw (vertex = instance * 1089 + row * 33 + column, rows = z):
 * six floats per vertex - o_position.xyzw (clip space) and o_vtx.pos.xz (world space).  Synchronises. */
VR_API int vr_debug_download_vertices(vr_terrain* t, uint32_t first, uint32_t count, float* out_xyzw_wxwz);
/* test helper: the device's linear -> sRGB8 render-target conversion applied to n host floats */
VR_API int vr_debug_srgb_encode(vr_context* ctx, const float* in, size_t n, uint8_t* out);
/* Test helper: sweeps every float with an exponent in [-60, 60) through the pixel shader's short reciprocal and square-root
 * sequences and counts differences from 1.0f / x and sqrtf(x): out[0] reciprocal, out[1] square root, out[2] values swept. */
VR_API int vr_debug_fastmath_check(vr_context* ctx, unsigned long long out[3]);

#ifdef __cplusplus
}
#endif
#endif /* VRTERRAIN_H */
