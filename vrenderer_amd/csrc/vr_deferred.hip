// Deferred lighting: G-buffer (28 B/pixel) -> HdrColor RGBA16F (8 B/pixel).
//
// Replaces donut::render::DeferredLightingPass::Render as called at
// Renderer.cpp:417-428 (Inputs: G-buffer, ambientColorTop/Bottom, scene lights,
// output = HdrColor).  Donut's source is absent from the reference checkout; the
// shading model restates its deferred_lighting_cs / ShadeSurface /
// GGX_AnalyticalLights_times_NdotL semantics.
//
// This is the bandwidth kernel of the path: 36 algorithmic bytes per pixel, no
// reuse.  Each lane owns 4 horizontally adjacent pixels so that every plane is read
// with one or two 16-byte loads per lane (1 KiB per wave instruction) and the
// output leaves as two 16-byte stores.  All seven loads are issued before any
// arithmetic so a wave keeps 112 B/lane in flight.  The sRGB decode table lives in
// LDS.  fp32 math; checked against the CPU oracle to per-channel RMS <= 1e-4.
#include "vr_internal.h"
#include "vr_tex_dev.h"

#include <math.h>
#include <string.h>

#include "vr_deferred_dev.h"

// Output of 4 pixels (o[2k] = r|g<<16, o[2k+1] = b, alpha 0).  Row-major frames are RGBA16F (8 B/px);
// the packed tile buffer that goes through the all-gather drops the always-zero alpha: RGB16F, 6 B/px,
// 25 % less xGMI traffic, restored by k_detile.
// NT (row-major frames only): streaming stores, with streaming loads in the kernel.  An 8K frame's lighting pass moves 1.2 GB
// through the caches and leaves 265 MB of dirty lines; the NEXT frame's tile pass then finds its texel tables evicted.
// Measured (tools/exp_serial.py, profiles/r03_serial_cache_experiment.txt): tile pass 373-379 us behind a lighting pass with
// plain loads and stores, 325-332 us behind one that streams both (= its time with no lighting pass in between); streaming
// only the loads or only the stores changes nothing; the lighting pass itself goes from 199-205 to 207 us.
template <bool PACKED, bool NT = false>
__device__ __forceinline__ void store_quad(uint2* __restrict__ out, size_t out_index, const uint32_t o[8])
{
    if (PACKED) {
        uint2* dst = reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + out_index * 3);   // 24 B per quad, 8-B aligned
        dst[0] = make_uint2(o[0], (o[1] & 0xffffu) | (o[2] << 16));
        dst[1] = make_uint2((o[2] >> 16) | (o[3] << 16), o[4]);
        dst[2] = make_uint2((o[5] & 0xffffu) | (o[6] << 16), (o[6] >> 16) | (o[7] << 16));
    } else {
        uint4* dst = reinterpret_cast<uint4*>(out + out_index);
        if (NT) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v0 = { o[0], o[1], o[2], o[3] }, v1 = { o[4], o[5], o[6], o[7] };
            __builtin_nontemporal_store(v0, reinterpret_cast<u32x4*>(dst)); __builtin_nontemporal_store(v1, reinterpret_cast<u32x4*>(dst) + 1);
        } else {
            dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
            dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
        }
    }
}

// PACKED = false: whole frame, row-major output; one lane = 4 consecutive pixels.
// PACKED = true : only owner tiles of this rank, output packed tile-major
//                 [local tile][128 rows][128 px]; block = 8 rows x 128 px of a tile.
#ifndef VR_DEFERRED_WAVES
#define VR_DEFERRED_WAVES 4
#endif
// The lighting pass as a pure stream: every plane of every pixel read, nothing known about any of them (the plane-state tracking is
// off, or the host holds the planes' pointers).  Round 3's kernel, kept as it was: with all seven loads in flight per lane and the
// table's barrier in front of them it moves the 36 B/px at 5.7 TB/s; the variant below, which asks first what it need not read,
// costs that case 10 us (221 vs 210 us at 8K).
template <bool PACKED, bool EXTRA, bool SHADOW = false, bool NT = false>
__global__ __launch_bounds__(256, VR_DEFERRED_WAVES) void k_deferred_stream(DeferredArgs a, const float* __restrict__ g_depth,
                                                   const uint32_t* __restrict__ g_diff, const uint32_t* __restrict__ g_spec,
                                                   const uint2* __restrict__ g_nrm, const uint2* __restrict__ g_emi,
                                                   uint2* __restrict__ out, const float* __restrict__ lut_g,
                                                   const int32_t* __restrict__ owned_tiles, ShadowArgs sh)
{
    __shared__ float lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();

    int px0, py;          // first pixel of this lane's quad
    size_t out_index;     // in pixels
    if (PACKED) {
        const int lt = blockIdx.x >> 4, rg = blockIdx.x & 15;
        const int tile = owned_tiles[lt];
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int row = rg * 8 + (threadIdx.x >> 5), col = (threadIdx.x & 31) * 4;
        px0 = tx * VR_OWNER_TILE + col; py = ty * VR_OWNER_TILE + row;
        out_index = ((size_t)lt * VR_OWNER_TILE + row) * VR_OWNER_TILE + col;
        if (px0 >= a.w || py >= a.h) return;
    } else {
        const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
        const size_t p = q * 4;
        if (p >= (size_t)a.w * a.h) return;
        py = (int)(p / (size_t)a.w); px0 = (int)(p - (size_t)py * a.w);
        out_index = p;
    }
    const size_t p = (size_t)py * a.w + px0;
    // issue every load first: 7 x 16 B per lane in flight
    // NT: streaming (non-temporal) G-buffer reads - together with the streaming stores (store_quad) the frame's 1.2 GB then
    // pass the caches by and the tile pass's texel tables (180 MB) are still in the Infinity Cache when the next frame's
    // tile pass starts.  Either one alone does not help: 929 MB of reads or 265 MB of dirty lines each flush the cache.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LD16(ptr) ({ u32x4 v_; if (NT) v_ = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr)); else v_ = *reinterpret_cast<const u32x4*>(ptr); \
                     make_uint4(v_.x, v_.y, v_.z, v_.w); })
    const uint4 dzu = LD16(g_depth + p);
    const float4 dz = make_float4(__uint_as_float(dzu.x), __uint_as_float(dzu.y), __uint_as_float(dzu.z), __uint_as_float(dzu.w));
    const uint4 df = LD16(g_diff + p);
    const uint4 sp = LD16(g_spec + p);
    const uint4 n0 = LD16(g_nrm + p);
    const uint4 n1 = LD16(g_nrm + p + 2);
    const uint4 e0 = LD16(g_emi + p);
    const uint4 e1 = LD16(g_emi + p + 2);
#undef LD16

    const float depth[4] = { dz.x, dz.y, dz.z, dz.w };
    const uint32_t dfa[4] = { df.x, df.y, df.z, df.w }, spa[4] = { sp.x, sp.y, sp.z, sp.w };
    const uint32_t na[8] = { n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w };
    const uint32_t ea[8] = { e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w };
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float rgb[3];
        shade_pixel<EXTRA, SHADOW>(a, lut, px0 + k, py, depth[k], dfa[k], spa[k], na[2 * k], na[2 * k + 1], ea[2 * k], ea[2 * k + 1], rgb, &sh);
        o[2 * k] = vr_float_to_half(rgb[0]) | (vr_float_to_half(rgb[1]) << 16);
        o[2 * k + 1] = vr_float_to_half(rgb[2]);          // alpha = 0
    }
    store_quad<PACKED, NT>(out, out_index, o);
}

template <bool PACKED, bool EXTRA, bool SHADOW = false, bool NT = false>
__global__ __launch_bounds__(256, VR_DEFERRED_WAVES) void k_deferred(DeferredArgs a, const float* __restrict__ g_depth,
                                                   const uint32_t* __restrict__ g_diff, const uint32_t* __restrict__ g_spec,
                                                   const uint2* __restrict__ g_nrm, const uint2* __restrict__ g_emi,
                                                   uint2* __restrict__ out, const float* __restrict__ lut_g,
                                                   const int32_t* __restrict__ owned_tiles, ShadowArgs sh, PlaneHints hints)
{
    __shared__ float lut[256];
    int px0, py;          // first pixel of this lane's quad
    size_t out_index;     // in pixels
    bool valid;
    if (PACKED) {
        const int lt = blockIdx.x >> 4, rg = blockIdx.x & 15;
        const int tile = owned_tiles[lt];
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int row = rg * 8 + (threadIdx.x >> 5), col = (threadIdx.x & 31) * 4;
        px0 = tx * VR_OWNER_TILE + col; py = ty * VR_OWNER_TILE + row;
        out_index = ((size_t)lt * VR_OWNER_TILE + row) * VR_OWNER_TILE + col;
        valid = px0 < a.w && py < a.h;
    } else {
        const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
        const size_t p = q * 4;
        valid = p < (size_t)a.w * a.h;
        py = (int)(p / (size_t)a.w); px0 = (int)(p - (size_t)py * a.w);
        out_index = p;
    }
    const size_t p = (size_t)py * a.w + px0;
    // What the library knows about the planes (plane-state tracking, vr_gbuffer): a region - 8 rows x 32 pixels, this quad lies
    // in one - that holds the clear values is not read at all, one that holds the terrain shader's specular constant in every
    // pixel keeps that plane in memory, and an emissive plane known to be zero is never read.  The values are substituted: the
    // arithmetic below is the same and so is the result.  (hints.region == NULL and emissive_zero == 0: nothing is known.)
    // The state byte and the sRGB table's entry are requested together, in front of everything else: one round trip, not two.
    uint32_t st = 0u;
    if (valid && hints.region != nullptr) st = hints.region[((size_t)(py >> 5) * hints.tiles32_x + (size_t)(px0 >> 5)) * 4 + (size_t)((py & 31) >> 3)];
    const float lut_r = lut_g[threadIdx.x];
    // issue every load first: up to 7 x 16 B per lane in flight
    // NT: streaming (non-temporal) G-buffer reads - together with the streaming stores (store_quad) the frame's 1.2 GB then
    // pass the caches by and the tile pass's texel tables (180 MB) are still in the Infinity Cache when the next frame's
    // tile pass starts.  Either one alone does not help: 929 MB of reads or 265 MB of dirty lines each flush the cache.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LD16(ptr) ({ u32x4 v_; if (NT) v_ = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr)); else v_ = *reinterpret_cast<const u32x4*>(ptr); \
                     make_uint4(v_.x, v_.y, v_.z, v_.w); })
    const bool known_clear = st == kRegionClear, known_spec = st == kRegionSpec;
    uint4 dzu = make_uint4(0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u), df = make_uint4(0u, 0u, 0u, 0u), n0 = df, n1 = df, e0 = df, e1 = df;
    uint4 sp = known_spec ? make_uint4(hints.spec_const, hints.spec_const, hints.spec_const, hints.spec_const) : df;
    if (valid && !known_clear) {
        dzu = LD16(g_depth + p);
        df = LD16(g_diff + p);
        if (!known_spec) sp = LD16(g_spec + p);
        n0 = LD16(g_nrm + p);
        n1 = LD16(g_nrm + p + 2);
        if (!hints.emissive_zero) { e0 = LD16(g_emi + p); e1 = LD16(g_emi + p + 2); }
    }
    lut[threadIdx.x] = lut_r;
    __syncthreads();
    if (!valid) return;
    // A wave whose texels all hold the clear values (sky: a fifth of the flythrough's pixels, whole rows of waves) has nothing to
    // shade: with albedo, F0, occlusion, the normal and the emissive term all zero every product of the model below is a
    // (finite) factor times zero - the radiance is +0 in every channel, which is what the loop would store.  (Wave-uniform.)
    // "Finite" is the one assumption: a light with non-finite parameters, or a clip-to-world matrix that sends the far plane to
    // w = 0, would make the loop produce NaN for such a texel where this stores 0 - inputs this path does not see (the host
    // builds both from a camera and a light list it has validated).
    {
        const uint32_t any = (dzu.x ^ 0x3f800000u) | (dzu.y ^ 0x3f800000u) | (dzu.z ^ 0x3f800000u) | (dzu.w ^ 0x3f800000u)
                           | df.x | df.y | df.z | df.w | sp.x | sp.y | sp.z | sp.w | n0.x | n0.y | n0.z | n0.w | n1.x | n1.y | n1.z | n1.w
                           | e0.x | e0.y | e0.z | e0.w | e1.x | e1.y | e1.z | e1.w;
        if (__all(any == 0u)) {
            const uint32_t z[8] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
            store_quad<PACKED, NT>(out, out_index, z);
            return;
        }
    }
    const float4 dz = make_float4(__uint_as_float(dzu.x), __uint_as_float(dzu.y), __uint_as_float(dzu.z), __uint_as_float(dzu.w));
#undef LD16

    const float depth[4] = { dz.x, dz.y, dz.z, dz.w };
    const uint32_t dfa[4] = { df.x, df.y, df.z, df.w }, spa[4] = { sp.x, sp.y, sp.z, sp.w };
    const uint32_t na[8] = { n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w };
    const uint32_t ea[8] = { e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w };
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float rgb[3];
        shade_pixel<EXTRA, SHADOW>(a, lut, px0 + k, py, depth[k], dfa[k], spa[k], na[2 * k], na[2 * k + 1], ea[2 * k], ea[2 * k + 1], rgb, &sh);
        o[2 * k] = vr_float_to_half(rgb[0]) | (vr_float_to_half(rgb[1]) << 16);
        o[2 * k + 1] = vr_float_to_half(rgb[2]);          // alpha = 0
    }
    store_quad<PACKED, NT>(out, out_index, o);
}

// Generic fallback for widths that are not a multiple of 4: one pixel per lane.
__global__ __launch_bounds__(256) void k_deferred_scalar(DeferredArgs a, const float* __restrict__ g_depth,
                                                          const uint32_t* __restrict__ g_diff, const uint32_t* __restrict__ g_spec,
                                                          const uint2* __restrict__ g_nrm, const uint2* __restrict__ g_emi,
                                                          uint2* __restrict__ out, const float* __restrict__ lut_g, ShadowArgs sh,
                                                          int use_shadow)
{
    __shared__ float lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)a.w * a.h) return;
    const int py = (int)(p / (size_t)a.w), px = (int)(p - (size_t)py * a.w);
    const uint2 n = g_nrm[p], e = g_emi[p];
    float rgb[3];
    if (use_shadow) shade_pixel<true, true>(a, lut, px, py, g_depth[p], g_diff[p], g_spec[p], n.x, n.y, e.x, e.y, rgb, &sh);
    else shade_pixel<true>(a, lut, px, py, g_depth[p], g_diff[p], g_spec[p], n.x, n.y, e.x, e.y, rgb);
    out[p] = make_uint2(vr_float_to_half(rgb[0]) | (vr_float_to_half(rgb[1]) << 16), vr_float_to_half(rgb[2]));
}

static int fill_light(const vr_light& l, DevLight& d, bool allow_extra)
{
    VR_REQUIRE(l.type == VR_LIGHT_DIRECTIONAL || l.type == VR_LIGHT_POINT || l.type == VR_LIGHT_SPOT, "unknown light type");
    VR_REQUIRE(allow_extra || (l.type != VR_LIGHT_SPOT && !(l.type == VR_LIGHT_POINT && l.radius > 0.0f)),
               "the tiled pass takes directional and punctual point lights only");
    VR_REQUIRE(l.type != VR_LIGHT_SPOT || l.outer_angle > l.inner_angle, "spot light needs outer_angle > inner_angle");
    for (int k = 0; k < 3; k++) { d.dir[k] = l.direction[k]; d.pos[k] = l.position[k]; d.color[k] = l.color[k]; }
    d.type = l.type; d.intensity = l.intensity;
    d.inv_range = l.type != VR_LIGHT_DIRECTIONAL ? l.angular_size_or_inv_range : 0.0f;
    const double half = l.type == VR_LIGHT_DIRECTIONAL ? 0.5 * (double)l.angular_size_or_inv_range : 0.0;
    d.cosH = (float)cos(half); d.sinH = (float)sin(half); d.tanH = (float)tan(half);
    d.radius = l.type != VR_LIGHT_DIRECTIONAL ? l.radius : 0.0f;
    d.inner_angle = l.inner_angle; d.outer_angle = l.outer_angle; d.pad0 = d.pad1 = 0.0f;
    return VR_OK;
}

// The lighting kernels' constant block for a view and a light list (also the fused tile pass's: vr_terrain_render_lit).
int vr_deferred_make_args(const vr_view* view, int w, int h, const vr_light* lights, int32_t num_lights, const float amb_top[3],
                          const float amb_bottom[3], DeferredArgs* out, bool* extra_lights)
{
    DeferredArgs& a = *out;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 16; i++) a.c2w[i] = view->clip_to_world[i];
    for (int i = 0; i < 3; i++) { a.cam[i] = view->camera_pos[i]; a.amb_top[i] = amb_top[i]; a.amb_bot[i] = amb_bottom[i]; }
    a.w = w; a.h = h; a.sx = 2.0f / (float)w; a.sy = -2.0f / (float)h;
    a.num_lights = num_lights;
    *extra_lights = false;
    for (int i = 0; i < num_lights; i++) {
        int rc = fill_light(lights[i], a.lights[i], true); if (rc) return rc;
        *extra_lights = *extra_lights || a.lights[i].type == VR_LIGHT_SPOT || a.lights[i].radius > 0.0f;
        if (a.lights[i].type != VR_LIGHT_DIRECTIONAL) a.exact_pos = 1;
    }
    return VR_OK;
}

static int deferred_light(vr_context* ctx, const vr_view* view, vr_gbuffer* gb, const vr_light* lights,
                          int32_t num_lights, const float amb_top[3], const float amb_bottom[3],
                          vr_image* hdr, const vr_partition* part, const vr_shadow_binding* shadow)
{
    VR_REQUIRE(ctx && view && gb && hdr && amb_top && amb_bottom, "NULL argument");
    ShadowArgs sh;
    memset(&sh, 0, sizeof(sh));
    if (shadow) {
        VR_REQUIRE(shadow->light_view && shadow->shadow_map, "shadow binding has NULL members");
        VR_REQUIRE(shadow->light_index >= 0 && shadow->light_index < num_lights, "shadow light_index out of range");
        VR_REQUIRE(lights[shadow->light_index].type == VR_LIGHT_DIRECTIONAL, "only a directional light carries the cascaded shadow map");
        VR_REQUIRE(shadow->shadow_map->w == shadow->shadow_map->h && shadow->shadow_map->w == shadow->light_view->viewport_w
                   && shadow->light_view->viewport_h == shadow->light_view->viewport_w, "shadow map must be square and match the light view's viewport");
        VR_REQUIRE(shadow->shadow_map->ctx->device == ctx->device, "shadow map lives on another device");
        for (int i = 0; i < 16; i++) sh.w2c[i] = shadow->light_view->world_to_clip[i];
        { VR_HIP(hipSetDevice(ctx->device)); const int rc = vr_gbuffer_materialise(shadow->shadow_map, ctx->stream); if (rc) return rc; }
        sh.depth = shadow->shadow_map->depth; sh.res = shadow->shadow_map->w; sh.light_index = shadow->light_index;
        sh.bias = shadow->depth_bias; sh.out_of_bounds = lights[shadow->light_index].out_of_bounds_shadow;
    }
    VR_REQUIRE(num_lights >= 0 && num_lights <= kMaxLights, "at most 16 lights (terrain_cb.h:15)");
    VR_REQUIRE(num_lights == 0 || lights, "lights is NULL");
    VR_REQUIRE(view->viewport_w == gb->w && view->viewport_h == gb->h && view->viewport_x == 0 && view->viewport_y == 0,
               "view viewport must cover the G-buffer");
    VR_HIP(hipSetDevice(ctx->device));
    DeferredArgs a;
    bool extra = false;
    { int rc = vr_deferred_make_args(view, gb->w, gb->h, lights, num_lights, amb_top, amb_bottom, &a, &extra); if (rc) return rc; }
    const size_t npx = (size_t)gb->w * gb->h;
    const bool packed = part != nullptr;     // a partition (even of one rank) selects the packed tile-major output
    PlaneHints hints;
    { int rc = vr_gbuffer_plane_hints(gb, ctx->stream, &hints); if (rc) return rc; }
    const bool hinted = hints.region != nullptr || hints.emissive_zero != 0;       // else: the pure-stream kernel
    VrKernelScope ks(ctx, VR_K_DEFERRED, ctx->stream, true);
    if (packed) {
        const PartTables* pt = nullptr;
        int rc = vr_partition_tables(ctx, gb->w, gb->h, part, &pt);
        if (rc) return rc;
        VR_REQUIRE((size_t)pt->max_owned * VR_OWNER_TILE * VR_OWNER_TILE * 6 <= hdr->capacity_bytes,
                   "hdr_out is smaller than vr_partition_packed_bytes()");
        VR_REQUIRE(gb->w % 4 == 0, "partitioned frames need a width that is a multiple of 4");
        a.tiles_x = (gb->w + VR_OWNER_TILE - 1) / VR_OWNER_TILE;
        if (pt->num_owned > 0) {
            if (!hinted) {
                auto kern = shadow ? k_deferred_stream<true, true, true> : (extra ? k_deferred_stream<true, true> : k_deferred_stream<true, false>);
                VR_LAUNCH_TIMED(ks, kern, dim3((unsigned)pt->num_owned * 16), dim3(256), ctx->stream, a, gb->depth, gb->diffuse,
                                   gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data, ctx->d_srgb_lut, pt->d_owned_tiles, sh);
            } else {
            auto kern = shadow ? k_deferred<true, true, true> : (extra ? k_deferred<true, true> : k_deferred<true, false>);
            VR_LAUNCH_TIMED(ks, kern, dim3((unsigned)pt->num_owned * 16), dim3(256), ctx->stream, a, gb->depth, gb->diffuse,
                               gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data, ctx->d_srgb_lut, pt->d_owned_tiles, sh, hints);
            }
        }
    } else {
        VR_REQUIRE(npx * 8 <= hdr->capacity_bytes, "hdr_out is smaller than the frame");
        if (gb->w % 4 == 0) {
            const size_t quads = npx / 4;
            // streaming loads and stores at every frame size, like the tile pass's stores (vr_raster.hip: measured per size)
            const bool nt = true;
            if (!hinted) {
                auto kern = shadow ? k_deferred_stream<false, true, true, true> : extra ? k_deferred_stream<false, true, false, true>
                                                                                        : k_deferred_stream<false, false, false, true>;
                VR_LAUNCH_TIMED(ks, kern, dim3((unsigned)((quads + 255) / 256)), dim3(256), ctx->stream, a, gb->depth, gb->diffuse,
                                   gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data, ctx->d_srgb_lut, (const int32_t*)nullptr, sh);
            } else {
            auto kern = shadow ? (nt ? k_deferred<false, true, true, true> : k_deferred<false, true, true, false>)
                               : extra ? (nt ? k_deferred<false, true, false, true> : k_deferred<false, true, false, false>)
                                       : (nt ? k_deferred<false, false, false, true> : k_deferred<false, false, false, false>);
            VR_LAUNCH_TIMED(ks, kern, dim3((unsigned)((quads + 255) / 256)), dim3(256), ctx->stream, a, gb->depth, gb->diffuse,
                               gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data, ctx->d_srgb_lut, (const int32_t*)nullptr, sh, hints);
            }
        } else {
            VR_LAUNCH_TIMED(ks, k_deferred_scalar, dim3((unsigned)((npx + 255) / 256)), dim3(256), ctx->stream, a, gb->depth, gb->diffuse,
                               gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data, ctx->d_srgb_lut, sh, shadow ? 1 : 0);
        }
    }
    VR_HIP(hipGetLastError());
    return VR_OK;
}

extern "C" VR_API int vr_deferred_light(vr_context* ctx, const vr_view* view, vr_gbuffer* gb, const vr_light* lights,
                                         int32_t num_lights, const float amb_top[3], const float amb_bottom[3],
                                         vr_image* hdr, const vr_partition* part)
{
    return deferred_light(ctx, view, gb, lights, num_lights, amb_top, amb_bottom, hdr, part, nullptr);
}

// DeferredLightingPass::Render with DirectionalLight::shadowMap set (Renderer.cpp:336, 427)
extern "C" VR_API int vr_deferred_light_shadowed(vr_context* ctx, const vr_view* view, vr_gbuffer* gb, const vr_light* lights,
                                                  int32_t num_lights, const float amb_top[3], const float amb_bottom[3],
                                                  vr_image* hdr, const vr_partition* part, const vr_shadow_binding* shadow)
{
    VR_REQUIRE(shadow, "shadow binding is NULL (use vr_deferred_light)");
    return deferred_light(ctx, view, gb, lights, num_lights, amb_top, amb_bottom, hdr, part, shadow);
}

// ---- tiled deferred lighting for many point lights (BASELINE config 5) ---------------------------
// Two kernels.
//   k_light_cull: one workgroup per 128x128 macro tile (= owner tile).  Depth range of the macro tile and of each of its
//      sixteen 32x32 light tiles -> the world-space boxes of those frustum cells (8 corners each) -> every light is tested
//      against the macro box and, if it touches it, against the sixteen small ones; survivors are appended to the light
//      tiles' lists in global memory (count, then indices, in light order: ballot + popcount prefix).
//   k_deferred_tiled: one workgroup = one 32x32 light tile, one lane = 4 horizontally adjacent pixels (16-byte loads
//      and stores, as in k_deferred).  The tile's lights are staged through LDS 256 at a time (32 B each, one fetch
//      latency for the whole list), then every lane walks them (LDS broadcast reads) with the same BRDF as k_deferred.
// A light culled here has zero attenuation for every pixel of the tile, so the sum equals the all-lights loop of the
// oracle.  (Round 2's first version culled inside the shading kernel - per-pixel position boxes, five barriers and a
// 32-KB list per workgroup; its fixed cost was 150 us per 8K frame over k_deferred's.)
constexpr int kLightTile = 32;
constexpr int kTileLightCap = VR_TILE_LIGHT_CAP;
constexpr int kStageLights = 256;
// 32 B per staged light: colour premultiplied by the intensity; w = 0 for a point light; for a directional one
// w = cos and inv_range = sin of its angular half size.
struct TiledLight { float vec[3]; float inv_range; float color[3]; float w; };

constexpr int kMacroTile = VR_OWNER_TILE;
constexpr int kSubSide = kMacroTile / kLightTile;       // 4 light tiles per macro tile side
constexpr int kSubTiles = kSubSide * kSubSide;
static_assert(kSubTiles == 16 && kMacroTile == 128, "k_light_cull's lane mapping assumes 128-px macro tiles of 4x4 light tiles");

// squared distance from a point to a box, compared with the light's range (attenuation is exactly 0 from the range outwards).
// The caller passes r2 = range^2 * 1.0001: the test's own fp32 rounding (three squares and two sums, the reciprocal and
// the square of the range: < 8 u relative) cannot turn a touching sphere into a miss.
__device__ __forceinline__ bool sphere_touches(const float* __restrict__ box, const float pos[3], float r2)
{
    float d2 = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) { const float d = fmax1(fmax1(box[c] - pos[c], pos[c] - box[3 + c]), 0.0f); d2 += d * d; }
    return d2 <= r2;
}

// RANGES: the light tiles' depth ranges come from the tile pass that rendered the G-buffer (vr_gbuffer::d_ranges: the same
// minima / maxima of the depths below 1.0, merged with atomics while the pixels were resolved) instead of from a second
// pass over the depth plane - the kernel's first phase, and most of its time: sixteen 16-byte loads per lane from 512-byte
// rows 30 KB apart.  The entries are consumed: reset to "none" for the next frame.
template <bool RANGES>
__global__ __launch_bounds__(256) void k_light_cull(DeferredArgs a, const DevLight* __restrict__ lights, int num_lights,
                                                     const float* __restrict__ g_depth, int macro_x,
                                                     const int32_t* __restrict__ owned_tiles, uint32_t* __restrict__ lists, int stride,
                                                     int tiles32_x, int tiles32_y, uint32_t* __restrict__ overflow_flag,
                                                     uint32_t* __restrict__ macro_scratch, uint2* __restrict__ ranges)
{
    __shared__ uint32_t s_dmin[kSubTiles], s_dmax[kSubTiles];   // bits of non-negative floats: ordered like the floats
    __shared__ float s_box[kSubTiles + 1][6];                   // the light tiles' boxes, then the macro tile's
    __shared__ uint32_t s_kept[4];
    __shared__ uint32_t s_covered;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = owned_tiles ? owned_tiles[blockIdx.x] : (int)blockIdx.x;
    const int ty = tile / macro_x, tx = tile - ty * macro_x;
    const int x0 = tx * kMacroTile, y0 = ty * kMacroTile;
    if (tid < kSubTiles) {
        uint2 r = make_uint2(0x7f800000u, 0u);
        if (RANGES) {
            const int gx = tx * kSubSide + (tid & 3), gy = ty * kSubSide + (tid >> 2);
            if (gx < tiles32_x && gy < tiles32_y) { uint2* e = ranges + (size_t)gy * tiles32_x + gx; r = *e; *e = make_uint2(0x7f800000u, 0u); }
        }
        s_dmin[tid] = r.x; s_dmax[tid] = r.y;
    }
    if (tid == 0) s_covered = 0u;
    __syncthreads();
    if (!RANGES) {
    // ---- depth range per light tile: lane = 4 px of a row; a wave holds two rows of the macro tile per step
    const int sub_x = (tid & 31) >> 3;
    // all 16 loads first (a row outside the frame re-reads the frame's first texels and is masked below): one memory
    // round trip per workgroup instead of sixteen
    float4 dv[kMacroTile / 8];
    const int px = x0 + (tid & 31) * 4;
#pragma unroll
    for (int r = 0; r < kMacroTile / 8; r++) {
        const int py = y0 + r * 8 + (tid >> 5);
        const bool ok = py < a.h && px < a.w;                           // the frame width is a multiple of 4
        dv[r] = *reinterpret_cast<const float4*>(g_depth + (ok ? (size_t)py * a.w + px : (size_t)0));
        if (!ok) dv[r] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    }
#pragma unroll
    for (int sub_y = 0; sub_y < kSubSide; sub_y++) {
        float dmin = 2.0f, dmax = -1.0f;
#pragma unroll
        for (int r = sub_y * 4; r < sub_y * 4 + 4; r++) {
            const float v[4] = { dv[r].x, dv[r].y, dv[r].z, dv[r].w };
#pragma unroll
            for (int k = 0; k < 4; k++) if (v[k] < 1.0f) { dmin = fmin1(dmin, v[k]); dmax = fmax1(dmax, v[k]); }
        }
#pragma unroll
        for (int off = 1; off <= 4; off <<= 1) { dmin = fmin1(dmin, __shfl_xor(dmin, off)); dmax = fmax1(dmax, __shfl_xor(dmax, off)); }
        dmin = fmin1(dmin, __shfl_xor(dmin, 32)); dmax = fmax1(dmax, __shfl_xor(dmax, 32));
        if ((lane & 39) == 0 && dmin <= dmax) {                          // lanes 0, 8, 16, 24: one per light-tile column
            atomicMin(&s_dmin[sub_y * kSubSide + sub_x], __float_as_uint(dmin) & 0x7fffffffu);
            atomicMax(&s_dmax[sub_y * kSubSide + sub_x], __float_as_uint(dmax) & 0x7fffffffu);
        }
    }
    __syncthreads();
    }
    // ---- boxes: 8 lanes per cell (its corners: window edges x {dmin, dmax}), reconstructed in the shading pass's exact order
    if (tid < (kSubTiles + 1) * 8) {
        const int b = tid >> 3, corner = tid & 7;
        uint32_t bmin, bmax; int cx0, cy0, cx1, cy1;
        if (b < kSubTiles) {
            bmin = s_dmin[b]; bmax = s_dmax[b];
            cx0 = x0 + (b & 3) * kLightTile; cy0 = y0 + (b >> 2) * kLightTile; cx1 = cx0 + kLightTile; cy1 = cy0 + kLightTile;
        } else {
            bmin = 0x7f800000u; bmax = 0u;
            for (int i = 0; i < kSubTiles; i++) { bmin = min(bmin, s_dmin[i]); bmax = max(bmax, s_dmax[i]); }
            cx0 = x0; cy0 = y0; cx1 = x0 + kMacroTile; cy1 = y0 + kMacroTile;
        }
        const bool covered = bmin <= bmax;
        // Each corner in the shading pass's arithmetic, together with a rigorous bound of what fp32 rounding can have done
        // to it - and to any pixel of the cell (derivation: DESIGN.md 4, "the culling boxes' pad").  With u = 2^-24 and
        // e_j = ((cx M0j + cy M1j) + depth M2j) + M3j, |cx|, |cy| <= 1:
        //   |e_j - exact| <= 8u T_j,  T_j = |M0j| + |M1j| + |depth M2j| + |M3j|   (4 roundings on the longest path + the
        //                                                                          3u the window -> clip step put into cx, cy)
        //   position_c = e_c * (1 / e_3):  |error| <= (8u T_c + |position_c| 8u T_3) / (e_3 - 8u T_3) + 2u |position_c|
        //                                  (from e_c/e_3 - E_c/E_3 = (e_c - E_c)/E_3 - (e_c/e_3)(e_3 - E_3)/E_3, E = the exact sums)
        // T_j grows with the depth and e_3 (= 1 / w, positive in front of the camera) is affine over the cell, so the
        // cell's worst case is T_j at its largest depth, the smallest e_3 of its corners and the largest |position_c| of
        // its corners.  A pixel's computed position then lies within 2 x that of the box of the computed corners (once for
        // the corners, once for the pixel); where w cancels completely (e_3 <= 8u T_3: depth next to 1 with a far plane
        // at 10^4 near planes away) nothing can be said and the cell keeps every light.
        float lo[3], hi[3], dmin3, wmax[3];
        float T[4];
        {
#pragma clang fp contract(off)
            const float wx = (float)(corner & 1 ? min(cx1, a.w) : cx0), wy = (float)(corner & 2 ? min(cy1, a.h) : cy0);
            const float depth = __uint_as_float(corner & 4 ? bmax : bmin);
            const float cx = wx * a.sx + -1.0f, cy = wy * a.sy + 1.0f;
            float e4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) e4[j] = ((cx * a.c2w[0 * 4 + j] + cy * a.c2w[1 * 4 + j]) + depth * a.c2w[2 * 4 + j]) + a.c2w[3 * 4 + j];
#pragma unroll
            for (int c = 0; c < 3; c++) { lo[c] = hi[c] = e4[c] / e4[3]; wmax[c] = fabsf(lo[c]); }
            dmin3 = e4[3];
            const float dfar = __uint_as_float(bmax);
#pragma unroll
            for (int j = 0; j < 4; j++) T[j] = ((fabsf(a.c2w[0 * 4 + j]) + fabsf(a.c2w[1 * 4 + j])) + fabsf(dfar * a.c2w[2 * 4 + j])) + fabsf(a.c2w[3 * 4 + j]);
        }
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                lo[c] = fmin1(lo[c], __shfl_xor(lo[c], off)); hi[c] = fmax1(hi[c], __shfl_xor(hi[c], off));
                wmax[c] = fmax1(wmax[c], __shfl_xor(wmax[c], off));
            }
            dmin3 = fmin1(dmin3, __shfl_xor(dmin3, off));
        }
        if (corner == 0) {
            const float u8 = 8.0f * 5.9604645e-8f, u2 = 2.0f * 5.9604645e-8f;
            const float dlo = dmin3 - u8 * T[3] * 1.001f;                // (the 1.001s: this arithmetic rounds too; all terms are positive)
            // nothing covered: an empty box (no light touches it; the tile's pixels are background and receive none)
            for (int c = 0; c < 3; c++) {
                // pad = f(W) with W = the largest |position_c| of any pixel <= wmax + pad itself: f is affine, f(W) = alpha + beta W,
                // so pad = (alpha + beta wmax) / (1 - beta); beta >= 1/2 means the positions themselves are noise
                float pad = 3.0e38f;
                if (dlo > 0.0f) {
                    const float beta = 2.0f * 1.001f * (u8 * T[3] / dlo + u2), alpha = 2.0f * 1.001f * (u8 * T[c] / dlo);
                    if (beta < 0.5f) pad = (alpha + beta * wmax[c]) / (1.0f - beta) * 1.001f;
                }
                if (!(pad < 3.0e38f)) pad = 3.0e38f;
                s_box[b][c] = covered ? fmax1(lo[c] - pad, -3.0e38f) : 3.0e38f; s_box[b][3 + c] = covered ? fmin1(hi[c] + pad, 3.0e38f) : -3.0e38f;
            }
            if (covered && b < kSubTiles) atomicOr(&s_covered, 1u << b);
        }
    }
    __syncthreads();
    const uint32_t covered_mask = s_covered;
    const uint32_t cap = (uint32_t)(stride - 1);
    uint32_t* __restrict__ my_list = nullptr;                         // lanes 0..15 of every wave: light tile `lane`'s list
    if (lane < kSubTiles) {
        const int gx = tx * kSubSide + (lane & 3), gy = ty * kSubSide + (lane >> 2);
        if (gx < tiles32_x && gy < tiles32_y) my_list = lists + (size_t)(gy * tiles32_x + gx) * stride;
    }
    // ---- A. every wave takes a quarter of the light list and keeps, in order, the lights that touch the macro tile's box
    // (its region of the scratch list: no synchronisation until all four are done; the next 64 lights are in flight
    // while the current ones are tested)
    struct Lite { float pos[3]; float inv_range; int type; };
    auto fetch = [&](int li, int hi) { Lite l; l.pos[0] = l.pos[1] = l.pos[2] = 0.0f; l.inv_range = 0.0f; l.type = -1;
        if (li < hi) { const DevLight& L = lights[li]; l.pos[0] = L.pos[0]; l.pos[1] = L.pos[1]; l.pos[2] = L.pos[2]; l.inv_range = L.inv_range; l.type = L.type; }
        return l; };
    const int quarter = (num_lights + 3) / 4;
    const int lo_w = min(wave * quarter, num_lights), hi_w = min(lo_w + quarter, num_lights);
    uint32_t* __restrict__ kept_list = macro_scratch + (size_t)tile * (size_t)(quarter * 4);
    uint32_t kept = 0u;                                               // (wave-uniform)
    if (covered_mask != 0u) {
        Lite cur = fetch(lo_w + lane, hi_w);
        for (int base = lo_w; base < hi_w; base += 64) {
            const Lite nxt = fetch(base + 64 + lane, hi_w);
            bool keep = false;
            if (cur.type >= 0) {
                if (cur.type == VR_LIGHT_DIRECTIONAL || !(cur.inv_range > 0.0f)) keep = true;
                else { const float r = 1.0f / cur.inv_range; keep = sphere_touches(s_box[kSubTiles], cur.pos, (r * r) * 1.0001f); }
            }
            const unsigned long long m = __ballot(keep);
            if (keep) kept_list[(uint32_t)lo_w + kept + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)(base + lane);
            kept += (uint32_t)__popcll(m);
            cur = nxt;
        }
    }
    if (lane == 0) s_kept[wave] = kept;
    __threadfence_block();
    __syncthreads();
    if (wave != 0) return;
    // ---- B. one wave sorts the survivors into the sixteen light tiles' lists, in light order (ballot + popcount prefix)
    uint32_t cnt[kSubTiles];
#pragma unroll
    for (int b = 0; b < kSubTiles; b++) cnt[b] = 0u;
    for (int w = 0; w < 4; w++) {
        const uint32_t K = s_kept[w];
        const uint32_t* __restrict__ src = kept_list + min(w * quarter, num_lights);
        for (uint32_t i0 = 0; i0 < K; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            uint32_t mask = 0u, li = 0u;
            if (i < K) {
                li = src[i];
                const DevLight& L = lights[li];
                if (L.type == VR_LIGHT_DIRECTIONAL || !(L.inv_range > 0.0f)) mask = covered_mask;
                else {
                    const float pos[3] = { L.pos[0], L.pos[1], L.pos[2] };
                    const float r = 1.0f / L.inv_range, r2 = (r * r) * 1.0001f;
#pragma unroll
                    for (int b = 0; b < kSubTiles; b++) mask |= sphere_touches(s_box[b], pos, r2) ? 1u << b : 0u;
                }
            }
#pragma unroll
            for (int b = 0; b < kSubTiles; b++) {
                const unsigned long long m = __ballot((mask >> b) & 1u);
                if (m == 0ull) continue;
                uint32_t* __restrict__ list = reinterpret_cast<uint32_t*>((uintptr_t)__shfl((unsigned long long)(uintptr_t)my_list, b));
                if ((mask >> b) & 1u) {
                    const uint32_t slot = cnt[b] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (slot < cap) list[1u + slot] = li;
                    else atomicOr(overflow_flag, 1u);
                }
                cnt[b] += (uint32_t)__popcll(m);
            }
        }
    }
    uint32_t mine = 0u;
#pragma unroll
    for (int b = 0; b < kSubTiles; b++) if (lane == b) mine = cnt[b];
    if (my_list) my_list[0] = min(mine, cap);
}

#ifndef VR_TILED_PXB
#define VR_TILED_PXB 1
#endif
template <bool PACKED, int PXB, bool NT = false>
__global__ __launch_bounds__(256) void k_deferred_tiled(DeferredArgs a, const DevLight* __restrict__ lights,
                                                         const float* __restrict__ g_depth, const uint32_t* __restrict__ g_diff,
                                                         const uint32_t* __restrict__ g_spec, const uint2* __restrict__ g_nrm,
                                                         const uint2* __restrict__ g_emi, uint2* __restrict__ out,
                                                         const float* __restrict__ lut_g, const int32_t* __restrict__ owned_tiles,
                                                         const uint32_t* __restrict__ lists, int stride, int tiles32_x, PlaneHints hints)
{
#pragma clang fp contract(fast)
    __shared__ float lut[256];
    __shared__ TiledLight s_light[kStageLights];
    const int tid = threadIdx.x;
    lut[tid] = lut_g[tid];

    int px0, py, gx, gy; size_t out_index;
    const int lx = (tid & 7) * 4, ly = tid >> 3;                     // 8 lanes x 4 px per row, 32 rows
    if (PACKED) {
        const int lt = blockIdx.x / kSubTiles, st = blockIdx.x - lt * kSubTiles;
        const int tile = owned_tiles[lt];
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        gx = tx * kSubSide + (st & 3); gy = ty * kSubSide + (st >> 2);
        const int ox = (st & 3) * kLightTile + lx, oy = (st >> 2) * kLightTile + ly;
        out_index = ((size_t)lt * VR_OWNER_TILE + oy) * VR_OWNER_TILE + ox;
    } else {
        gy = blockIdx.x / tiles32_x; gx = blockIdx.x - gy * tiles32_x;
        out_index = (size_t)(gy * kLightTile + ly) * a.w + (gx * kLightTile + lx);
    }
    px0 = gx * kLightTile + lx; py = gy * kLightTile + ly;
    const bool inside = px0 < a.w && py < a.h;                       // the frame width is a multiple of 4
    // (a packed owner tile may hang over the frame's edge: light tiles outside it have no list)
    const bool listed = gx < tiles32_x && gy * kLightTile < a.h;
    const uint32_t* __restrict__ list = lists + (size_t)(gy * tiles32_x + gx) * stride;
    const uint32_t n = listed ? list[0] : 0u;

    // Plane-state tracking (as in k_deferred): this light tile is a raster tile's four regions - one word.  A tile that holds only
    // clear values has no light and no surface: +0 everywhere, nothing read (workgroup-uniform, in front of the barrier); a region's
    // constant specular plane and a zero emissive plane are not read.
    uint32_t st = 0u;
    if (hints.region != nullptr && listed) {
        const uint32_t st4 = reinterpret_cast<const uint32_t*>(hints.region)[gy * tiles32_x + gx];
        if (st4 == kRegionClear * 0x01010101u) {
            const uint32_t z[8] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
            if (inside) store_quad<PACKED, NT>(out, out_index, z);
            return;
        }
        st = (st4 >> (8 * (ly >> 3))) & 255u;
    }
    float depth[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
    uint32_t dfa[4] = { 0, 0, 0, 0 }, spa[4] = { 0, 0, 0, 0 }, na[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, ea[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (st == kRegionSpec) spa[0] = spa[1] = spa[2] = spa[3] = hints.spec_const;
    if (inside && st != kRegionClear) {
        const size_t p = (size_t)py * a.w + px0;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));     // NT: streaming loads, as in k_deferred
#define LD16(ptr) ({ u32x4 v_; if (NT) v_ = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr)); else v_ = *reinterpret_cast<const u32x4*>(ptr); \
                     make_uint4(v_.x, v_.y, v_.z, v_.w); })
        const uint4 dzu = LD16(g_depth + p);
        const float4 dz = make_float4(__uint_as_float(dzu.x), __uint_as_float(dzu.y), __uint_as_float(dzu.z), __uint_as_float(dzu.w));
        const uint4 df = LD16(g_diff + p);
        uint4 sp = make_uint4(spa[0], spa[1], spa[2], spa[3]);
        if (st != kRegionSpec) sp = LD16(g_spec + p);
        const uint4 n0 = LD16(g_nrm + p);
        const uint4 n1 = LD16(g_nrm + p + 2);
        uint4 e0 = make_uint4(0u, 0u, 0u, 0u), e1 = e0;
        if (!hints.emissive_zero) { e0 = LD16(g_emi + p); e1 = LD16(g_emi + p + 2); }
#undef LD16
        depth[0] = dz.x; depth[1] = dz.y; depth[2] = dz.z; depth[3] = dz.w;
        dfa[0] = df.x; dfa[1] = df.y; dfa[2] = df.z; dfa[3] = df.w; spa[0] = sp.x; spa[1] = sp.y; spa[2] = sp.z; spa[3] = sp.w;
        na[0] = n0.x; na[1] = n0.y; na[2] = n0.z; na[3] = n0.w; na[4] = n1.x; na[5] = n1.y; na[6] = n1.z; na[7] = n1.w;
        ea[0] = e0.x; ea[1] = e0.y; ea[2] = e0.z; ea[3] = e0.w; ea[4] = e1.x; ea[5] = e1.y; ea[6] = e1.z; ea[7] = e1.w;
    }
    uint32_t o[8];
    // The tile's lights go through LDS 256 at a time: one round - staged once, before the pixel loop - unless the tile
    // keeps more than that; then every pixel batch walks the rounds itself (barriers inside; n is workgroup-uniform).
    const bool multi = n > (uint32_t)kStageLights;
    auto stage = [&](uint32_t chunk) {
        if (chunk + (uint32_t)tid < n) {
            const DevLight L = lights[list[1u + chunk + tid]];
            const bool dir = L.type == VR_LIGHT_DIRECTIONAL;
            TiledLight t;
            const float* v = dir ? L.dir : L.pos;
            t.vec[0] = v[0]; t.vec[1] = v[1]; t.vec[2] = v[2]; t.inv_range = dir ? L.sinH : L.inv_range;
            t.color[0] = L.color[0] * L.intensity; t.color[1] = L.color[1] * L.intensity; t.color[2] = L.color[2] * L.intensity;
            t.w = dir ? L.cosH : 0.0f;
            s_light[tid] = t;
        }
    };
    if (!multi) stage(0u);
    __syncthreads();                                                 // (also publishes the decode table)
#pragma unroll
    for (int kb = 0; kb < 4; kb += PXB) {
        // (a background pixel's position is never used: albedo = F0 = N = 0 and it receives no light)
        Surface sv[PXB];
        float dT[PXB][3], sT[PXB][3];
#pragma unroll
        for (int j = 0; j < PXB; j++) {
            const int k = kb + j;
            sv[j] = decode_surface(a, lut, px0 + k, py, depth[k], dfa[k], spa[k], na[2 * k], na[2 * k + 1], ea[2 * k], ea[2 * k + 1]);
#pragma unroll
            for (int c = 0; c < 3; c++) dT[j][c] = sT[j][c] = 0.0f;
        }
        for (uint32_t chunk = 0; chunk < n; chunk += kStageLights) {
            const uint32_t m = min(n - chunk, (uint32_t)kStageLights);
            if (multi) { __syncthreads(); stage(chunk); __syncthreads(); }
            for (uint32_t i = 0; i < m; i++) {
                const TiledLight& t = s_light[i];
                if (t.w > 0.0f) {                                  // directional (block-uniform branch)
                    const float tanH = t.inv_range * fast_rcp(t.w);
#pragma unroll
                    for (int j = 0; j < PXB; j++)
                        if (depth[kb + j] < 1.0f) add_light(sv[j], VR_LIGHT_DIRECTIONAL, t.vec, 0.0f, t.color, 1.0f, t.w, t.inv_range, tanH, dT[j], sT[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < PXB; j++)
                        if (depth[kb + j] < 1.0f) add_light(sv[j], VR_LIGHT_POINT, t.vec, t.inv_range, t.color, 1.0f, 1.0f, 0.0f, 0.0f, dT[j], sT[j], nullptr, true);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PXB; j++) {
            const int k = kb + j;
            float rgb[3];
            finish_pixel(a, sv[j], dT[j], sT[j], rgb);
            o[2 * k] = vr_float_to_half(rgb[0]) | (vr_float_to_half(rgb[1]) << 16);
            o[2 * k + 1] = vr_float_to_half(rgb[2]);
        }
    }
    if (inside) store_quad<PACKED, NT>(out, out_index, o);
}

extern "C" VR_API int vr_deferred_light_tiled(vr_context* ctx, const vr_view* view, vr_gbuffer* gb, const vr_light* lights,
                                               int32_t num_lights, const float amb_top[3], const float amb_bottom[3],
                                               vr_image* hdr, const vr_partition* part)
{
    VR_REQUIRE(ctx && view && gb && hdr && amb_top && amb_bottom, "NULL argument");
    VR_REQUIRE(num_lights >= 0 && num_lights <= 65536 && (num_lights == 0 || lights), "bad light list");
    VR_REQUIRE(view->viewport_w == gb->w && view->viewport_h == gb->h && view->viewport_x == 0 && view->viewport_y == 0,
               "view viewport must cover the G-buffer");
    VR_HIP(hipSetDevice(ctx->device));
    if ((size_t)num_lights > ctx->light_capacity) {
        VR_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_lights); ctx->d_lights = nullptr; ctx->light_capacity = 0; ctx->h_lights_on_device.clear();
        const size_t cap = (size_t)num_lights < 1024 ? 1024 : (size_t)num_lights;
        VR_HIP(hipMalloc(&ctx->d_lights, cap * sizeof(DevLight)));
        ctx->light_capacity = cap;
    }
    if (!ctx->d_flags) { VR_HIP(hipMalloc(&ctx->d_flags, 64)); VR_HIP(hipMemsetAsync(ctx->d_flags, 0, 64, ctx->stream)); }
    PlaneHints hints;                                         // (a reader: a pending clear happens now)
    { const int rc = vr_gbuffer_plane_hints(gb, ctx->stream, &hints); if (rc) return rc; }
    ctx->h_lights.resize((size_t)num_lights);
    for (int i = 0; i < num_lights; i++) { int rc = fill_light(lights[i], ctx->h_lights[i], false); if (rc) return rc; }
    // a scene's light list rarely changes between frames: upload only when it differs from what the device holds
    const size_t light_bytes = (size_t)num_lights * sizeof(DevLight);
    if (num_lights && (ctx->h_lights_on_device.size() != (size_t)num_lights
                       || memcmp(ctx->h_lights_on_device.data(), ctx->h_lights.data(), light_bytes) != 0)) {
        ctx->h_lights_on_device.clear();
        VR_HIP(hipMemcpyAsync(ctx->d_lights, ctx->h_lights.data(), light_bytes, hipMemcpyHostToDevice, ctx->stream));
        ctx->h_lights_on_device = ctx->h_lights;
    }
    DeferredArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 16; i++) a.c2w[i] = view->clip_to_world[i];
    for (int i = 0; i < 3; i++) { a.cam[i] = view->camera_pos[i]; a.amb_top[i] = amb_top[i]; a.amb_bot[i] = amb_bottom[i]; }
    a.w = gb->w; a.h = gb->h; a.sx = 2.0f / (float)gb->w; a.sy = -2.0f / (float)gb->h; a.num_lights = 0;
    for (int i = 0; i < num_lights; i++) if (ctx->h_lights[i].type != VR_LIGHT_DIRECTIONAL) a.exact_pos = 1;
    const bool packed = part != nullptr;
    VR_REQUIRE(gb->w % 4 == 0, "the tiled pass needs a frame width that is a multiple of 4");
    // light lists: one per 32x32 light tile, count + up to min(num_lights, VR_TILE_LIGHT_CAP) indices
    const int macro_x = (gb->w + kMacroTile - 1) / kMacroTile, macro_y = (gb->h + kMacroTile - 1) / kMacroTile;
    const int tx = (gb->w + kLightTile - 1) / kLightTile, ty = (gb->h + kLightTile - 1) / kLightTile;
    const int stride = (num_lights < kTileLightCap ? num_lights : kTileLightCap) + 1;
    const size_t words = (size_t)tx * ty * stride;
    if (words > ctx->light_list_words) {
        VR_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_light_lists); ctx->d_light_lists = nullptr; ctx->light_list_words = 0;
        VR_HIP(hipMalloc(&ctx->d_light_lists, words * sizeof(uint32_t)));
        ctx->light_list_words = words;
    }
    // per macro tile: the lights that touch its box, in four regions (one per wave of k_light_cull)
    const size_t scratch_words = (size_t)macro_x * macro_y * (size_t)(((num_lights + 3) / 4) * 4) + 4;
    if (scratch_words > ctx->macro_scratch_words) {
        VR_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_macro_scratch); ctx->d_macro_scratch = nullptr; ctx->macro_scratch_words = 0;
        VR_HIP(hipMalloc(&ctx->d_macro_scratch, scratch_words * sizeof(uint32_t)));
        ctx->macro_scratch_words = scratch_words;
    }
    // The light tiles' depth ranges, if the tile pass that filled this G-buffer left them (for the same split) and nothing has
    // written to it since.  The culling stage consumes them (every entry it reads is reset), hence CLEAN afterwards - for a
    // split, only if this rank's share is non-empty (else nothing was written either).
    const bool use_ranges = gb->ranges_state == vr_gbuffer::RANGES_VALID && gb->d_ranges && gb->ranges_world == (part ? part->world_size : 1)
                         && gb->ranges_rank == (part ? part->rank : 0);
    if (use_ranges) gb->ranges_state = vr_gbuffer::RANGES_CLEAN;
    // two launches, each timed under its own id; the shading kernel's events are stamped by its dispatch like the streaming
    // pass's, so its stop event serves as the next frame's geometry start hint (vr_terrain_prepare)
    if (packed) {
        const PartTables* pt = nullptr;
        int rc = vr_partition_tables(ctx, gb->w, gb->h, part, &pt);
        if (rc) return rc;
        VR_REQUIRE((size_t)pt->max_owned * VR_OWNER_TILE * VR_OWNER_TILE * 6 <= hdr->capacity_bytes, "hdr_out is smaller than vr_partition_packed_bytes()");
        a.tiles_x = (gb->w + VR_OWNER_TILE - 1) / VR_OWNER_TILE;
        if (pt->num_owned > 0) {
            { VrKernelScope kc(ctx, VR_K_LIGHT_CULL);
              hipLaunchKernelGGL(use_ranges ? k_light_cull<true> : k_light_cull<false>, dim3((unsigned)pt->num_owned), dim3(256), 0, ctx->stream, a, ctx->d_lights,
                                 num_lights, gb->depth, macro_x, pt->d_owned_tiles, ctx->d_light_lists, stride, tx, ty, ctx->d_flags, ctx->d_macro_scratch,
                                 gb->d_ranges); }
            VrKernelScope ks(ctx, VR_K_DEFERRED_TILED, ctx->stream, true);
            VR_LAUNCH_TIMED(ks, (k_deferred_tiled<true, VR_TILED_PXB>), dim3((unsigned)pt->num_owned * kSubTiles), dim3(256), ctx->stream, a,
                            ctx->d_lights, gb->depth, gb->diffuse, gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data,
                            ctx->d_srgb_lut, pt->d_owned_tiles, ctx->d_light_lists, stride, tx, hints);
        }
    } else {
        VR_REQUIRE((size_t)gb->w * gb->h * 8 <= hdr->capacity_bytes, "hdr_out is smaller than the frame");
        { VrKernelScope kc(ctx, VR_K_LIGHT_CULL);
          hipLaunchKernelGGL(use_ranges ? k_light_cull<true> : k_light_cull<false>, dim3((unsigned)(macro_x * macro_y)), dim3(256), 0, ctx->stream, a, ctx->d_lights,
                             num_lights, gb->depth, macro_x, (const int32_t*)nullptr, ctx->d_light_lists, stride, tx, ty, ctx->d_flags, ctx->d_macro_scratch,
                             gb->d_ranges); }
        VrKernelScope ks(ctx, VR_K_DEFERRED_TILED, ctx->stream, true);
        const bool nt = true;                                         // as in the streaming pass
        if (nt) VR_LAUNCH_TIMED(ks, (k_deferred_tiled<false, VR_TILED_PXB, true>), dim3((unsigned)(tx * ty)), dim3(256), ctx->stream, a, ctx->d_lights,
                                gb->depth, gb->diffuse, gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data, ctx->d_srgb_lut,
                                (const int32_t*)nullptr, ctx->d_light_lists, stride, tx, hints);
        else VR_LAUNCH_TIMED(ks, (k_deferred_tiled<false, VR_TILED_PXB, false>), dim3((unsigned)(tx * ty)), dim3(256), ctx->stream, a, ctx->d_lights,
                             gb->depth, gb->diffuse, gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data, ctx->d_srgb_lut,
                             (const int32_t*)nullptr, ctx->d_light_lists, stride, tx, hints);
    }
    VR_HIP(hipGetLastError());
    return VR_OK;
}

extern "C" VR_API int vr_deferred_tiled_status(vr_context* ctx)
{
    VR_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (!ctx->d_flags) return VR_OK;                         // no tiled pass has run on this context
    VR_HIP(hipSetDevice(ctx->device));
    uint32_t flags = 0;
    VR_HIP(hipMemcpyAsync(&flags, ctx->d_flags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
    VR_HIP(hipMemsetAsync(ctx->d_flags, 0, sizeof(flags), ctx->stream));
    VR_HIP(hipStreamSynchronize(ctx->stream));
    if (flags & 1u) {
        vr_set_error("vr_deferred_light_tiled: a 32x32 tile kept more than %d lights; the excess was dropped", kTileLightCap);
        return VR_ERR_OVERFLOW;
    }
    return VR_OK;
}

// ---- frame assembly after the all-gather ------------------------------------
// gathered = world_size packed RGB16F buffers back to back; one lane expands 2 pixels (12 B -> 16 B).
__global__ __launch_bounds__(256) void k_detile(const uint32_t* __restrict__ gathered, uint4* __restrict__ frame, int w, int h,
                                                 int tiles_x, const int32_t* __restrict__ tile_slot)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;     // pixel pair index
    const size_t p = i * 2;
    if (p >= (size_t)w * h) return;
    const int y = (int)(p / (size_t)w), x = (int)(p - (size_t)y * w);
    const int tx = x / VR_OWNER_TILE, ty = y / VR_OWNER_TILE;
    const int slot = tile_slot[ty * tiles_x + tx];
    const size_t src = ((size_t)slot * VR_OWNER_TILE + (y - ty * VR_OWNER_TILE)) * VR_OWNER_TILE + (x - tx * VR_OWNER_TILE);
    const uint32_t* g = gathered + src * 3 / 2;                    // 6 B per pixel, src is even
    const uint32_t d0 = g[0], d1 = g[1], d2 = g[2];                // r0 g0 | b0 r1 | g1 b1
    frame[i] = make_uint4(d0, d1 & 0xffffu, (d1 >> 16) | (d2 << 16), d2 >> 16);
}

extern "C" VR_API int vr_frame_detile(vr_context* ctx, const void* gathered, int32_t world, vr_image* frame)
{
    VR_REQUIRE(ctx && gathered && frame, "NULL argument");
    VR_REQUIRE(frame->w % 2 == 0 && world >= 1, "frame width must be even");
    VR_HIP(hipSetDevice(ctx->device));
    const PartTables* pt = nullptr;
    { int rc = vr_partition_slot_tables(ctx, frame->w, frame->h, world, &pt); if (rc) return rc; }
    const size_t pairs = (size_t)frame->w * frame->h / 2;
    const int tiles_x = (frame->w + VR_OWNER_TILE - 1) / VR_OWNER_TILE;
    VrKernelScope ks(ctx, VR_K_DETILE);
    hipLaunchKernelGGL(k_detile, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)gathered,
                       (uint4*)frame->data, frame->w, frame->h, tiles_x, pt->d_tile_slot);
    VR_HIP(hipGetLastError());
    return VR_OK;
}
