// Host side of libvrterrain.so: context, render targets, view helper, partition tables.
#include "vr_internal.h"
#include "vr_experiments.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

// ---- errors ---------------------------------------------------------------------
static thread_local char g_last_error[512] = "";
void vr_set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}
extern "C" VR_API const char* vr_last_error(void) { return g_last_error; }
extern "C" VR_API const char* vr_version(void) { return "vrterrain 0.1 (gfx950)"; }
extern "C" VR_API uint32_t vr_build_experiments(void) { return kExpMask; }

// ---- sRGB tables (SRGBA8 fetch / render-target conversion) -------------------------
static double srgb_eotf(double c) { return c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4); }

static void timing_reset(vr_context* c);

// ---- context ----------------------------------------------------------------------
extern "C" VR_API int vr_context_create(int device, vr_context** out)
{
    VR_REQUIRE(out != nullptr, "out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { vr_set_error("no HIP device available (%s)", hipGetErrorString(e)); return VR_ERR_NO_DEVICE; }
    VR_REQUIRE(device >= 0 && device < n, "device ordinal out of range");
    VR_HIP(hipSetDevice(device));
    vr_context* c = new vr_context();
    struct Guard { vr_context*& c; ~Guard() { if (c) vr_context_destroy(c); } } guard{ c };   // frees what exists on an early return
    c->device = device;
    c->stream = nullptr;   // default stream until vr_context_set_stream
    for (int i = 0; i < 256; i++) c->h_srgb_lut[i] = (float)srgb_eotf((double)i / 255.0);
    c->h_srgb_thr[0] = 0.0f;
    for (int k = 1; k < 256; k++) c->h_srgb_thr[k] = (float)srgb_eotf(((double)k - 0.5) / 255.0);
    VR_HIP(hipMalloc(&c->d_srgb_lut, 256 * sizeof(float)));
    VR_HIP(hipMalloc(&c->d_srgb_thr, 257 * sizeof(float)));          // [256] = NaN: no x is >= it (readers that index thr[code + 1])
    VR_HIP(hipMemcpy(c->d_srgb_lut, c->h_srgb_lut, 256 * sizeof(float), hipMemcpyHostToDevice));
    VR_HIP(hipMemcpy(c->d_srgb_thr, c->h_srgb_thr, 256 * sizeof(float), hipMemcpyHostToDevice));
    { const uint32_t nan_bits = 0x7fc00000u; VR_HIP(hipMemcpy(c->d_srgb_thr + 256, &nan_bits, sizeof(nan_bits), hipMemcpyHostToDevice)); }
    {   // start table for the device encoder: code of the smallest float in each bucket
        uint8_t tab[kEncTabSize];
        for (int b = 0; b < kEncTabSize; b++) {
            const uint32_t bits = (uint32_t)(b + kEncTabBase) << 16;
            float x; memcpy(&x, &bits, 4);
            int lo = 0, hi = 255;
            while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (x >= c->h_srgb_thr[mid]) lo = mid; else hi = mid - 1; }
            tab[b] = (uint8_t)lo;
            // the device encoder adds at most one to tab[b]: no bucket may hold two thresholds
            const uint32_t top = bits | 0xffffu;
            float xt; memcpy(&xt, &top, 4);
            int hi_code = 0;
            for (int k = 255; k > 0; k--) if (xt >= c->h_srgb_thr[k]) { hi_code = k; break; }
            if (b < kEncTabSize - 1 && hi_code - lo > 1) {
                vr_set_error("sRGB encode table: bucket %d spans more than one threshold", b);
                return VR_ERR_INVALID_ARGUMENT;
            }
        }
        VR_HIP(hipMalloc(&c->d_enc_tab, (kEncTabSize + 3) / 4 * 4));          // read as dwords by the kernels
        VR_HIP(hipMemset(c->d_enc_tab, 0, (kEncTabSize + 3) / 4 * 4));
        VR_HIP(hipMemcpy(c->d_enc_tab, tab, kEncTabSize, hipMemcpyHostToDevice));
    }
    *out = c;
    c = nullptr;                 // released to the caller: the guard lets go
    return VR_OK;
}

extern "C" VR_API void vr_context_destroy(vr_context* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipFree(c->d_srgb_lut); (void)hipFree(c->d_srgb_thr); (void)hipFree(c->d_enc_tab);
    (void)hipFree(c->d_lights); (void)hipFree(c->d_flags); (void)hipFree(c->d_light_lists); (void)hipFree(c->d_macro_scratch);
    for (PartTables* pt : c->part_tables) {
        (void)hipFree(pt->d_owned_tiles); (void)hipFree(pt->d_tile_slot); (void)hipFree(pt->d_raster_tiles);
        delete pt;
    }
    timing_reset(c);
    if (c->ev_sysfence) (void)hipEventDestroy(c->ev_sysfence);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_ring) (void)hipEventDestroy(e);
    delete c;
}

extern "C" VR_API int vr_context_set_stream(vr_context* c, void* s)
{
    VR_REQUIRE(c != nullptr, "ctx is NULL");
    c->stream = (hipStream_t)s;
    return VR_OK;
}

extern "C" VR_API int vr_context_set_option(vr_context* c, int option, int value)
{
    VR_REQUIRE(c != nullptr, "ctx is NULL");
    VR_REQUIRE(option == VR_OPT_ASYNC_GEOMETRY || option == VR_OPT_DISPATCH_EVENTS || option == VR_OPT_RASTER_TILE || option == VR_OPT_PLANE_TRACKING || option == VR_OPT_SCRATCH_WORST_CASE, "unknown option");
    if (option == VR_OPT_ASYNC_GEOMETRY) c->async_geometry = value != 0;
    else if (option == VR_OPT_PLANE_TRACKING) c->plane_tracking = value != 0;
    else if (option == VR_OPT_SCRATCH_WORST_CASE) c->scratch_worst_case = value != 0;
    else if (option == VR_OPT_RASTER_TILE) {
        VR_REQUIRE(value == 0 || value == 32 || value == 64, "VR_OPT_RASTER_TILE: 0 (by size), 32 or 64");
        c->raster_tile_force = value == 32 ? 5 : value == 64 ? 6 : 0;
    }
    else { VR_HIP(hipStreamSynchronize(c->stream)); c->dispatch_events = value != 0; c->last_stop = nullptr; }
    return VR_OK;
}

extern "C" VR_API int vr_context_synchronize(vr_context* c)
{
    VR_REQUIRE(c != nullptr, "ctx is NULL");
    VR_HIP(hipStreamSynchronize(c->stream));
    return VR_OK;
}

// ---- per-kernel timing ---------------------------------------------------------------
// Events that order device work against device work (and time kernels) need no system-scope fence: nothing here hands data
// to the host through them - downloads go through stream synchronisation.  Without the flag every stamped event ends its
// kernel with a release to system scope (a write-back of the caches) in front of the next dispatch.
static unsigned vr_event_flags()
{
    static const unsigned flags = getenv("VR_EVENT_SYSTEM_FENCE") ? hipEventDefault : hipEventDisableSystemFence;
    return flags;
}
static hipEvent_t take_event(vr_context* c)
{
    hipEvent_t e = nullptr;
    if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    if (hipEventCreateWithFlags(&e, vr_event_flags()) != hipSuccess) return nullptr;
    return e;
}
VrKernelScope::VrKernelScope(vr_context* ctx, int id) : VrKernelScope(ctx, id, ctx->stream) {}
VrKernelScope::VrKernelScope(vr_context* ctx, int id, hipStream_t stream) : VrKernelScope(ctx, id, stream, false) {}
static hipEvent_t ring_event(vr_context* c)
{
    constexpr size_t kRing = 64;          // far more than the launches a later wait can still refer to
    if (c->ev_ring.size() < kRing) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, vr_event_flags()) != hipSuccess) return nullptr;
        c->ev_ring.push_back(e);
        return e;
    }
    hipEvent_t e = c->ev_ring[c->ev_ring_pos];
    c->ev_ring_pos = (c->ev_ring_pos + 1) % kRing;
    return e;
}
VrKernelScope::VrKernelScope(vr_context* ctx, int id_, hipStream_t stream, bool attach_) : c(ctx), st(stream), attach(attach_), id(id_)
{
    // level 2: only the launches whose events are stamped by the dispatch itself (the tile pass, the lighting passes): timing
    // them costs the host nothing, while two hipEventRecord calls around each of a frame's dozen small kernels make a loop
    // with a ~110 us frame period host-bound
    if (c->timing == 1 || (c->timing == 2 && attach)) {
        e0 = take_event(c); e1 = take_event(c);
        if (!e0 || !e1) {
            if (e0) c->ev_pool.push_back(e0);
            if (e1) c->ev_pool.push_back(e1);
            e0 = e1 = nullptr; return;
        }
        pooled = true;
        if (!attach) { (void)hipEventRecord(e0, st); commit(); }
    } else if (attach && c->dispatch_events) {
        e0 = ring_event(c); e1 = ring_event(c);
        if (!e0 || !e1) e0 = e1 = nullptr;
    }
}
void VrKernelScope::commit()
{
    if (committed || !pooled || !e0 || !e1) return;
    c->ev_begin.push_back(e0); c->ev_end.push_back(e1); c->ev_id.push_back(id);
    committed = true;
}
void VrKernelScope::launched()
{
    commit();
    if (st == c->stream) c->last_stop = e1;
}
VrKernelScope::~VrKernelScope()
{
    if (!e0 || !e1 || !pooled) return;
    if (!attach) (void)hipEventRecord(e1, st);
    else if (!committed) { c->ev_pool.push_back(e0); c->ev_pool.push_back(e1); }
}

static void timing_reset(vr_context* c)
{
    c->ev_epoch++;                     // every handle to a pooled event taken before this point is stale now
    c->last_stop = nullptr;
    for (hipEvent_t e : c->ev_begin) c->ev_pool.push_back(e);
    for (hipEvent_t e : c->ev_end) c->ev_pool.push_back(e);
    c->ev_begin.clear(); c->ev_end.clear(); c->ev_id.clear();
}
extern "C" VR_API int vr_timing_enable(vr_context* c, int enable)
{
    VR_REQUIRE(c != nullptr, "ctx is NULL");
    VR_HIP(hipSetDevice(c->device));
    VR_HIP(hipStreamSynchronize(c->stream));
    // pairs recorded on a terrain's geometry stream may still be pending: wait for each before its events are recycled
    for (hipEvent_t e : c->ev_end) (void)hipEventSynchronize(e);
    timing_reset(c);
    c->timing = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    // events for the launches to come are made HERE, not one by one inside the host's frame loop (hipEventCreate per stamped
    // launch was ~20 us of a rank's ~100 us of host time per frame): enough for a few hundred frames, more are made on demand
    if (c->timing) {
        const size_t want = c->timing == 1 ? 8192 : 2048;
        while (c->ev_pool.size() < want) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, vr_event_flags()) != hipSuccess) break;
            c->ev_pool.push_back(e);
        }
    }
    return VR_OK;
}
extern "C" VR_API int vr_timing_collect(vr_context* c, float ms_sum[VR_K_COUNT], int32_t launches[VR_K_COUNT])
{
    VR_REQUIRE(c && ms_sum && launches, "NULL argument");
    VR_HIP(hipSetDevice(c->device));
    VR_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < VR_K_COUNT; i++) { ms_sum[i] = 0.0f; launches[i] = 0; }
    int rc = VR_OK;
    for (size_t i = 0; i < c->ev_id.size(); i++) {
        float ms = 0.0f;
        hipError_t e = hipEventSynchronize(c->ev_end[i]);      // may have been recorded on a terrain's geometry stream
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_begin[i], c->ev_end[i]);
        if (e != hipSuccess) { vr_set_error("vr_timing_collect: %s", hipGetErrorString(e)); rc = VR_ERR_HIP; continue; }
        ms_sum[c->ev_id[i]] += ms; launches[c->ev_id[i]]++;
    }
    timing_reset(c);                                     // also on the error path: a bad pair must not persist
    return rc;
}
extern "C" VR_API int vr_timing_kernel_count(void) { return VR_K_COUNT; }
extern "C" VR_API const char* vr_kernel_name(int id)
{
    static const char* names[VR_K_COUNT] = { "k_select", "k_vertex", "k_setup", "k_clip", "k_scan", "k_fill", "k_raster",
                                              "k_deferred", "k_detile", "k_fill_u32 (clear)", "k_deferred_tiled", "k_node_heights (all levels)",
                                              "k_tm_histogram", "k_tm_exposure", "k_tonemap", "k_detile_ldr", "k_raster (depth only)", "k_light_cull",
                                              "k_raster (fused with lighting)" };
    return (id >= 0 && id < VR_K_COUNT) ? names[id] : "?";
}

// ---- defaults (TerrainPass.h:23-30, QuadTree.cpp:236, terrain_vs.hlsl:20, Renderer.h:40) ----
extern "C" VR_API void vr_terrain_default_params(vr_terrain_params* p)
{
    memset(p, 0, sizeof(*p));
    p->max_instances = 4096; p->surface_size = 2048.0f; p->world_size = 2048.0f; p->grid_size = 32;
    p->min_lod_distance = 4.0f; p->morph_start = 0.85f;
}
extern "C" VR_API void vr_render_default_params(vr_render_params* p)
{
    memset(p, 0, sizeof(*p));
    p->max_height = 400.0f;
}

// ---- view helper -------------------------------------------------------------------
// What the caller's Donut objects produce (source absent from the reference checkout,
// restated): FirstPersonCamera::LookAt (Renderer.cpp:97), perspProjD3DStyle
// (Renderer.cpp:315), PlanarView::UpdateCache -> view frustum (TerrainPass.cpp:181).
namespace {
struct V3 { float x, y, z; };
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline V3 normalized(V3 v) { float l = sqrtf(dot(v, v)); if (l > 0.0f) { v.x /= l; v.y /= l; v.z /= l; } return v; }

void plane_from(float out[4], float x, float y, float z, float d)
{
    float l2 = dot({ x, y, z }, { x, y, z });
    float s = l2 > 0.0f ? 1.0f / sqrtf(l2) : 0.0f;
    out[0] = x * s; out[1] = y * s; out[2] = z * s; out[3] = d * s;
}

bool invert4x4(const float m[16], float out[16])
{
    double a[4][8];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = m[i * 4 + j]; a[i][j + 4] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; c++) {
        int piv = c; double best = fabs(a[c][c]);
        for (int r = c + 1; r < 4; r++) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
        if (best == 0.0) return false;
        if (piv != c) for (int j = 0; j < 8; j++) { double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
        double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; j++) a[c][j] *= inv;
        for (int r = 0; r < 4; r++) if (r != c) { double f = a[r][c]; if (f != 0.0) for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[i * 4 + j] = (float)a[i][j + 4];
    return true;
}
} // namespace

extern "C" VR_API int vr_view_from_camera(const float eye[3], const float target[3], const float up_in[3],
                                           float vfov, float z_near, float z_far, int32_t w, int32_t h, vr_view* o)
{
    VR_REQUIRE(eye && target && up_in && o, "NULL argument");
    VR_REQUIRE(w > 0 && h > 0 && z_far > z_near && z_near > 0.0f, "bad projection parameters");
    memset(o, 0, sizeof(*o));
    V3 dir = normalized({ target[0] - eye[0], target[1] - eye[1], target[2] - eye[2] });
    V3 up = normalized({ up_in[0], up_in[1], up_in[2] });
    V3 right = normalized(cross(dir, up));
    up = normalized(cross(right, dir));
    const V3 axes[3] = { right, up, dir };
    float* m = o->world_to_view;
    const float r3[3] = { right.x, right.y, right.z }, u3[3] = { up.x, up.y, up.z }, d3[3] = { dir.x, dir.y, dir.z };
    for (int i = 0; i < 3; i++) { m[i * 4 + 0] = r3[i]; m[i * 4 + 1] = u3[i]; m[i * 4 + 2] = d3[i]; m[i * 4 + 3] = 0.0f; }
    V3 ne = { -eye[0], -eye[1], -eye[2] };
    for (int j = 0; j < 3; j++) m[12 + j] = dot(ne, axes[j]);
    m[15] = 1.0f;
    float ys = 1.0f / tanf(0.5f * vfov), xs = ys / ((float)w / (float)h), zs = 1.0f / (z_far - z_near);
    float* p = o->view_to_clip;
    p[0] = xs; p[5] = ys; p[10] = z_far * zs; p[11] = 1.0f; p[14] = -z_near * z_far * zs;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
        o->world_to_clip[i * 4 + j] = ((m[i * 4 + 0] * p[0 * 4 + j] + m[i * 4 + 1] * p[1 * 4 + j]) + m[i * 4 + 2] * p[2 * 4 + j]) + m[i * 4 + 3] * p[3 * 4 + j];
    if (!invert4x4(o->world_to_clip, o->clip_to_world)) { vr_set_error("singular view-projection"); return VR_ERR_INVALID_ARGUMENT; }
    o->camera_pos[0] = eye[0]; o->camera_pos[1] = eye[1]; o->camera_pos[2] = eye[2]; o->camera_pos[3] = 1.0f;
    const float* c = o->world_to_clip;
    plane_from(o->planes[0], -c[2], -c[6], -c[10], c[14]);
    plane_from(o->planes[1], -c[3] + c[2], -c[7] + c[6], -c[11] + c[10], c[15] - c[14]);
    plane_from(o->planes[2], -c[3] - c[0], -c[7] - c[4], -c[11] - c[8], c[15] + c[12]);
    plane_from(o->planes[3], -c[3] + c[0], -c[7] + c[4], -c[11] + c[8], c[15] - c[12]);
    plane_from(o->planes[4], -c[3] + c[1], -c[7] + c[5], -c[11] + c[9], c[15] - c[13]);
    plane_from(o->planes[5], -c[3] - c[1], -c[7] - c[5], -c[11] - c[9], c[15] + c[13]);
    o->viewport_x = 0; o->viewport_y = 0; o->viewport_w = w; o->viewport_h = h;
    float det = right.x * (up.y * dir.z - up.z * dir.y) - up.x * (right.y * dir.z - right.z * dir.y)
              + dir.x * (right.y * up.z - right.z * up.y);
    o->mirrored = det < 0.0f ? 1 : 0;
    o->reverse_depth = 0;
    return VR_OK;
}

// ---- shadow view (row f1) ---------------------------------------------------------------
extern "C" VR_API void vr_shadow_default_params(vr_shadow_params* p, float world_size)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->resolution = 2048;                                            // Renderer.cpp:83
    p->max_shadow_distance = world_size;                             // :349
    p->light_space_z_up = world_size; p->light_space_z_down = world_size;   // :350-352
    p->depth_bias = 0.0f;
}

// CascadedShadowMap::SetupForPlanarViewStable(light, projectionFrustum, inverseViewMatrix, maxShadowDistance,
// zUp, zDown, 4.0f, 0, 1) (Renderer.cpp:345-352) [DONUT-RECOLLECTION]: with one cascade the split exponent has
// nothing to split; the cascade is the bounding sphere of the camera frustum slice [0, maxShadowDistance],
// snapped to whole shadow texels in light space, under an orthographic D3D projection.
extern "C" VR_API int vr_shadow_view_setup(const vr_light* light, const vr_view* cam, const vr_shadow_params* p, vr_view* o)
{
    VR_REQUIRE(light && cam && p && o, "NULL argument");
    VR_REQUIRE(light->type == VR_LIGHT_DIRECTIONAL, "the cascaded shadow map belongs to a directional light (Renderer.cpp:336)");
    VR_REQUIRE(p->resolution >= 16 && p->resolution <= 16384 && p->max_shadow_distance > 0.0f
               && p->light_space_z_up + p->light_space_z_down > 0.0f, "bad shadow parameters");
    VR_REQUIRE(cam->view_to_clip[0] != 0.0f && cam->view_to_clip[5] != 0.0f, "camera projection is degenerate");
    memset(o, 0, sizeof(*o));
    const float* W = cam->world_to_view;
    const V3 fwd = { W[2], W[6], W[10] };
    const float d = p->max_shadow_distance;
    const float hw = d / cam->view_to_clip[0], hh = d / cam->view_to_clip[5];
    const float r2 = hw * hw + hh * hh;
    float c = (d * d + r2) / (2.0f * d);
    if (c > d) c = d;
    const float radius = sqrtf((d - c) * (d - c) + r2);
    const V3 centre = { cam->camera_pos[0] + fwd.x * c, cam->camera_pos[1] + fwd.y * c, cam->camera_pos[2] + fwd.z * c };
    const V3 zl = normalized({ light->direction[0], light->direction[1], light->direction[2] });
    VR_REQUIRE(dot(zl, zl) > 0.0f, "light direction is zero");
    V3 up = { 0.0f, 1.0f, 0.0f };
    if (fabsf(zl.y) > 0.99f) up = { 0.0f, 0.0f, 1.0f };
    const V3 xl = normalized(cross(up, zl));
    const V3 yl = cross(zl, xl);
    const float texel = (2.0f * radius) / (float)p->resolution;
    const float cx = floorf(dot(centre, xl) / texel) * texel, cy = floorf(dot(centre, yl) / texel) * texel;
    const float cz = dot(centre, zl) - p->light_space_z_up;
    const V3 origin = { (xl.x * cx + yl.x * cy) + zl.x * cz, (xl.y * cx + yl.y * cy) + zl.y * cz, (xl.z * cx + yl.z * cy) + zl.z * cz };
    float* m = o->world_to_view;
    const float x3[3] = { xl.x, xl.y, xl.z }, y3[3] = { yl.x, yl.y, yl.z }, z3[3] = { zl.x, zl.y, zl.z };
    for (int i = 0; i < 3; i++) { m[i * 4 + 0] = x3[i]; m[i * 4 + 1] = y3[i]; m[i * 4 + 2] = z3[i]; m[i * 4 + 3] = 0.0f; }
    const V3 no = { -origin.x, -origin.y, -origin.z };
    m[12] = dot(no, xl); m[13] = dot(no, yl); m[14] = dot(no, zl); m[15] = 1.0f;
    float* q = o->view_to_clip;
    q[0] = 1.0f / radius; q[5] = 1.0f / radius; q[10] = 1.0f / (p->light_space_z_up + p->light_space_z_down); q[15] = 1.0f;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
        o->world_to_clip[i * 4 + j] = ((m[i * 4 + 0] * q[0 * 4 + j] + m[i * 4 + 1] * q[1 * 4 + j]) + m[i * 4 + 2] * q[2 * 4 + j]) + m[i * 4 + 3] * q[3 * 4 + j];
    if (!invert4x4(o->world_to_clip, o->clip_to_world)) { vr_set_error("singular shadow view-projection"); return VR_ERR_INVALID_ARGUMENT; }
    o->camera_pos[0] = origin.x; o->camera_pos[1] = origin.y; o->camera_pos[2] = origin.z; o->camera_pos[3] = 1.0f;
    const float* cc = o->world_to_clip;
    plane_from(o->planes[0], -cc[2], -cc[6], -cc[10], cc[14]);
    plane_from(o->planes[1], -cc[3] + cc[2], -cc[7] + cc[6], -cc[11] + cc[10], cc[15] - cc[14]);
    plane_from(o->planes[2], -cc[3] - cc[0], -cc[7] - cc[4], -cc[11] - cc[8], cc[15] + cc[12]);
    plane_from(o->planes[3], -cc[3] + cc[0], -cc[7] + cc[4], -cc[11] + cc[8], cc[15] - cc[12]);
    plane_from(o->planes[4], -cc[3] + cc[1], -cc[7] + cc[5], -cc[11] + cc[9], cc[15] - cc[13]);
    plane_from(o->planes[5], -cc[3] - cc[1], -cc[7] - cc[5], -cc[11] - cc[9], cc[15] + cc[13]);
    o->viewport_x = 0; o->viewport_y = 0; o->viewport_w = p->resolution; o->viewport_h = p->resolution;
    const float det = xl.x * (yl.y * zl.z - yl.z * zl.y) - yl.x * (xl.y * zl.z - xl.z * zl.y) + zl.x * (xl.y * yl.z - xl.z * yl.y);
    o->mirrored = det < 0.0f ? 1 : 0;
    o->reverse_depth = 0;
    return VR_OK;
}

// ---- G-buffer (RenderTargets::Init / Clear; Renderer.h:60-101, Renderer.cpp:382) --------
extern "C" VR_API int vr_gbuffer_create(vr_context* ctx, int32_t w, int32_t h, vr_gbuffer** out)
{
    VR_REQUIRE(ctx && out, "NULL argument");
    VR_REQUIRE(w > 0 && h > 0 && w <= 16384 && h <= 16384, "G-buffer size out of range");
    VR_HIP(hipSetDevice(ctx->device));
    vr_gbuffer* g = new vr_gbuffer();
    g->ctx = ctx; g->w = w; g->h = h;
    size_t n = (size_t)w * h;
    // one allocation, planes 256-byte aligned (28 B/pixel)
    size_t a = 256;
    size_t o_depth = 0, o_diff = (o_depth + n * 4 + a - 1) / a * a, o_spec = (o_diff + n * 4 + a - 1) / a * a;
    size_t o_nrm = (o_spec + n * 4 + a - 1) / a * a, o_emi = (o_nrm + n * 8 + a - 1) / a * a, total = o_emi + n * 8;
    uint8_t* base = nullptr;
    hipError_t e = hipMalloc(&base, total);
    if (e != hipSuccess) { delete g; vr_set_error("hipMalloc(%zu) failed: %s", total, hipGetErrorString(e)); return VR_ERR_OUT_OF_MEMORY; }
    g->depth = (float*)(base + o_depth); g->diffuse = (uint32_t*)(base + o_diff); g->specular = (uint32_t*)(base + o_spec);
    g->normals = (uint2*)(base + o_nrm); g->emissive = (uint2*)(base + o_emi);
    const int rc = vr_gbuffer_clear(g);
    if (rc != VR_OK) { (void)hipFree(base); delete g; return rc; }      // the caller owns the object only on success
    *out = g;
    return VR_OK;
}

extern "C" VR_API void vr_gbuffer_destroy(vr_gbuffer* g)
{
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    (void)hipFree(g->depth);   // base of the single allocation
    (void)hipFree(g->d_ranges);
    (void)hipFree(g->d_region);
    delete g;
}

__global__ void k_fill_u32x4(uint4* p, size_t n4, uint32_t v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 val = make_uint4(v, v, v, v);
    for (; i < n4; i += stride) p[i] = val;
}
__global__ void k_fill_u32_tail(uint32_t* p, size_t begin, size_t n, uint32_t v)
{
    size_t i = begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
static int fill_u32(hipStream_t s, void* ptr, size_t count, uint32_t v)
{
    size_t n4 = count / 4;
    if (n4) {
        size_t blocks = (n4 + 255) / 256; if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_fill_u32x4, dim3((unsigned)blocks), dim3(256), 0, s, (uint4*)ptr, n4, v);
    }
    if (count > n4 * 4) hipLaunchKernelGGL(k_fill_u32_tail, dim3(1), dim3(64), 0, s, (uint32_t*)ptr, n4 * 4, count, v);
    VR_HIP(hipGetLastError());
    return VR_OK;
}

static int gbuffer_clear_now(vr_gbuffer* g, hipStream_t s)
{
    vr_gbuffer_touch(g);
    size_t n = (size_t)g->w * g->h;
    int rc;
    VrKernelScope scope(g->ctx, VR_K_CLEAR, s);
    if ((rc = fill_u32(s, g->depth, n, 0x3f800000u))) return rc;      // depth = 1.0 (non-reversed)
    if ((rc = fill_u32(s, g->diffuse, n, 0u))) return rc;
    if ((rc = fill_u32(s, g->specular, n, 0u))) return rc;
    if ((rc = fill_u32(s, g->normals, n * 2, 0u))) return rc;
    if ((rc = fill_u32(s, g->emissive, n * 2, 0u))) return rc;
    g->emissive_zero = true;               // (stream-ordered: every later pass on the context's stream sees the zeros)
    g->region_fill = (int)kRegionClear;    // every region holds the clear values (the array is filled when a tile pass next asks for it)
    g->clear_pending = false;
    return VR_OK;
}

extern "C" VR_API int vr_gbuffer_clear(vr_gbuffer* g)
{
    VR_REQUIRE(g != nullptr, "gbuffer is NULL");
    VR_HIP(hipSetDevice(g->ctx->device));
    if (g->ctx->plane_tracking && !g->escaped && g->cleared_once) {
        // lazy (vr_internal.h): the planes are cleared by the next whole-frame tile pass, or by whoever looks at them first
        vr_gbuffer_touch(g);
        g->clear_pending = true;
        return VR_OK;
    }
    g->cleared_once = true;
    return gbuffer_clear_now(g, g->ctx->stream);
}

int vr_gbuffer_materialise(vr_gbuffer* g, hipStream_t s)
{
    if (!g->clear_pending) return VR_OK;
    return gbuffer_clear_now(g, s);
}

int vr_gbuffer_region_prepare(vr_gbuffer* g, hipStream_t s, uint8_t** out)
{
    const int tiles = ((g->w + 31) / 32) * ((g->h + 31) / 32);
    if (!g->d_region) {
        VR_HIP(hipMalloc(&g->d_region, (size_t)tiles * 4));
        g->region_tiles = tiles;
        if (g->region_fill < 0) g->region_fill = 0;
    }
    if (g->region_fill >= 0) {
        VR_HIP(hipMemsetAsync(g->d_region, g->region_fill, (size_t)tiles * 4, s));
        g->region_fill = -1;
    }
    *out = g->d_region;
    return VR_OK;
}

uint32_t vr_specular_constant(const vr_context* c)
{
    // sRGB-encode 1.0 * 0.01 with the context's threshold table (the tile pass's own encode)
    int lo = 0, hi = 255;
    const float x = 1.0f * 0.01f;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (x >= c->h_srgb_thr[mid]) lo = mid; else hi = mid - 1; }
    const uint32_t sc = (uint32_t)lo;
    return sc | (sc << 8) | (sc << 16) | 0xff000000u;
}

int vr_gbuffer_plane_hints(vr_gbuffer* g, hipStream_t s, PlaneHints* out)
{
    out->region = nullptr; out->spec_const = vr_specular_constant(g->ctx); out->emissive_zero = 0; out->tiles32_x = (g->w + 31) / 32;
    { const int rc = vr_gbuffer_materialise(g, s); if (rc) return rc; }       // (a reader: a pending clear happens now)
    if (!g->ctx->plane_tracking || g->escaped) return VR_OK;
    uint8_t* r = nullptr;
    const int rc = vr_gbuffer_region_prepare(g, s, &r);
    if (rc) return rc;
    out->region = r; out->emissive_zero = g->emissive_zero ? 1 : 0;
    return VR_OK;
}

extern "C" VR_API int vr_gbuffer_region_census(vr_gbuffer* g, uint32_t counts[4])
{
    VR_REQUIRE(g && counts, "NULL argument");
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    const int tiles = ((g->w + 31) / 32) * ((g->h + 31) / 32);
    counts[3] = (uint32_t)tiles * 4u;
    if (!g->ctx->plane_tracking || g->escaped) { counts[0] = counts[3]; return VR_OK; }      // no region is known to hold anything
    if (g->clear_pending) { counts[2] = counts[3]; return VR_OK; }            // (cleared, as far as anyone can tell)
    if (!g->d_region || g->region_fill >= 0) { counts[g->region_fill == (int)kRegionClear ? 2 : 0] = counts[3]; return VR_OK; }
    VR_HIP(hipSetDevice(g->ctx->device));
    std::vector<uint8_t> h((size_t)tiles * 4);
    VR_HIP(hipMemcpyAsync(h.data(), g->d_region, h.size(), hipMemcpyDeviceToHost, g->ctx->stream));
    VR_HIP(hipStreamSynchronize(g->ctx->stream));
    for (uint8_t b : h) counts[b <= 2 ? b : 0]++;
    return VR_OK;
}

extern "C" VR_API int vr_gbuffer_plane_known_zero(vr_gbuffer* g, int plane)
{
    if (!g || plane != 4) return 0;
    return (g->ctx->plane_tracking && (g->emissive_zero || g->clear_pending) && !g->escaped) ? 1 : 0;
}

__global__ void k_fill_u32x2(uint2* p, size_t n, uint32_t x, uint32_t y)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint2(x, y);
}
int vr_gbuffer_ranges_prepare(vr_gbuffer* g, hipStream_t s)
{
    const int tiles = ((g->w + 31) / 32) * ((g->h + 31) / 32);
    if (!g->d_ranges || g->ranges_tiles != tiles) {
        VR_HIP(hipStreamSynchronize(s));
        (void)hipFree(g->d_ranges); g->d_ranges = nullptr; g->ranges_state = vr_gbuffer::RANGES_NONE;
        VR_HIP(hipMalloc(&g->d_ranges, (size_t)tiles * sizeof(uint2)));
        g->ranges_tiles = tiles;
    }
    if (g->ranges_state != vr_gbuffer::RANGES_CLEAN) {
        hipLaunchKernelGGL(k_fill_u32x2, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, s, g->d_ranges, (size_t)tiles, 0x7f800000u, 0u);
        VR_HIP(hipGetLastError());
        g->ranges_state = vr_gbuffer::RANGES_CLEAN;
    }
    return VR_OK;
}

extern "C" VR_API int vr_gbuffer_describe(vr_gbuffer* g, vr_gbuffer_desc* d)
{
    VR_REQUIRE(g && d, "NULL argument");
    { VR_HIP(hipSetDevice(g->ctx->device)); const int rc = vr_gbuffer_materialise(g, g->ctx->stream); if (rc) return rc; }
    vr_gbuffer_touch(g);           // the caller gets the device pointers: whatever it writes through them is unknown here,
    g->escaped = true;             // now and for as long as the G-buffer lives (no depth ranges, no plane-state tracking any more)
    g->emissive_zero = false;
    g->region_fill = 0;
    d->width = g->w; d->height = g->h; d->depth = g->depth; d->diffuse = g->diffuse; d->specular = g->specular;
    d->normals = g->normals; d->emissive = g->emissive;
    return VR_OK;
}

static int plane_info(vr_gbuffer* g, int plane, void** ptr, size_t* bytes)
{
    size_t n = (size_t)g->w * g->h;
    switch (plane) {
    case 0: *ptr = g->depth; *bytes = n * 4; return VR_OK;
    case 1: *ptr = g->diffuse; *bytes = n * 4; return VR_OK;
    case 2: *ptr = g->specular; *bytes = n * 4; return VR_OK;
    case 3: *ptr = g->normals; *bytes = n * 8; return VR_OK;
    case 4: *ptr = g->emissive; *bytes = n * 8; return VR_OK;
    }
    vr_set_error("bad plane index %d", plane);
    return VR_ERR_INVALID_ARGUMENT;
}
extern "C" VR_API int vr_gbuffer_download(vr_gbuffer* g, int plane, void* host, size_t bytes)
{
    VR_REQUIRE(g && host, "NULL argument");
    void* p; size_t nb; int rc = plane_info(g, plane, &p, &nb); if (rc) return rc;
    VR_REQUIRE(bytes == nb, "byte count does not match the plane size");
    VR_HIP(hipSetDevice(g->ctx->device));
    if ((rc = vr_gbuffer_materialise(g, g->ctx->stream))) return rc;
    VR_HIP(hipMemcpyAsync(host, p, nb, hipMemcpyDeviceToHost, g->ctx->stream));
    VR_HIP(hipStreamSynchronize(g->ctx->stream));
    return VR_OK;
}
extern "C" VR_API int vr_gbuffer_upload(vr_gbuffer* g, int plane, const void* host, size_t bytes)
{
    VR_REQUIRE(g && host, "NULL argument");
    void* p; size_t nb; int rc = plane_info(g, plane, &p, &nb); if (rc) return rc;
    VR_REQUIRE(bytes == nb, "byte count does not match the plane size");
    VR_HIP(hipSetDevice(g->ctx->device));
    if ((rc = vr_gbuffer_materialise(g, g->ctx->stream))) return rc;
    vr_gbuffer_touch(g);
    if (plane == 4) g->emissive_zero = false;
    g->region_fill = 0;                    // (no region is known clear - that includes the emissive plane - or constant any more)
    VR_HIP(hipMemcpyAsync(p, host, nb, hipMemcpyHostToDevice, g->ctx->stream));
    VR_HIP(hipStreamSynchronize(g->ctx->stream));
    return VR_OK;
}

// ---- HDR image -------------------------------------------------------------------
extern "C" VR_API int vr_image_create(vr_context* ctx, int32_t w, int32_t h, void* external, vr_image** out)
{
    VR_REQUIRE(ctx && out, "NULL argument");
    VR_REQUIRE(w > 0 && h > 0, "bad image size");
    VR_HIP(hipSetDevice(ctx->device));
    vr_image* im = new vr_image();
    im->ctx = ctx; im->w = w; im->h = h; im->owned = (external == nullptr);
    im->capacity_bytes = (size_t)w * h * 8;
    if (external) im->data = external;
    else {
        hipError_t e = hipMalloc(&im->data, im->capacity_bytes);
        if (e != hipSuccess) { delete im; vr_set_error("hipMalloc(%zu) failed", (size_t)w * h * 8); return VR_ERR_OUT_OF_MEMORY; }
    }
    *out = im;
    return VR_OK;
}
extern "C" VR_API void vr_image_destroy(vr_image* im)
{
    if (!im) return;
    (void)hipSetDevice(im->ctx->device);
    if (im->read_pending) (void)hipEventSynchronize(im->ev_read_done);       // a stage on another stream may still be reading it
    if (im->ev_written) (void)hipEventDestroy(im->ev_written);
    if (im->ev_read_done) (void)hipEventDestroy(im->ev_read_done);
    if (im->owned) (void)hipFree(im->data);
    delete im;
}
extern "C" VR_API void* vr_image_device_ptr(vr_image* im) { return im ? im->data : nullptr; }
extern "C" VR_API int vr_image_download(vr_image* im, void* host, size_t bytes)
{
    VR_REQUIRE(im && host, "NULL argument");
    VR_REQUIRE(bytes <= im->capacity_bytes, "byte count exceeds the image");
    VR_HIP(hipSetDevice(im->ctx->device));
    VR_HIP(hipMemcpyAsync(host, im->data, bytes, hipMemcpyDeviceToHost, im->ctx->stream));
    VR_HIP(hipStreamSynchronize(im->ctx->stream));
    return VR_OK;
}

extern "C" VR_API int vr_image_upload(vr_image* im, const void* host, size_t bytes)
{
    VR_REQUIRE(im && host, "NULL argument");
    VR_REQUIRE(bytes <= im->capacity_bytes, "byte count exceeds the image");
    VR_HIP(hipSetDevice(im->ctx->device));
    VR_HIP(hipMemcpyAsync(im->data, host, bytes, hipMemcpyHostToDevice, im->ctx->stream));
    VR_HIP(hipStreamSynchronize(im->ctx->stream));
    return VR_OK;
}

// ---- LDR image (LdrColor, Renderer.h:81-92) ------------------------------------------------
extern "C" VR_API int vr_ldr_image_create(vr_context* ctx, int32_t w, int32_t h, size_t capacity_bytes, void* external, vr_ldr_image** out)
{
    VR_REQUIRE(ctx && out, "NULL argument");
    VR_REQUIRE(w > 0 && h > 0, "bad image size");
    VR_HIP(hipSetDevice(ctx->device));
    const size_t cap = capacity_bytes ? capacity_bytes : (size_t)w * h * 4;
    void* data = external;
    if (!external) {
        hipError_t e = hipMalloc(&data, cap);
        if (e != hipSuccess) { vr_set_error("hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e)); return VR_ERR_OUT_OF_MEMORY; }
    }
    vr_ldr_image* im = new vr_ldr_image();
    im->ctx = ctx; im->w = w; im->h = h; im->data = data; im->owned = external == nullptr; im->capacity_bytes = cap;
    *out = im;
    return VR_OK;
}
extern "C" VR_API void vr_ldr_image_destroy(vr_ldr_image* im)
{
    if (!im) return;
    if (im->owned) { (void)hipSetDevice(im->ctx->device); (void)hipFree(im->data); }
    delete im;
}
extern "C" VR_API void* vr_ldr_image_device_ptr(vr_ldr_image* im) { return im ? im->data : nullptr; }
extern "C" VR_API size_t vr_ldr_image_capacity(vr_ldr_image* im) { return im ? im->capacity_bytes : 0; }
extern "C" VR_API int vr_ldr_image_download(vr_ldr_image* im, void* host, size_t bytes)
{
    VR_REQUIRE(im && host, "NULL argument");
    VR_REQUIRE(bytes <= im->capacity_bytes, "byte count exceeds the image");
    VR_HIP(hipSetDevice(im->ctx->device));
    VR_HIP(hipMemcpyAsync(host, im->data, bytes, hipMemcpyDeviceToHost, im->ctx->stream));
    VR_HIP(hipStreamSynchronize(im->ctx->stream));
    return VR_OK;
}

extern "C" VR_API int vr_ldr_image_upload(vr_ldr_image* im, const void* host, size_t bytes)
{
    VR_REQUIRE(im && host, "NULL argument");
    VR_REQUIRE(bytes <= im->capacity_bytes, "byte count exceeds the image");
    VR_HIP(hipSetDevice(im->ctx->device));
    VR_HIP(hipMemcpyAsync(im->data, host, bytes, hipMemcpyHostToDevice, im->ctx->stream));
    VR_HIP(hipStreamSynchronize(im->ctx->stream));
    return VR_OK;
}

// ---- screen-tile partition ----------------------------------------------
static void owner_grid(int w, int h, int* tx, int* ty)
{
    *tx = (w + VR_OWNER_TILE - 1) / VR_OWNER_TILE; *ty = (h + VR_OWNER_TILE - 1) / VR_OWNER_TILE;
}
static int count_owned(int tx, int ty, int rank, int world)
{
    int n = 0;
    for (int y = 0; y < ty; y++) for (int x = 0; x < tx; x++) n += ((x + y) % world) == rank;
    return n;
}

extern "C" VR_API int vr_partition_num_tiles(int32_t w, int32_t h, const vr_partition* part, int32_t* tiles_x,
                                              int32_t* tiles_y, int32_t* owned, int32_t* max_owned)
{
    VR_REQUIRE(w > 0 && h > 0, "bad frame size");
    int world = part ? part->world_size : 1, rank = part ? part->rank : 0;
    VR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad partition");
    int tx, ty; owner_grid(w, h, &tx, &ty);
    if (tiles_x) *tiles_x = tx;
    if (tiles_y) *tiles_y = ty;
    if (owned) *owned = count_owned(tx, ty, rank, world);
    if (max_owned) { int m = 0; for (int r = 0; r < world; r++) { int c = count_owned(tx, ty, r, world); if (c > m) m = c; } *max_owned = m; }
    return VR_OK;
}

extern "C" VR_API size_t vr_partition_packed_bytes(int32_t w, int32_t h, int32_t world)
{
    if (w <= 0 || h <= 0 || world < 1) return 0;
    vr_partition p = { 0, world };
    int32_t mo = 0;
    vr_partition_num_tiles(w, h, &p, nullptr, nullptr, nullptr, &mo);
    return (size_t)mo * VR_OWNER_TILE * VR_OWNER_TILE * 6;      // RGB16F: the always-zero alpha is not exchanged
}

extern "C" VR_API int vr_partition_prepare(vr_context* ctx, int32_t w, int32_t h, const vr_partition* part)
{
    VR_REQUIRE(ctx && w > 0 && h > 0, "bad arguments");
    const PartTables* pt;
    return vr_partition_tables(ctx, w, h, part, &pt);
}

// most recently used last: the eviction below (the older half) can then never take a table that an API call in progress
// has just looked up
static const PartTables* part_tables_touch(vr_context* ctx, size_t i)
{
    PartTables* pt = ctx->part_tables[i];
    if (i + 1 != ctx->part_tables.size()) { ctx->part_tables.erase(ctx->part_tables.begin() + (long)i); ctx->part_tables.push_back(pt); }
    return pt;
}

int vr_partition_slot_tables(vr_context* ctx, int w, int h, int world, const PartTables** out)
{
    for (size_t i = 0; i < ctx->part_tables.size(); i++) {
        PartTables* pt = ctx->part_tables[i];
        if (pt->w == w && pt->h == h && pt->world == world) { *out = part_tables_touch(ctx, i); return VR_OK; }
    }
    const vr_partition p0 = { 0, world };
    return vr_partition_tables(ctx, w, h, &p0, out);
}

int vr_partition_tables(vr_context* ctx, int w, int h, const vr_partition* part, const PartTables** out)
{
    int world = part ? part->world_size : 1, rank = part ? part->rank : 0;
    VR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad partition");
    const int tile_shift = vr_raster_tile_shift(w, h, world, ctx->raster_tile_force);
    for (size_t i = 0; i < ctx->part_tables.size(); i++) {
        PartTables* pt = ctx->part_tables[i];
        if (pt->w == w && pt->h == h && pt->rank == rank && pt->world == world && pt->tile_shift == tile_shift) { *out = part_tables_touch(ctx, i); return VR_OK; }
    }
    VR_HIP(hipSetDevice(ctx->device));
    int tx, ty; owner_grid(w, h, &tx, &ty);
    int max_owned = 0;
    for (int r = 0; r < world; r++) { int c = count_owned(tx, ty, r, world); if (c > max_owned) max_owned = c; }
    std::vector<int32_t> owned, slot((size_t)tx * ty), raster;
    std::vector<int> next(world, 0);
    const int rtile = 1 << tile_shift;
    const int rtx = (w + rtile - 1) / rtile, rty = (h + rtile - 1) / rtile;
    const int sub = VR_OWNER_TILE / rtile;
    for (int y = 0; y < ty; y++) for (int x = 0; x < tx; x++) {
        int o = (x + y) % world;
        slot[(size_t)y * tx + x] = o * max_owned + next[o]++;
        if (o == rank) {
            owned.push_back(y * tx + x);
            for (int sy = 0; sy < sub; sy++) for (int sx = 0; sx < sub; sx++) {
                int rx = x * sub + sx, ry = y * sub + sy;
                if (rx < rtx && ry < rty) raster.push_back(ry * rtx + rx);
            }
        }
    }
    PartTables* pt = new PartTables();
    pt->w = w; pt->h = h; pt->rank = rank; pt->world = world; pt->tile_shift = tile_shift;
    auto fail = [&](hipError_t e, const char* what) {
        vr_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, what, hipGetErrorString(e));
        (void)hipFree(pt->d_owned_tiles); (void)hipFree(pt->d_tile_slot); (void)hipFree(pt->d_raster_tiles);
        delete pt;
        return e == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc(&pt->d_owned_tiles, sizeof(int32_t) * (owned.size() + 1))) != hipSuccess) return fail(e, "hipMalloc(owned tiles)");
    if ((e = hipMalloc(&pt->d_tile_slot, sizeof(int32_t) * slot.size())) != hipSuccess) return fail(e, "hipMalloc(tile slots)");
    if ((e = hipMalloc(&pt->d_raster_tiles, sizeof(int32_t) * (raster.size() + 1))) != hipSuccess) return fail(e, "hipMalloc(raster tiles)");
    // blocking copies from pageable memory: the tables are complete before any kernel that names them is queued
    if (!owned.empty() && (e = hipMemcpy(pt->d_owned_tiles, owned.data(), sizeof(int32_t) * owned.size(), hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e, "hipMemcpy(owned tiles)");
    if ((e = hipMemcpy(pt->d_tile_slot, slot.data(), sizeof(int32_t) * slot.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(tile slots)");
    if (!raster.empty() && (e = hipMemcpy(pt->d_raster_tiles, raster.data(), sizeof(int32_t) * raster.size(), hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e, "hipMemcpy(raster tiles)");
    pt->num_owned = (int)owned.size(); pt->max_owned = max_owned; pt->num_raster_tiles = (int)raster.size();
    // A host that keeps resizing its window (or varies the split) would otherwise grow this cache for as long as the
    // context lives: beyond 64 keys the older half goes - behind a device-wide wait, since queued kernels may still name
    // those tables (no caller keeps a PartTables pointer across API calls).
    if (ctx->part_tables.size() >= 64) {
        (void)hipDeviceSynchronize();
        for (size_t i = 0; i < 32; i++) {
            PartTables* old = ctx->part_tables[i];
            (void)hipFree(old->d_owned_tiles); (void)hipFree(old->d_tile_slot); (void)hipFree(old->d_raster_tiles);
            delete old;
        }
        ctx->part_tables.erase(ctx->part_tables.begin(), ctx->part_tables.begin() + 32);
    }
    ctx->part_tables.push_back(pt);
    *out = pt;
    return VR_OK;
}
