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

struct DeferredArgs {
    float c2w[16];
    float cam[3];
    float sx, sy;          // windowToClipScale = (2/W, -2/H)
    int w, h;
    int num_lights;
    float amb_top[3], amb_bot[3];
    int tiles_x;           // owner tiles per row (packed mode)
    int exact_pos;         // the light list has positional lights: reconstruct the world position in the checker's arithmetic
    DevLight lights[kMaxLights];
};

#define VR_PI 3.14159265358979323846f
#define VR_INV_PI 0.318309886183790671538f

// Per-pixel shading.  Unlike the G-buffer pass (bit-exact integer/byte outputs), this
// kernel's contract is the stated floating-point tolerance (per-channel RMS <= 1e-4 vs
// the fp32 oracle; measured ~1e-8): it uses v_rcp_f32 / v_rsq_f32 (1 ulp), lets the
// compiler contract mul+add into FMA, and folds the three divisions of the GGX term
// (D, G, sphere normalisation) into one reciprocal.  ~5 transcendental-rate
// instructions per pixel and light instead of ~15 IEEE divisions.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// min / max / saturate as single instructions (v_max_f32, v_min_f32, v_med3_f32).  The generic a > b ? a : b forms of
// vr_internal.h keep C's NaN behaviour and cost a compare + select each; nothing on these paths is NaN (the only
// non-finite inputs a G-buffer can carry, emissive halves, are added at the very end).
__device__ __forceinline__ float fmax1(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float fmin1(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fsat1(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float dot3c(float ax, float ay, float az, float bx, float by, float bz)
{
#pragma clang fp contract(fast)
    return ax * bx + ay * by + az * bz;
}

// Everything of a pixel that does not depend on the light.
struct Surface {
    float albedo[3], F0[3], E[3], N[3], wp[3], vi[3], R[3];
    float occlusion, alpha, a2, kk, gv;
};

// known_wp: the world position, when the caller has already reconstructed it (the tiled pass's tile box)
__device__ __forceinline__ Surface decode_surface(const DeferredArgs& a, const float* __restrict__ lut, int px, int py, float depth,
                                                  uint32_t diff, uint32_t spec, uint32_t n01, uint32_t n23, uint32_t e01, uint32_t e23,
                                                  const float* known_wp = nullptr)
{
#pragma clang fp contract(fast)
    Surface s;
#pragma unroll
    for (int c = 0; c < 3; c++) { s.albedo[c] = lut[(diff >> (8 * c)) & 255u]; s.F0[c] = lut[(spec >> (8 * c)) & 255u]; }
    s.occlusion = (float)(spec >> 24) * (1.0f / 255.0f);
    const float sn16 = 1.0f / 32767.0f;
    s.N[0] = fmax1((float)(int16_t)(n01 & 0xffffu) * sn16, -1.0f); s.N[1] = fmax1((float)(int16_t)(n01 >> 16) * sn16, -1.0f);
    s.N[2] = fmax1((float)(int16_t)(n23 & 0xffffu) * sn16, -1.0f);
    const float rough = fmax1((float)(int16_t)(n23 >> 16) * sn16, -1.0f);
    s.E[0] = vr_half_to_float(e01 & 0xffffu); s.E[1] = vr_half_to_float(e01 >> 16); s.E[2] = vr_half_to_float(e23 & 0xffffu);
    if (known_wp) { s.wp[0] = known_wp[0]; s.wp[1] = known_wp[1]; s.wp[2] = known_wp[2]; }
    else {
        // ReconstructWorldPosition: window -> clip -> world
        float cx, cy;
        {
    #pragma clang fp contract(off)
            cx = ((float)px + 0.5f) * a.sx + -1.0f; cy = ((float)py + 0.5f) * a.sy + 1.0f;
        }
        float wp4[4];
    #pragma unroll
        for (int j = 0; j < 4; j++) wp4[j] = cx * a.c2w[0 * 4 + j] + cy * a.c2w[1 * 4 + j] + depth * a.c2w[2 * 4 + j] + a.c2w[3 * 4 + j];
        const float rw = fast_rcp(wp4[3]);
        s.wp[0] = wp4[0] * rw; s.wp[1] = wp4[1] * rw; s.wp[2] = wp4[2] * rw;
        if (a.exact_pos) {
            // Far from the camera clip -> world is ill-conditioned (w = depth * c2w[11] + c2w[15] cancels to a few significant
            // bits: at depth 0.9999 one rounding moves the point by a world unit).  A directional light does not care, a
            // point light's distance and direction do, so with positional lights in the list the position is evaluated in the
            // checker's exact order - no contraction, IEEE divisions (wave-uniform branch; the sun-only pass is unchanged).
    #pragma clang fp contract(off)
            float e4[4];
    #pragma unroll
            for (int j = 0; j < 4; j++) e4[j] = ((cx * a.c2w[0 * 4 + j] + cy * a.c2w[1 * 4 + j]) + depth * a.c2w[2 * 4 + j]) + a.c2w[3 * 4 + j];
            s.wp[0] = e4[0] / e4[3]; s.wp[1] = e4[1] / e4[3]; s.wp[2] = e4[2] / e4[3];
        }
    }
    const float d[3] = { s.wp[0] - a.cam[0], s.wp[1] - a.cam[1], s.wp[2] - a.cam[2] };
    const float dl = fast_rsq(dot3c(d[0], d[1], d[2], d[0], d[1], d[2]));
    s.vi[0] = d[0] * dl; s.vi[1] = d[1] * dl; s.vi[2] = d[2] * dl;      // viewIncident; V = -vi
    const float NdotVi = dot3c(s.vi[0], s.vi[1], s.vi[2], s.N[0], s.N[1], s.N[2]);
    const float two = 2.0f * NdotVi;
#pragma unroll
    for (int c = 0; c < 3; c++) s.R[c] = s.vi[c] - s.N[c] * two;        // reflect(viewIncident, N)
    const float NdotV = fsat1(-NdotVi);
    s.alpha = fmax1(0.01f, rough * rough);
    s.a2 = s.alpha * s.alpha;
    s.kk = ((rough + 1.0f) * (rough + 1.0f)) * 0.125f;
    s.gv = NdotV * (1.0f - s.kk) + s.kk;
    return s;
}

// One light's contribution (ShadeSurface + GGX_AnalyticalLights_times_NdotL).  type/vec/inv_range etc.
// are the DevLight fields; passed separately so that they may come from SGPRs or from LDS.
// Spot cone and spherical-source terms of ShadeSurface (only evaluated for such lights).
struct LightExtra { float axis[3]; float radius, inner_angle, outer_angle; };

__device__ __forceinline__ void add_light(const Surface& s, int type, const float vec[3], float inv_range, const float color[3],
                                          float intensity, float cosH, float sinH, float tanH, float diffuseTerm[3], float specularTerm[3],
                                          const LightExtra* extra = nullptr)
{
#pragma clang fp contract(fast)
    float L[3], irr;                                                  // L = -incidentVector
    if (type == VR_LIGHT_DIRECTIONAL) {
        L[0] = -vec[0]; L[1] = -vec[1]; L[2] = -vec[2];
        irr = intensity;
    } else {
        const float stl[3] = { vec[0] - s.wp[0], vec[1] - s.wp[1], vec[2] - s.wp[2] };
        const float d2 = dot3c(stl[0], stl[1], stl[2], stl[0], stl[1], stl[2]);
        const float rd = fast_rsq(d2);
        L[0] = stl[0] * rd; L[1] = stl[1] * rd; L[2] = stl[2] * rd;
        float att = 1.0f;
        if (inv_range > 0.0f) {
            const float q2 = d2 * (inv_range * inv_range);
            const float sa = fsat1(1.0f - q2 * q2);
            att = sa * sa;
            if (att == 0.0f) return;
        }
        irr = intensity * (rd * rd);
        if (extra != nullptr) {
            if (type == VR_LIGHT_SPOT) {
                const float LdotD = fmin1(fmax1(-dot3c(L[0], L[1], L[2], extra->axis[0], extra->axis[1], extra->axis[2]), -1.0f), 1.0f);
                const float ts = fsat1((acosf(LdotD) - extra->inner_angle) * fast_rcp(extra->outer_angle - extra->inner_angle));
                const float spotlight = 1.0f - ts * ts * (3.0f - 2.0f * ts);
                if (spotlight == 0.0f) return;
                att *= spotlight;
            }
            if (extra->radius > 0.0f) {
                const float x = fmin1(extra->radius * rd, 1.0f);
                const float halfAng = atanf(x);
                irr = (intensity * fast_rcp(extra->radius * extra->radius)) * (halfAng * halfAng);
                tanH = x; cosH = fast_rsq(1.0f + x * x); sinH = x * cosH;
            }
        }
        irr *= att;
    }
    const float NdotLd = fmax1(dot3c(s.N[0], s.N[1], s.N[2], L[0], L[1], L[2]), 0.0f);
    const float kd = (NdotLd * VR_INV_PI) * irr;
    // area-light correction of L towards R (closed form of Donut's slerp)
    const float cosT = fmin1(fmax1(dot3c(s.R[0], s.R[1], s.R[2], L[0], L[1], L[2]), -1.0f), 1.0f);
    float k1 = 0.0f, k2 = 1.0f;                                       // cosT >= cosH: CL = R
    if (cosT < cosH) {
        k2 = sinH * fast_rsq(fmax1(1.0f - cosT * cosT, 1e-12f));
        k1 = cosH - cosT * k2;
    }
    const float CL[3] = { L[0] * k1 + s.R[0] * k2, L[1] * k1 + s.R[1] * k2, L[2] * k1 + s.R[2] * k2 };
    const float Hv[3] = { CL[0] - s.vi[0], CL[1] - s.vi[1], CL[2] - s.vi[2] };
    const float hl2 = dot3c(Hv[0], Hv[1], Hv[2], Hv[0], Hv[1], Hv[2]);
    const float hs = hl2 > 0.0f ? fast_rsq(hl2) : 0.0f;
    const float NdotH = fsat1(dot3c(s.N[0], s.N[1], s.N[2], Hv[0], Hv[1], Hv[2]) * hs);
    const float NdotL = fsat1(dot3c(s.N[0], s.N[1], s.N[2], CL[0], CL[1], CL[2]));
    const float VdotH = fsat1(-dot3c(s.vi[0], s.vi[1], s.vi[2], Hv[0], Hv[1], Hv[2]) * hs);
    const float corrAlpha = fsat1(s.alpha + 0.5f * tanH);
    const float dd = (NdotH * NdotH) * (s.a2 - 1.0f) + 1.0f;
    const float gl = NdotL * (1.0f - s.kk) + s.kk;
    // D * G * NdotL / 4 * irradiance with D = a2/(pi dd^2) (alpha/corrAlpha)^2, G = 1/(gl gv)
    const float num = (s.a2 * s.a2) * (NdotL * irr) * (0.25f * VR_INV_PI);
    const float den = ((corrAlpha * dd) * (corrAlpha * dd)) * (gl * s.gv);
    const float ks = num * fast_rcp(den);
    const float om = 1.0f - VdotH;
    const float om2 = om * om;
    const float fw = (om2 * om2) * om;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float F = s.F0[c] + (1.0f - s.F0[c]) * fw;
        diffuseTerm[c] += (s.albedo[c] * kd) * color[c];
        specularTerm[c] += (F * ks) * color[c];
    }
}

__device__ __forceinline__ void finish_pixel(const DeferredArgs& a, const Surface& s, const float diffuseTerm[3], const float specularTerm[3],
                                             float out[3])
{
#pragma clang fp contract(fast)
    const float tt = s.N[1] * 0.5f + 0.5f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float amb = (a.amb_bot[c] + (a.amb_top[c] - a.amb_bot[c]) * tt) * s.occlusion;
        out[c] = (diffuseTerm[c] + amb * s.albedo[c]) + (specularTerm[c] + amb * s.F0[c]) + s.E[c];
    }
}

// ---- shadow term (row f1) -----------------------------------------------------------------------
// [DONUT-RECOLLECTION of EvaluateShadowGather16] world -> light clip -> uv; outside the map:
// outOfBoundsShadow; else the 4x4 texel footprint, each texel compared LessEqual (receiver depth - bias
// <= stored depth), weights [1-fx, 1, 1, fx] x [1-fy, 1, 1, fy] / 9.  A comparison is a step function,
// so unlike the BRDF this part is evaluated exactly as the checker does: IEEE divisions, no contraction,
// its own world-position reconstruction (the shaded one uses v_rcp_f32) - otherwise pixels whose receiver
// depth sits on a stored depth would flip and the RMS contract could not hold.
struct ShadowArgs {
    float w2c[16];                 // light's world -> clip
    const float* depth;            // res x res shadow map
    int res, light_index;
    float bias, out_of_bounds;
};

__device__ __forceinline__ float shadow_factor(const DeferredArgs& a, const ShadowArgs& s, int px, int py, float depth)
{
    const float cx = ((float)px + 0.5f) * a.sx + -1.0f, cy = ((float)py + 0.5f) * a.sy + 1.0f;
    float w4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w4[j] = ((cx * a.c2w[0 * 4 + j] + cy * a.c2w[1 * 4 + j]) + depth * a.c2w[2 * 4 + j]) + a.c2w[3 * 4 + j];
    const float wx_ = w4[0] / w4[3], wy_ = w4[1] / w4[3], wz_ = w4[2] / w4[3];
    float c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = ((wx_ * s.w2c[0 * 4 + j] + wy_ * s.w2c[1 * 4 + j]) + wz_ * s.w2c[2 * 4 + j]) + s.w2c[3 * 4 + j];
    const bool w_one = c[3] == 1.0f;                                 // orthographic light: x / 1 == x, skip the divisions
    const float xc = w_one ? c[0] : c[0] / c[3], yc = w_one ? c[1] : c[1] / c[3], zc = w_one ? c[2] : c[2] / c[3];
    const float u = xc * 0.5f + 0.5f, v = 0.5f - yc * 0.5f;
    if (!(u >= 0.0f && u <= 1.0f && v >= 0.0f && v <= 1.0f && zc >= 0.0f && zc <= 1.0f)) return s.out_of_bounds;
    const float z = zc - s.bias;
    const float tx = u * (float)s.res - 0.5f, ty = v * (float)s.res - 0.5f;
    const float fxl = floorf(tx), fyl = floorf(ty);
    const float fx = tx - fxl, fy = ty - fyl;
    const int ix = (int)fxl - 1, iy = (int)fyl - 1;
    const float wgx[4] = { 1.0f - fx, 1.0f, 1.0f, fx }, wgy[4] = { 1.0f - fy, 1.0f, 1.0f, fy };
    int xs[4];
#pragma unroll
    for (int i = 0; i < 4; i++) xs[i] = min(max(ix + i, 0), s.res - 1);
    // the four texels of a footprint row are adjacent unless the footprint hangs over the map's edge: one 16-byte
    // load per row (4-byte aligned is enough for global loads) instead of four - wave-uniform choice
    const bool inner = __all(ix >= 0 && ix + 3 <= s.res - 1);
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float* row_p = s.depth + (size_t)min(max(iy + j, 0), s.res - 1) * s.res;
        float d0, d1, d2, d3;
        if (inner) {
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const f4u d = *reinterpret_cast<const f4u*>(row_p + ix);
            d0 = d.x; d1 = d.y; d2 = d.z; d3 = d.w;
        } else { d0 = row_p[xs[0]]; d1 = row_p[xs[1]]; d2 = row_p[xs[2]]; d3 = row_p[xs[3]]; }
        float row = 0.0f;
        row = row + (z <= d0 ? 1.0f : 0.0f) * wgx[0];
        row = row + (z <= d1 ? 1.0f : 0.0f) * wgx[1];
        row = row + (z <= d2 ? 1.0f : 0.0f) * wgx[2];
        row = row + (z <= d3 ? 1.0f : 0.0f) * wgx[3];
        sum = sum + row * wgy[j];
    }
    return sum / 9.0f;
}

// EXTRA: the light list contains spot or spherical lights (compiled out of the common variant, which
// keeps the streaming kernel at its leanest for directional / punctual lights).
template <bool EXTRA, bool SHADOW = false>
__device__ __forceinline__ void shade_pixel(const DeferredArgs& a, const float* __restrict__ lut, int px, int py, float depth,
                                            uint32_t diff, uint32_t spec, uint32_t n01, uint32_t n23, uint32_t e01, uint32_t e23,
                                            float out[3], const ShadowArgs* sh = nullptr)
{
    const Surface s = decode_surface(a, lut, px, py, depth, diff, spec, n01, n23, e01, e23);
    float diffuseTerm[3] = { 0.0f, 0.0f, 0.0f }, specularTerm[3] = { 0.0f, 0.0f, 0.0f };
    float sf = 1.0f;
    if (SHADOW) sf = shadow_factor(a, *sh, px, py, depth);
    for (int i = 0; i < a.num_lights; i++) {
        const DevLight& Lc = a.lights[i];
        const float* vec = Lc.type == VR_LIGHT_DIRECTIONAL ? Lc.dir : Lc.pos;
        if (SHADOW && i == sh->light_index) {          // a directional light (checked on the host): irradiance = intensity * shadow
            if (sf == 0.0f) continue;
            add_light(s, Lc.type, vec, Lc.inv_range, Lc.color, Lc.intensity * sf, Lc.cosH, Lc.sinH, Lc.tanH, diffuseTerm, specularTerm);
            continue;
        }
        if (EXTRA && (Lc.type == VR_LIGHT_SPOT || Lc.radius > 0.0f)) {
            LightExtra ex; ex.axis[0] = Lc.dir[0]; ex.axis[1] = Lc.dir[1]; ex.axis[2] = Lc.dir[2];
            ex.radius = Lc.radius; ex.inner_angle = Lc.inner_angle; ex.outer_angle = Lc.outer_angle;
            add_light(s, Lc.type, vec, Lc.inv_range, Lc.color, Lc.intensity, Lc.cosH, Lc.sinH, Lc.tanH, diffuseTerm, specularTerm, &ex);
        } else {
            add_light(s, Lc.type, vec, Lc.inv_range, Lc.color, Lc.intensity, Lc.cosH, Lc.sinH, Lc.tanH, diffuseTerm, specularTerm);
        }
    }
    finish_pixel(a, s, diffuseTerm, specularTerm, out);
}

// Output of 4 pixels (o[2k] = r|g<<16, o[2k+1] = b, alpha 0).  Row-major frames are RGBA16F (8 B/px);
// the packed tile buffer that goes through the all-gather drops the always-zero alpha: RGB16F, 6 B/px,
// 25 % less xGMI traffic, restored by k_detile.
template <bool PACKED>
__device__ __forceinline__ void store_quad(uint2* __restrict__ out, size_t out_index, const uint32_t o[8])
{
    if (PACKED) {
        uint2* dst = reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + out_index * 3);   // 24 B per quad, 8-B aligned
        dst[0] = make_uint2(o[0], (o[1] & 0xffffu) | (o[2] << 16));
        dst[1] = make_uint2((o[2] >> 16) | (o[3] << 16), o[4]);
        dst[2] = make_uint2((o[5] & 0xffffu) | (o[6] << 16), (o[6] >> 16) | (o[7] << 16));
    } else {
        uint4* dst = reinterpret_cast<uint4*>(out + out_index);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

// PACKED = false: whole frame, row-major output; one lane = 4 consecutive pixels.
// PACKED = true : only owner tiles of this rank, output packed tile-major
//                 [local tile][128 rows][128 px]; block = 8 rows x 128 px of a tile.
#ifndef VR_DEFERRED_WAVES
#define VR_DEFERRED_WAVES 4
#endif
template <bool PACKED, bool EXTRA, bool SHADOW = false>
__global__ __launch_bounds__(256, VR_DEFERRED_WAVES) void k_deferred(DeferredArgs a, const float* __restrict__ g_depth,
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
    const float4 dz = *reinterpret_cast<const float4*>(g_depth + p);
    const uint4 df = *reinterpret_cast<const uint4*>(g_diff + p);
    const uint4 sp = *reinterpret_cast<const uint4*>(g_spec + p);
    const uint4 n0 = *reinterpret_cast<const uint4*>(g_nrm + p);
    const uint4 n1 = *reinterpret_cast<const uint4*>(g_nrm + p + 2);
    const uint4 e0 = *reinterpret_cast<const uint4*>(g_emi + p);
    const uint4 e1 = *reinterpret_cast<const uint4*>(g_emi + p + 2);

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
    store_quad<PACKED>(out, out_index, o);
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
        sh.depth = shadow->shadow_map->depth; sh.res = shadow->shadow_map->w; sh.light_index = shadow->light_index;
        sh.bias = shadow->depth_bias; sh.out_of_bounds = lights[shadow->light_index].out_of_bounds_shadow;
    }
    VR_REQUIRE(num_lights >= 0 && num_lights <= kMaxLights, "at most 16 lights (terrain_cb.h:15)");
    VR_REQUIRE(num_lights == 0 || lights, "lights is NULL");
    VR_REQUIRE(view->viewport_w == gb->w && view->viewport_h == gb->h && view->viewport_x == 0 && view->viewport_y == 0,
               "view viewport must cover the G-buffer");
    VR_HIP(hipSetDevice(ctx->device));
    DeferredArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 16; i++) a.c2w[i] = view->clip_to_world[i];
    for (int i = 0; i < 3; i++) { a.cam[i] = view->camera_pos[i]; a.amb_top[i] = amb_top[i]; a.amb_bot[i] = amb_bottom[i]; }
    a.w = gb->w; a.h = gb->h; a.sx = 2.0f / (float)gb->w; a.sy = -2.0f / (float)gb->h;
    a.num_lights = num_lights;
    bool extra = false;
    for (int i = 0; i < num_lights; i++) {
        int rc = fill_light(lights[i], a.lights[i], true); if (rc) return rc;
        extra = extra || a.lights[i].type == VR_LIGHT_SPOT || a.lights[i].radius > 0.0f;
        if (a.lights[i].type != VR_LIGHT_DIRECTIONAL) a.exact_pos = 1;
    }
    const size_t npx = (size_t)gb->w * gb->h;
    const bool packed = part != nullptr;     // a partition (even of one rank) selects the packed tile-major output
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
            auto kern = shadow ? k_deferred<true, true, true> : (extra ? k_deferred<true, true> : k_deferred<true, false>);
            VR_LAUNCH_TIMED(ks, kern, dim3((unsigned)pt->num_owned * 16), dim3(256), ctx->stream, a, gb->depth, gb->diffuse,
                               gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data, ctx->d_srgb_lut, pt->d_owned_tiles, sh);
        }
    } else {
        VR_REQUIRE(npx * 8 <= hdr->capacity_bytes, "hdr_out is smaller than the frame");
        if (gb->w % 4 == 0) {
            const size_t quads = npx / 4;
            auto kern = shadow ? k_deferred<false, true, true> : (extra ? k_deferred<false, true> : k_deferred<false, false>);
            VR_LAUNCH_TIMED(ks, kern, dim3((unsigned)((quads + 255) / 256)), dim3(256), ctx->stream, a, gb->depth, gb->diffuse,
                               gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data, ctx->d_srgb_lut, (const int32_t*)nullptr, sh);
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
// One workgroup = one 32x32 pixel tile, one lane = 4 horizontally adjacent pixels (16-byte loads and
// stores, as in k_deferred).  Phases:
//   1. load the lane's 4 pixels, reconstruct their world positions and reduce the world-space
//      bounding box of the tile's covered pixels (wave shuffles, then 4 partial boxes through LDS);
//   2. cull: lane t tests light c*256+t (sphere = position/range vs the tile box; directional lights
//      always pass); survivors are appended to an LDS list in light order with a wave ballot + popcount
//      prefix, and their constants are staged in LDS (48 B each);
//   3. shade: the lane walks the tile's list once per pixel (LDS broadcast reads) with the same BRDF
//      as k_deferred.
// A light culled here has zero attenuation for every pixel of the tile, so the sum equals the
// all-lights loop of the oracle.
constexpr int kLightTile = 32;
constexpr int kTileLightCap = VR_TILE_LIGHT_CAP;
// 32 B per staged light (4 workgroups of 1024 lights fit a CU's LDS): colour premultiplied by the intensity;
// w = 0 for a point light, 1 + half angular size for a directional one (its cos/sin/tan are then taken per pixel).
struct TiledLight { float vec[3]; float inv_range; float color[3]; float w; };

// Coarse culling, one workgroup per 128x128 macro tile (= owner tile): depth range of the tile's covered pixels -> the
// world-space box of that frustum cell (its 8 corners) -> the lights whose sphere touches the box, in light order, as a
// global list (count, then indices).  The 32x32 tiles of k_deferred_tiled then test a few dozen lights each instead of
// all of them (1024 lights x 32,400 tiles x 64 B per light was most of that kernel's time at 8K).
constexpr int kMacroTile = VR_OWNER_TILE;
__global__ __launch_bounds__(256) void k_light_macro_cull(DeferredArgs a, const DevLight* __restrict__ lights, int num_lights,
                                                           const float* __restrict__ g_depth, int macro_x,
                                                           const int32_t* __restrict__ owned_tiles, uint32_t* __restrict__ lists, int stride)
{
    __shared__ float s_min[4], s_max[4];
    __shared__ float s_box[6];
    __shared__ uint32_t s_wave_count[4];
    __shared__ uint32_t s_count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = owned_tiles ? owned_tiles[blockIdx.x] : (int)blockIdx.x;
    const int ty = tile / macro_x, tx = tile - ty * macro_x;
    const int x0 = tx * kMacroTile, y0 = ty * kMacroTile;
    uint32_t* __restrict__ list = lists + (size_t)tile * stride;
    float dmin = 2.0f, dmax = -1.0f;
    for (int r = 0; r < kMacroTile / 8; r++) {
        const int py = y0 + r * 8 + (tid >> 5), px = x0 + (tid & 31) * 4;
        if (py < a.h && px < a.w) {                                 // the frame width is a multiple of 4
            const float4 d = *reinterpret_cast<const float4*>(g_depth + (size_t)py * a.w + px);
            const float v[4] = { d.x, d.y, d.z, d.w };
#pragma unroll
            for (int k = 0; k < 4; k++) if (v[k] < 1.0f) { dmin = fmin1(dmin, v[k]); dmax = fmax1(dmax, v[k]); }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { dmin = fmin1(dmin, __shfl_xor(dmin, off)); dmax = fmax1(dmax, __shfl_xor(dmax, off)); }
    if (lane == 0) { s_min[wave] = dmin; s_max[wave] = dmax; }
    if (tid == 0) s_count = 0u;
    __syncthreads();
    dmin = fmin1(fmin1(s_min[0], s_min[1]), fmin1(s_min[2], s_min[3]));
    dmax = fmax1(fmax1(s_max[0], s_max[1]), fmax1(s_max[2], s_max[3]));
    if (!(dmin <= dmax)) { if (tid == 0) list[0] = 0u; return; }    // nothing covered: every tile inside skips its lights
    if (wave == 0) {
        // the cell's 8 corners (window edges of the tile x {dmin, dmax}), reconstructed in the shading pass's exact order
        float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f }, far2 = 0.0f;
        if (lane < 8) {
#pragma clang fp contract(off)
            const float wx = (float)(lane & 1 ? min(x0 + kMacroTile, a.w) : x0), wy = (float)(lane & 2 ? min(y0 + kMacroTile, a.h) : y0);
            const float depth = lane & 4 ? dmax : dmin;
            const float cx = wx * a.sx + -1.0f, cy = wy * a.sy + 1.0f;
            float e4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) e4[j] = ((cx * a.c2w[0 * 4 + j] + cy * a.c2w[1 * 4 + j]) + depth * a.c2w[2 * 4 + j]) + a.c2w[3 * 4 + j];
#pragma unroll
            for (int c = 0; c < 3; c++) { lo[c] = hi[c] = e4[c] / e4[3]; const float dc = lo[c] - a.cam[c]; far2 += dc * dc; }
        }
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) {
#pragma unroll
            for (int c = 0; c < 3; c++) { lo[c] = fmin1(lo[c], __shfl_xor(lo[c], off)); hi[c] = fmax1(hi[c], __shfl_xor(hi[c], off)); }
            far2 = fmax1(far2, __shfl_xor(far2, off));
        }
        // Far from the camera clip -> world loses bits (w cancels): a pixel's reconstructed position and these corners may
        // each be off by ~1.5e-3 of their distance (2.4 units at 1600).  The pad covers both.
        const float pad = 4.0e-3f * sqrtf(far2) + 1.0e-2f;
        if (lane == 0) { for (int c = 0; c < 3; c++) { s_box[c] = lo[c] - pad; s_box[3 + c] = hi[c] + pad; } }
    }
    __syncthreads();
    const float lo[3] = { s_box[0], s_box[1], s_box[2] }, hi[3] = { s_box[3], s_box[4], s_box[5] };
    for (int base = 0; base < num_lights; base += 256) {
        const int li = base + tid;
        bool keep = false;
        if (li < num_lights) {
            const DevLight L = lights[li];
            if (L.type == VR_LIGHT_DIRECTIONAL || !(L.inv_range > 0.0f)) keep = true;
            else {
                float d2 = 0.0f;
#pragma unroll
                for (int c = 0; c < 3; c++) { const float d = fmax1(fmax1(lo[c] - L.pos[c], L.pos[c] - hi[c]), 0.0f); d2 += d * d; }
                const float r = 1.0f / L.inv_range;
                keep = d2 <= (r * r) * 1.0001f;
            }
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave_count[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = s_count;
        for (int w = 0; w < wave; w++) before += s_wave_count[w];
        const uint32_t total = s_wave_count[0] + s_wave_count[1] + s_wave_count[2] + s_wave_count[3];
        if (keep) list[1u + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)li;
        __syncthreads();
        if (tid == 0) s_count += total;
        __syncthreads();
    }
    if (tid == 0) list[0] = s_count;
}

template <bool PACKED>
__global__ __launch_bounds__(256) void k_deferred_tiled(DeferredArgs a, const DevLight* __restrict__ lights, int num_lights,
                                                         const float* __restrict__ g_depth, const uint32_t* __restrict__ g_diff,
                                                         const uint32_t* __restrict__ g_spec, const uint2* __restrict__ g_nrm,
                                                         const uint2* __restrict__ g_emi, uint2* __restrict__ out,
                                                         const float* __restrict__ lut_g, const int32_t* __restrict__ owned_tiles,
                                                         uint32_t* __restrict__ overflow_flag, const uint32_t* __restrict__ macro_lists,
                                                         int macro_x, int macro_stride)
{
#pragma clang fp contract(fast)
    __shared__ float lut[256];
    __shared__ TiledLight s_light[kTileLightCap];
    __shared__ float s_box[4][6];
    __shared__ uint32_t s_wave_count[4];
    __shared__ uint32_t s_count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    lut[tid] = lut_g[tid];
    if (tid == 0) s_count = 0u;

    int px0, py; size_t out_index;
    const int lx = (tid & 7) * 4, ly = tid >> 3;                     // 8 lanes x 4 px per row, 32 rows
    if (PACKED) {
        const int sub = VR_OWNER_TILE / kLightTile;                  // 4 light tiles per owner-tile side
        const int lt = blockIdx.x / (sub * sub), st = blockIdx.x - lt * (sub * sub);
        const int tile = owned_tiles[lt];
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int ox = (st % sub) * kLightTile + lx, oy = (st / sub) * kLightTile + ly;
        px0 = tx * VR_OWNER_TILE + ox; py = ty * VR_OWNER_TILE + oy;
        out_index = ((size_t)lt * VR_OWNER_TILE + oy) * VR_OWNER_TILE + ox;
    } else {
        const int tiles_x = (a.w + kLightTile - 1) / kLightTile;
        const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
        px0 = tx * kLightTile + lx; py = ty * kLightTile + ly;
        out_index = (size_t)py * a.w + px0;
    }
    const bool inside = px0 < a.w && py < a.h;                       // the frame width is a multiple of 4
    __syncthreads();

    float depth[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
    uint32_t dfa[4] = { 0, 0, 0, 0 }, spa[4] = { 0, 0, 0, 0 }, na[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, ea[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (inside) {
        const size_t p = (size_t)py * a.w + px0;
        const float4 dz = *reinterpret_cast<const float4*>(g_depth + p);
        const uint4 df = *reinterpret_cast<const uint4*>(g_diff + p);
        const uint4 sp = *reinterpret_cast<const uint4*>(g_spec + p);
        const uint4 n0 = *reinterpret_cast<const uint4*>(g_nrm + p);
        const uint4 n1 = *reinterpret_cast<const uint4*>(g_nrm + p + 2);
        const uint4 e0 = *reinterpret_cast<const uint4*>(g_emi + p);
        const uint4 e1 = *reinterpret_cast<const uint4*>(g_emi + p + 2);
        depth[0] = dz.x; depth[1] = dz.y; depth[2] = dz.z; depth[3] = dz.w;
        dfa[0] = df.x; dfa[1] = df.y; dfa[2] = df.z; dfa[3] = df.w; spa[0] = sp.x; spa[1] = sp.y; spa[2] = sp.z; spa[3] = sp.w;
        na[0] = n0.x; na[1] = n0.y; na[2] = n0.z; na[3] = n0.w; na[4] = n1.x; na[5] = n1.y; na[6] = n1.z; na[7] = n1.w;
        ea[0] = e0.x; ea[1] = e0.y; ea[2] = e0.z; ea[3] = e0.w; ea[4] = e1.x; ea[5] = e1.y; ea[6] = e1.z; ea[7] = e1.w;
    }
    // ---- 1. tile bounding box in world space (background pixels receive no light: albedo = F0 = N = 0)
    const float big = 3.0e38f;
    float lo[3] = { big, big, big }, hi[3] = { -big, -big, -big };
    float wpos[4][3];                                                // the pixels' world positions, reused by the shading below
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float* pos = wpos[k];
        pos[0] = pos[1] = pos[2] = 0.0f;
        if (!(inside && depth[k] < 1.0f)) continue;
        if (a.exact_pos) {       // the positions the shading below uses (decode_surface): far pixels are ill-conditioned
#pragma clang fp contract(off)
            const float cx = ((float)(px0 + k) + 0.5f) * a.sx + -1.0f, cy = ((float)py + 0.5f) * a.sy + 1.0f;
            float e4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) e4[j] = ((cx * a.c2w[0 * 4 + j] + cy * a.c2w[1 * 4 + j]) + depth[k] * a.c2w[2 * 4 + j]) + a.c2w[3 * 4 + j];
            pos[0] = e4[0] / e4[3]; pos[1] = e4[1] / e4[3]; pos[2] = e4[2] / e4[3];
        } else {
            const float cx = ((float)(px0 + k) + 0.5f) * a.sx - 1.0f, cy = ((float)py + 0.5f) * a.sy + 1.0f;
            float w4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) w4[j] = cx * a.c2w[0 * 4 + j] + cy * a.c2w[1 * 4 + j] + depth[k] * a.c2w[2 * 4 + j] + a.c2w[3 * 4 + j];
            const float rw = fast_rcp(w4[3]);
            pos[0] = w4[0] * rw; pos[1] = w4[1] * rw; pos[2] = w4[2] * rw;
        }
#pragma unroll
        for (int c = 0; c < 3; c++) { lo[c] = fmin1(lo[c], pos[c]); hi[c] = fmax1(hi[c], pos[c]); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) { lo[c] = fmin1(lo[c], __shfl_xor(lo[c], off)); hi[c] = fmax1(hi[c], __shfl_xor(hi[c], off)); }
    }
    if (lane == 0) { for (int c = 0; c < 3; c++) { s_box[wave][c] = lo[c]; s_box[wave][3 + c] = hi[c]; } }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; c++) {
        lo[c] = fmin1(fmin1(s_box[0][c], s_box[1][c]), fmin1(s_box[2][c], s_box[3][c]));
        hi[c] = fmax1(fmax1(s_box[0][3 + c], s_box[1][3 + c]), fmax1(s_box[2][3 + c], s_box[3][3 + c]));
    }
    const bool any_covered = lo[0] <= hi[0];
    // reconstruction here and in decode_surface may differ in the last bit: pad the box
#pragma unroll
    for (int c = 0; c < 3; c++) { const float pad = 1e-4f * fmax1(fabsf(lo[c]), fabsf(hi[c])) + 1e-6f; lo[c] -= pad; hi[c] += pad; }

    // ---- 2. cull the macro tile's list (k_light_macro_cull), 256 lights per round, list kept in light order
    const uint32_t* __restrict__ mlist = macro_lists + (size_t)((py / kMacroTile) * macro_x + (px0 / kMacroTile)) * macro_stride;
    const int n_macro = any_covered ? (int)mlist[0] : 0;       // (py, px0: this lane's pixel; the whole 32x32 tile lies in one macro tile)
    for (int base = 0; base < n_macro; base += 256) {
        const int li = base + tid;
        bool keep = false;
        DevLight L;
        if (li < n_macro) {
            L = lights[mlist[1 + li]];
            if (L.type == VR_LIGHT_DIRECTIONAL || !(L.inv_range > 0.0f)) keep = true;
            else {
                float d2 = 0.0f;
#pragma unroll
                for (int c = 0; c < 3; c++) { const float d = fmax1(fmax1(lo[c] - L.pos[c], L.pos[c] - hi[c]), 0.0f); d2 += d * d; }
                const float r = 1.0f / L.inv_range;
                keep = d2 <= (r * r) * 1.0001f;                      // attenuation is exactly 0 from the range outwards
            }
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave_count[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = s_count;
        for (int w = 0; w < wave; w++) before += s_wave_count[w];
        const uint32_t total = s_wave_count[0] + s_wave_count[1] + s_wave_count[2] + s_wave_count[3];
        if (keep) {
            const uint32_t slot = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (slot < (uint32_t)kTileLightCap) {
                TiledLight t;
                const float* v = L.type == VR_LIGHT_DIRECTIONAL ? L.dir : L.pos;
                t.vec[0] = v[0]; t.vec[1] = v[1]; t.vec[2] = v[2]; t.inv_range = L.inv_range;
                t.color[0] = L.color[0] * L.intensity; t.color[1] = L.color[1] * L.intensity; t.color[2] = L.color[2] * L.intensity;
                t.w = L.type == VR_LIGHT_DIRECTIONAL ? 1.0f + atan2f(L.sinH, L.cosH) : 0.0f;
                s_light[slot] = t;
            } else atomicOr(overflow_flag, 1u);
        }
        __syncthreads();
        if (tid == 0) s_count = min(s_count + total, (uint32_t)kTileLightCap);
        __syncthreads();
    }

    // ---- 3. shade
    if (!inside) return;
    const uint32_t n = s_count;
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // (a background pixel's position is never used: albedo = F0 = N = 0 and it receives no light)
        const Surface s = decode_surface(a, lut, px0 + k, py, depth[k], dfa[k], spa[k], na[2 * k], na[2 * k + 1], ea[2 * k], ea[2 * k + 1], wpos[k]);
        float diffuseTerm[3] = { 0.0f, 0.0f, 0.0f }, specularTerm[3] = { 0.0f, 0.0f, 0.0f };
        if (depth[k] < 1.0f) {
            for (uint32_t i = 0; i < n; i++) {
                const TiledLight& t = s_light[i];
                if (t.w > 0.0f) {                                  // directional (block-uniform branch)
                    const float half = t.w - 1.0f, ch = __cosf(half), sh = __sinf(half);
                    add_light(s, VR_LIGHT_DIRECTIONAL, t.vec, 0.0f, t.color, 1.0f, ch, sh, sh * fast_rcp(ch), diffuseTerm, specularTerm);
                } else {
                    add_light(s, VR_LIGHT_POINT, t.vec, t.inv_range, t.color, 1.0f, 1.0f, 0.0f, 0.0f, diffuseTerm, specularTerm);
                }
            }
        }
        float rgb[3];
        finish_pixel(a, s, diffuseTerm, specularTerm, rgb);
        o[2 * k] = vr_float_to_half(rgb[0]) | (vr_float_to_half(rgb[1]) << 16);
        o[2 * k + 1] = vr_float_to_half(rgb[2]);
    }
    store_quad<PACKED>(out, out_index, o);
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
        (void)hipFree(ctx->d_lights); ctx->d_lights = nullptr; ctx->light_capacity = 0;
        const size_t cap = (size_t)num_lights < 1024 ? 1024 : (size_t)num_lights;
        VR_HIP(hipMalloc(&ctx->d_lights, cap * sizeof(DevLight)));
        ctx->light_capacity = cap;
    }
    if (!ctx->d_flags) { VR_HIP(hipMalloc(&ctx->d_flags, 64)); VR_HIP(hipMemsetAsync(ctx->d_flags, 0, 64, ctx->stream)); }
    ctx->h_lights.resize((size_t)num_lights);
    for (int i = 0; i < num_lights; i++) { int rc = fill_light(lights[i], ctx->h_lights[i], false); if (rc) return rc; }
    if (num_lights) VR_HIP(hipMemcpyAsync(ctx->d_lights, ctx->h_lights.data(), (size_t)num_lights * sizeof(DevLight), hipMemcpyHostToDevice, ctx->stream));
    DeferredArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 16; i++) a.c2w[i] = view->clip_to_world[i];
    for (int i = 0; i < 3; i++) { a.cam[i] = view->camera_pos[i]; a.amb_top[i] = amb_top[i]; a.amb_bot[i] = amb_bottom[i]; }
    a.w = gb->w; a.h = gb->h; a.sx = 2.0f / (float)gb->w; a.sy = -2.0f / (float)gb->h; a.num_lights = 0;
    for (int i = 0; i < num_lights; i++) if (ctx->h_lights[i].type != VR_LIGHT_DIRECTIONAL) a.exact_pos = 1;
    const bool packed = part != nullptr;
    VR_REQUIRE(gb->w % 4 == 0, "the tiled pass needs a frame width that is a multiple of 4");
    // coarse lists: one per 128x128 macro tile, count + up to num_lights indices
    const int macro_x = (gb->w + kMacroTile - 1) / kMacroTile, macro_y = (gb->h + kMacroTile - 1) / kMacroTile;
    const int stride = num_lights + 1;
    const size_t words = (size_t)macro_x * macro_y * stride;
    if (words > ctx->macro_list_words) {
        VR_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_macro_lists); ctx->d_macro_lists = nullptr; ctx->macro_list_words = 0;
        VR_HIP(hipMalloc(&ctx->d_macro_lists, words * sizeof(uint32_t)));
        ctx->macro_list_words = words;
    }
    VrKernelScope ks(ctx, VR_K_DEFERRED_TILED);
    if (packed) {
        const PartTables* pt = nullptr;
        int rc = vr_partition_tables(ctx, gb->w, gb->h, part, &pt);
        if (rc) return rc;
        VR_REQUIRE((size_t)pt->max_owned * VR_OWNER_TILE * VR_OWNER_TILE * 6 <= hdr->capacity_bytes, "hdr_out is smaller than vr_partition_packed_bytes()");
        a.tiles_x = (gb->w + VR_OWNER_TILE - 1) / VR_OWNER_TILE;
        const int sub = VR_OWNER_TILE / kLightTile;
        if (pt->num_owned > 0) {
            hipLaunchKernelGGL(k_light_macro_cull, dim3((unsigned)pt->num_owned), dim3(256), 0, ctx->stream, a, ctx->d_lights, num_lights,
                               gb->depth, macro_x, pt->d_owned_tiles, ctx->d_macro_lists, stride);
            hipLaunchKernelGGL(k_deferred_tiled<true>, dim3((unsigned)pt->num_owned * sub * sub), dim3(256), 0, ctx->stream, a, ctx->d_lights,
                               num_lights, gb->depth, gb->diffuse, gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data,
                               ctx->d_srgb_lut, pt->d_owned_tiles, ctx->d_flags, ctx->d_macro_lists, macro_x, stride);
        }
    } else {
        VR_REQUIRE((size_t)gb->w * gb->h * 8 <= hdr->capacity_bytes, "hdr_out is smaller than the frame");
        const int tx = (gb->w + kLightTile - 1) / kLightTile, ty = (gb->h + kLightTile - 1) / kLightTile;
        hipLaunchKernelGGL(k_light_macro_cull, dim3((unsigned)(macro_x * macro_y)), dim3(256), 0, ctx->stream, a, ctx->d_lights, num_lights,
                           gb->depth, macro_x, (const int32_t*)nullptr, ctx->d_macro_lists, stride);
        hipLaunchKernelGGL(k_deferred_tiled<false>, dim3((unsigned)(tx * ty)), dim3(256), 0, ctx->stream, a, ctx->d_lights, num_lights,
                           gb->depth, gb->diffuse, gb->specular, gb->normals, gb->emissive, (uint2*)hdr->data, ctx->d_srgb_lut,
                           (const int32_t*)nullptr, ctx->d_flags, ctx->d_macro_lists, macro_x, stride);
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
