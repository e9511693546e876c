// Device-side shading model of the deferred lighting pass (DeferredLightingPass::Render, Renderer.cpp:417-428: Donut's
// deferred_lighting_cs / ShadeSurface / GGX_AnalyticalLights_times_NdotL, restated - DESIGN.md 2): G-buffer texel -> radiance.
// Shared by the lighting kernels (vr_deferred.hip) and by the tile pass's fused variant (vr_raster.hip: vr_terrain_render_lit),
// which must produce the same bits from the same encoded texel - hence ONE copy of the arithmetic.
#pragma once
#include "vr_internal.h"
#include "vr_tex_dev.h"

#include <math.h>

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

int vr_deferred_make_args(const vr_view* view, int w, int h, const vr_light* lights, int32_t num_lights, const float amb_top[3],
                          const float amb_bottom[3], DeferredArgs* out, bool* extra_lights);      // (vr_deferred.hip)

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

__device__ __forceinline__ Surface decode_surface(const DeferredArgs& a, const float* __restrict__ lut, int px, int py, float depth,
                                                  uint32_t diff, uint32_t spec, uint32_t n01, uint32_t n23, uint32_t e01, uint32_t e23)
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
    {
        // ReconstructWorldPosition: window -> clip -> world
        float cx, cy;
        {
    #pragma clang fp contract(off)
            cx = ((float)px + 0.5f) * a.sx + -1.0f; cy = ((float)py + 0.5f) * a.sy + 1.0f;
        }
        if (a.exact_pos) {
            // Far from the camera clip -> world is ill-conditioned (w = depth * c2w[11] + c2w[15] cancels to a few significant
            // bits: at depth 0.9999 one rounding moves the point by a world unit).  A directional light does not care, a
            // point light's distance and direction do, so with positional lights in the list the sums are evaluated in the
            // checker's order, without contraction (wave-uniform branch; the sun-only pass is unchanged).  One correctly
            // rounded 1/w and three products instead of the checker's three divisions: a last-place difference of the
            // quotients is not amplified, the cancellation is in the sums.
    #pragma clang fp contract(off)
            float e4[4];
    #pragma unroll
            for (int j = 0; j < 4; j++) e4[j] = ((cx * a.c2w[0 * 4 + j] + cy * a.c2w[1 * 4 + j]) + depth * a.c2w[2 * 4 + j]) + a.c2w[3 * 4 + j];
            const float rw4 = vr_rcp_exact(e4[3]);                    // = 1.0f / w for 2^-60 <= |w| <= 2^60 (vr_internal.h)
            s.wp[0] = e4[0] * rw4; s.wp[1] = e4[1] * rw4; s.wp[2] = e4[2] * rw4;
        } else {
            float wp4[4];
    #pragma unroll
            for (int j = 0; j < 4; j++) wp4[j] = cx * a.c2w[0 * 4 + j] + cy * a.c2w[1 * 4 + j] + depth * a.c2w[2 * 4 + j] + a.c2w[3 * 4 + j];
            const float rw = fast_rcp(wp4[3]);
            s.wp[0] = wp4[0] * rw; s.wp[1] = wp4[1] * rw; s.wp[2] = wp4[2] * rw;
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
                                          const LightExtra* extra = nullptr, bool punctual = false)
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
    // punctual (compile-time; the tiled pass's point lights, cosH = 1, sinH = 0): k1 = 1, k2 = 0 and CL = L, except where
    // R.L rounds to 1 or more - there the general form takes R, which then equals L to within rounding
    const float CL[3] = { punctual ? L[0] : L[0] * k1 + s.R[0] * k2, punctual ? L[1] : L[1] * k1 + s.R[1] * k2,
                          punctual ? L[2] : L[2] * k1 + s.R[2] * k2 };
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

