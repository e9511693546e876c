// Device-side texture sampling: what the linear-clamp sampler does for
// terrain_vs.hlsl:32 (SampleLevel) and terrain_ps.hlsl:15,23 (Sample), in fp32
// with explicit operation order and no contraction, plus the SRGBA8/SNORM16/half
// render-target conversions.  There is no texture unit in this path: texels are
// plain byte/dword loads that hit L2 / Infinity Cache (heightmap 5.3 MB, albedo
// 21 MB with mips).
#pragma once
#include "vr_internal.h"
#include <hip/hip_fp16.h>

__device__ __forceinline__ int vr_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct BilinearSetup { int i00, i10, i01, i11; float fx, fy; };

__device__ __forceinline__ BilinearSetup vr_bilinear_setup(int w, int h, float u, float v)
{
    BilinearSetup s;
    float x = __builtin_fmaf(u, (float)w, -0.5f), y = __builtin_fmaf(v, (float)h, -0.5f);     // fused, as the oracle's tex_bilinear
    float xf = floorf(x), yf = floorf(y);
    s.fx = x - xf; s.fy = y - yf;
    xf = vr_min(vr_max(xf, -1.0f), (float)w); yf = vr_min(vr_max(yf, -1.0f), (float)h);
    int x0 = (int)xf, y0 = (int)yf;
    int x1 = vr_clampi(x0 + 1, 0, w - 1), y1 = vr_clampi(y0 + 1, 0, h - 1);
    x0 = vr_clampi(x0, 0, w - 1); y0 = vr_clampi(y0, 0, h - 1);
    const int r0 = __mul24(y0, w), r1 = __mul24(y1, w);
    s.i00 = r0 + x0; s.i10 = r0 + x1; s.i01 = r1 + x0; s.i11 = r1 + x1;
    return s;
}

struct LodSplit { int l0; float f; };
__device__ __forceinline__ LodSplit vr_lod_split(int levels, float lod)
{
    float maxl = (float)(levels - 1);
    if (!(lod > 0.0f)) lod = 0.0f;
    if (lod > maxl) lod = maxl;
    float lf = floorf(lod);
    LodSplit r; r.l0 = (int)lf; r.f = lod - lf;
    return r;
}

// Implicit LOD from screen-space uv differences (isotropic, D3D11 7.18.11) with the
// pinned cubic log2 (max error 1.1e-3 LOD) so every implementation agrees exactly.
__device__ __forceinline__ float vr_lod_from_derivs_f(float dudx, float dvdx, float dudy, float dvdy, float wf, float hf)
{
    float ax = dudx * wf, ay = dvdx * hf, bx = dudy * wf, by = dvdy * hf;
    float r2x = __builtin_fmaf(ax, ax, ay * ay), r2y = __builtin_fmaf(bx, bx, by * by);      // fused sums of squares (oracle: lod_from_derivs)
    float r2 = __builtin_fmaxf(r2x, r2y);
    // 0.5 * log2(r2) from the bits, unconditionally (no compare / select): the caller clamps to [0, last level]
    uint32_t bits = __float_as_uint(r2);
    int e = (int)((bits >> 23) & 255u) - 127;
    float tt = __uint_as_float((bits & 0x7fffffu) | 0x3f800000u) - 1.0f;
    float p = tt * __builtin_fmaf(tt, __builtin_fmaf(tt, 0.1563861f, -0.57725066f), 1.4208646f);
    return 0.5f * ((float)e + p);
}
__device__ __forceinline__ float vr_lod_from_derivs(float dudx, float dvdx, float dudy, float dvdy, int w, int h)
{
    return vr_lod_from_derivs_f(dudx, dvdx, dudy, dvdy, (float)w, (float)h);
}

// linear -> sRGB8: number of thresholds <= x (thr[0] = 0): round-to-nearest OETF.
__device__ __forceinline__ uint32_t vr_srgb_encode(float x, const float* thr)
{
    if (!(x >= 0.0f)) return 0u;
    int lo = 0, hi = 255;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        int mid = (lo + hi + 1) >> 1;
        bool ge = x >= thr[mid];
        lo = ge ? mid : lo; hi = ge ? hi : mid - 1;
    }
    return (uint32_t)lo;
}

// Same result as vr_srgb_encode (the largest k with thr[k] <= x) without the 8-step search: the
// float's exponent and top 7 mantissa bits select a bucket whose smallest value has code tab[b]
// (host-built from the same thresholds).  A bucket spans less than one code (at most 0.875 of one, at x -> 1),
// so it holds at most one threshold - vr_context_create verifies that for every bucket - and ONE comparison
// against the next threshold finishes the job.  tab / thr live in LDS; thr has 257 entries, thr[256] = NaN (never <= x).
constexpr int kThrTabSize = 257;
__device__ __forceinline__ uint32_t vr_srgb_encode_fast(float x, const float* __restrict__ thr, const uint8_t* __restrict__ tab)
{
    int b = (int)(__float_as_uint(x) >> 16) - kEncTabBase;
    b = b < 0 ? 0 : (b > kEncTabSize - 1 ? kEncTabSize - 1 : b);   // below 2^-13 -> code 0, >= 1 -> 255
    const uint32_t g = tab[b];
    const uint32_t code = g + (x >= thr[g + 1u] ? 1u : 0u);
    return x > 0.0f ? code : 0u;                                   // zero (either sign), negatives and NaN
}

__device__ __forceinline__ uint32_t vr_snorm16(float v)
{
    if (!(v == v)) return 0u;
    v = vr_min(vr_max(v, -1.0f), 1.0f);
    float s = v * 32767.0f;
    int i = (int)(s >= 0.0f ? s + 0.5f : s - 0.5f);
    return (uint32_t)i & 0xffffu;
}
__device__ __forceinline__ float vr_snorm16_decode(uint32_t u16)
{
    int s = (int)(int16_t)(uint16_t)u16;
    return vr_max((float)s / 32767.0f, -1.0f);
}
__device__ __forceinline__ float vr_half_to_float(uint32_t h16)
{
    return __half2float(__ushort_as_half((unsigned short)h16));
}
__device__ __forceinline__ uint32_t vr_float_to_half(float f)
{
    return (uint32_t)__half_as_ushort(__float2half_rn(f));
}
