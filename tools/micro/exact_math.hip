// Scratch: exhaustive search for SHORT instruction sequences that are bit-identical to the IEEE results the oracle computes
// (1.0f / x, sqrtf(x), 1.0f / sqrtf(x)) over the ranges the tile pass's pixel shader needs.  Every float of the range is
// tried; a candidate is usable when its mismatch count is 0.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/exact_math.hip -o tools/micro/exact_math
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ float rcp_a(float d)      // the product's sequence (round 2): rcp + 6 fma
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;
    float rem = __builtin_fmaf(-d, q, 1.0f);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(rem, r, q);
}
__device__ __forceinline__ float rcp_b(float d)      // rcp + 4 fma: two Newton steps with exact residuals
{
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_c(float d)      // rcp + 2 fma
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float sqrt_a(float x)     // the product's sequence (round 2): sqrt + neighbour selection
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = __builtin_fmaf(-dn, s, x), vs = __builtin_fmaf(-up, s, x);
    float r = vp <= 0.0f ? dn : s;
    r = vs > 0.0f ? up : r;
    return r;
}
__device__ __forceinline__ float sqrt_b(float x, float* half_rsq)     // rsq + 2 mul + 5 fma, no select (the compiler's own refinement, unscaled)
{
    const float r = __builtin_amdgcn_rsqf(x);
    float s = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    h = __builtin_fmaf(h, e, h);
    s = __builtin_fmaf(s, e, s);
    const float d = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(d, h, s);
    *half_rsq = h;
    return s;
}
__device__ __forceinline__ float invsqrt_b(float x)  // 1 / sqrt(x) with both roundings: sqrt_b, then the reciprocal seeded by 2h (no v_rcp)
{
    float h;
    const float s = sqrt_b(x, &h);
    float r = h + h;
    float e = __builtin_fmaf(-s, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-s, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float invsqrt_c(float x)  // as b with one more refinement of the reciprocal
{
    float h;
    const float s = sqrt_b(x, &h);
    float r = h + h;
    float e = __builtin_fmaf(-s, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-s, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-s, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

__global__ __launch_bounds__(256) void k(unsigned long long* __restrict__ out, int elo, int ehi)
{
    const uint32_t lo = (uint32_t)(127 + elo) << 23, hi = (uint32_t)(127 + ehi) << 23;
    unsigned long long bad[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (uint64_t b = (uint64_t)lo + (uint64_t)blockIdx.x * 256 + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * 256) {
        const float x = __uint_as_float((uint32_t)b);
        const uint32_t rr = __float_as_uint(1.0f / x), rn = __float_as_uint(1.0f / -x), sr = __float_as_uint(sqrtf(x)), ir = __float_as_uint(1.0f / sqrtf(x));
        bad[0] += (__float_as_uint(rcp_a(x)) != rr) + (__float_as_uint(rcp_a(-x)) != rn);
        bad[1] += (__float_as_uint(rcp_b(x)) != rr) + (__float_as_uint(rcp_b(-x)) != rn);
        bad[2] += (__float_as_uint(rcp_c(x)) != rr) + (__float_as_uint(rcp_c(-x)) != rn);
        bad[3] += __float_as_uint(sqrt_a(x)) != sr;
        float h;
        bad[4] += __float_as_uint(sqrt_b(x, &h)) != sr;
        bad[5] += __float_as_uint(invsqrt_b(x)) != ir;
        bad[6] += __float_as_uint(invsqrt_c(x)) != ir;
        bad[7] += 1;
    }
    for (int i = 0; i < 8; i++) atomicAdd(&out[i], bad[i]);
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 64);
    const int ranges[3][2] = { { -60, 60 }, { -12, 12 }, { -6, 2 } };
    const char* names[8] = { "rcp + 6 fma (product)", "rcp + 4 fma", "rcp + 2 fma", "sqrt + neighbour select (product)", "rsq + 2 mul + 5 fma", "1/sqrt: rsq-seeded, 2 steps", "1/sqrt: rsq-seeded, 3 steps", "values" };
    for (auto& r : ranges) {
        hipMemset(d, 0, 64);
        hipLaunchKernelGGL(k, dim3(8192), dim3(256), 0, 0, d, r[0], r[1]);
        unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("exponents [%d, %d):\n", r[0], r[1]);
        for (int i = 0; i < 8; i++) printf("  %-36s %llu\n", names[i], h[i]);
    }
    return 0;
}
