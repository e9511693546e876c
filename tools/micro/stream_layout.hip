// Scratch microbenchmark: does a lighting-pass-like stream run faster over an interleaved G-buffer than over five planes?
// planes   : per 4 pixels a lane loads 16 B from each of depth/diffuse/specular (4 B/px) and 2 x 16 B from normals/emissive
//            (8 B/px), seven streams, and stores 2 x 16 B (8 B/px).
// interleaved: the same 112 B per lane come from one record of 7 x 16 B; records of a wave are contiguous (7 KiB per wave).
// hipcc -O3 --offload-arch=gfx950 stream_layout.hip -o stream_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k_planes(const uint4* d, const uint4* a, const uint4* s, const uint4* n, const uint4* e, uint4* out, size_t quads)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const uint4 v0 = d[q], v1 = a[q], v2 = s[q], v3 = n[2 * q], v4 = n[2 * q + 1], v5 = e[2 * q], v6 = e[2 * q + 1];
    uint4 r0 = make_uint4(v0.x ^ v1.x ^ v3.x, v0.y ^ v2.y ^ v4.y, v0.z ^ v5.z, v0.w ^ v6.w);
    uint4 r1 = make_uint4(v1.w + v2.x, v3.y + v4.z, v5.x + v6.y, v2.w);
    out[2 * q] = r0; out[2 * q + 1] = r1;
}
// record layout per wave: [plane-major within the wave] 7 slabs of 64 x 16 B, so each load instruction of a wave is 1 KiB contiguous
__global__ __launch_bounds__(256) void k_interleaved(const uint4* g, uint4* out, size_t quads)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const size_t wave = q >> 6, lane = q & 63;
    const uint4* base = g + wave * (7 * 64) + lane;
    const uint4 v0 = base[0], v1 = base[64], v2 = base[128], v3 = base[192], v4 = base[256], v5 = base[320], v6 = base[384];
    uint4 r0 = make_uint4(v0.x ^ v1.x ^ v3.x, v0.y ^ v2.y ^ v4.y, v0.z ^ v5.z, v0.w ^ v6.w);
    uint4 r1 = make_uint4(v1.w + v2.x, v3.y + v4.z, v5.x + v6.y, v2.w);
    out[2 * q] = r0; out[2 * q + 1] = r1;
}

int main()
{
    const size_t px = (size_t)7680 * 4320, quads = px / 4;
    uint4 *d, *a, *s, *n, *e, *g, *out;
    hipMalloc(&d, px * 4); hipMalloc(&a, px * 4); hipMalloc(&s, px * 4); hipMalloc(&n, px * 8); hipMalloc(&e, px * 8);
    hipMalloc(&g, px * 28); hipMalloc(&out, px * 8);
    hipMemset(d, 1, px * 4); hipMemset(a, 2, px * 4); hipMemset(s, 3, px * 4); hipMemset(n, 4, px * 8); hipMemset(e, 5, px * 8); hipMemset(g, 6, px * 28);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)((quads + 255) / 256);
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_planes, dim3(grid), dim3(256), 0, 0, d, a, s, n, e, out, quads);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("planes      %.1f us  %.2f TB/s\n", ms * 100.0f, px * 36.0 / (ms * 1e-4) / 1e12);
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_interleaved, dim3(grid), dim3(256), 0, 0, g, out, quads);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("interleaved %.1f us  %.2f TB/s\n", ms * 100.0f, px * 36.0 / (ms * 1e-4) / 1e12);
    }
    return 0;
}
