// Scratch microbenchmark: what does this part's memory system take for a WRITE-ONLY stream of the tile pass's size and shape?
// The tile pass writes 28 B per pixel (929 MB at 8K) and reads little from HBM; its HBM "floor" had been priced at 8 TB/s, the
// figure for reads and writes together.  Variants:
//   linear / linear_nt : every lane stores 16 contiguous bytes, grid-strided - the friendliest possible fill of 929 MB;
//   tiles / tiles_nt   : the tile pass's own pattern - one 256-thread workgroup per 64x64 tile with the tile pass's LDS
//                        footprint (4 workgroups per CU), a lane owns a column of four pixels, per pixel three 4-byte and two
//                        8-byte stores into five row-major planes of a 7680x4320 target (256 / 512 contiguous bytes per wave);
//   tiles_window       : the same stores folded into a 256-KB window of each plane (what VR_EXP_STOREWIN measured).
// hipcc -O3 --offload-arch=gfx950 fill_rate.hip -o fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void k_linear(u4* out, size_t n16)
{
    const u4 v = { 1u, 2u, 3u, 4u };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

template <bool NT, bool WINDOW>
__global__ __launch_bounds__(256) void k_tiles(char* base, int w, int h, int tiles_x)
{
    __shared__ unsigned long long vis[64 * 64 + 400];          // the tile pass's LDS footprint: four workgroups per CU
    const int tid = threadIdx.x;
    if (w < 0) { vis[tid] = tid; __syncthreads(); base[0] = (char)vis[tid ^ 1]; }
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int ox = tx * 64, oy = ty * 64;
    const size_t px = (size_t)w * h;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)(uint32_t)(px * 28), 0x00020000);
    const int o_diff = (int)(px * 4), o_spec = (int)(px * 8), o_nrm = (int)(px * 12), o_emi = (int)(px * 20);
    constexpr int aux = NT ? 2 : 0;
    for (int g = tid; g < 64 * 64 / 4; g += 256) {
        const int lx = g & 63, ly0 = (g >> 6) * 4;
        const int gx = ox + lx, gy0 = oy + ly0;
        if (gx >= w || gy0 >= h) continue;
        uint32_t pix4 = ((uint32_t)gy0 * (uint32_t)w + (uint32_t)gx) << 2;
#pragma unroll
        for (int k = 0; k < 4; k++, pix4 += (uint32_t)w << 2) {
            if (gy0 + k >= h) continue;
            const uint32_t p4 = WINDOW ? (pix4 & 0x3fffcu) : pix4, p8 = p4 + p4;
            const u2 nv = { (uint32_t)g, (uint32_t)k }, zv = { 0u, 0u };
            __builtin_amdgcn_raw_buffer_store_b32((uint32_t)g, r, p4, 0, aux);
            __builtin_amdgcn_raw_buffer_store_b32((uint32_t)k, r, p4, o_diff, aux);
            __builtin_amdgcn_raw_buffer_store_b32(7u, r, p4, o_spec, aux);
            __builtin_amdgcn_raw_buffer_store_b64(nv, r, p8, o_nrm, aux);
            __builtin_amdgcn_raw_buffer_store_b64(zv, r, p8, o_emi, aux);
        }
    }
}

template <typename F>
static void time_it(const char* name, double bytes, F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    float best = 1e9f, sum = 0.0f;
    for (int rep = 0; rep < 5; rep++) {
        float ms;
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        ms /= 10.0f; sum += ms; if (ms < best) best = ms;
    }
    printf("%-14s best %.1f us (%.2f TB/s)   mean %.1f us (%.2f TB/s)\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12, sum / 5 * 1e3,
           bytes / (sum / 5 * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
    const int w = 7680, h = 4320;
    const size_t px = (size_t)w * h, bytes = px * 28;
    char* buf; hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    const int tiles_x = w / 64, tiles = tiles_x * ((h + 63) / 64);
    time_it("linear", (double)bytes, [&] { hipLaunchKernelGGL(k_linear<false>, dim3(256 * 16), dim3(256), 0, 0, (u4*)buf, bytes / 16); });
    time_it("linear_nt", (double)bytes, [&] { hipLaunchKernelGGL(k_linear<true>, dim3(256 * 16), dim3(256), 0, 0, (u4*)buf, bytes / 16); });
    time_it("tiles", (double)bytes, [&] { hipLaunchKernelGGL((k_tiles<false, false>), dim3(tiles), dim3(256), 0, 0, buf, w, h, tiles_x); });
    time_it("tiles_nt", (double)bytes, [&] { hipLaunchKernelGGL((k_tiles<true, false>), dim3(tiles), dim3(256), 0, 0, buf, w, h, tiles_x); });
    time_it("tiles_window", (double)bytes, [&] { hipLaunchKernelGGL((k_tiles<false, true>), dim3(tiles), dim3(256), 0, 0, buf, w, h, tiles_x); });
    time_it("tiles_win_nt", (double)bytes, [&] { hipLaunchKernelGGL((k_tiles<true, true>), dim3(tiles), dim3(256), 0, 0, buf, w, h, tiles_x); });
    hipFree(buf);
    return 0;
}
