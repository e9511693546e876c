// Texture upload + mip-chain generation on the device, and the synthetic input
// generators (the reference's media/ directory is git-ignored: .gitignore:36,52).
#include "vr_internal.h"
#include "vr_tex_dev.h"

#include <string.h>

// 2x2 box downsample of an R8_UNORM level: round(avg * 255) == (sum + 2) >> 2.
__global__ void k_mip_r8(const uint8_t* __restrict__ src, int sw, int sh, uint8_t* __restrict__ dst, int dw, int dh)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh) return;
    int x0 = 2 * x, x1 = min(2 * x + 1, sw - 1), y0 = 2 * y, y1 = min(2 * y + 1, sh - 1);
    int sum = src[y0 * sw + x0] + src[y0 * sw + x1] + src[y1 * sw + x0] + src[y1 * sw + x1];
    dst[y * dw + x] = (uint8_t)((sum + 2) >> 2);
}

// SRGBA8 level: decode to linear, average, re-encode (what a blit into an SRGBA8
// render target does); alpha is linear.
__global__ void k_mip_srgba8(const uint32_t* __restrict__ src, int sw, int sh, uint32_t* __restrict__ dst, int dw, int dh,
                             const float* __restrict__ lut_g, const float* __restrict__ thr_g)
{
    __shared__ float lut[256];
    __shared__ float thr[256];
    int tid = threadIdx.y * blockDim.x + threadIdx.x;
    for (int i = tid; i < 256; i += blockDim.x * blockDim.y) { lut[i] = lut_g[i]; thr[i] = thr_g[i]; }
    __syncthreads();
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh) return;
    int x0 = 2 * x, x1 = min(2 * x + 1, sw - 1), y0 = 2 * y, y1 = min(2 * y + 1, sh - 1);
    uint32_t p00 = src[(size_t)y0 * sw + x0], p10 = src[(size_t)y0 * sw + x1];
    uint32_t p01 = src[(size_t)y1 * sw + x0], p11 = src[(size_t)y1 * sw + x1];
    uint32_t o = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int sh8 = 8 * c;
        float a = (lut[(p00 >> sh8) & 255u] + lut[(p10 >> sh8) & 255u]) + (lut[(p01 >> sh8) & 255u] + lut[(p11 >> sh8) & 255u]);
        o |= vr_srgb_encode(a * 0.25f, thr) << sh8;
    }
    uint32_t al = ((p00 >> 24) + (p10 >> 24) + (p01 >> 24) + (p11 >> 24) + 2u) >> 2;
    dst[(size_t)y * dw + x] = o | (al << 24);
}

__global__ void k_build_quads(const uint8_t* __restrict__ src, int w, int h, uint32_t* __restrict__ dst)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ix >= w + 2 || iy >= h + 2) return;
    const int x0 = min(max(ix - 1, 0), w - 1), x1 = min(max(ix, 0), w - 1);
    const int y0 = min(max(iy - 1, 0), h - 1), y1 = min(max(iy, 0), h - 1);
    dst[(size_t)iy * (w + 2) + ix] = (uint32_t)src[y0 * w + x0] | ((uint32_t)src[y0 * w + x1] << 8)
                                   | ((uint32_t)src[y1 * w + x0] << 16) | ((uint32_t)src[y1 * w + x1] << 24);
}

__global__ void k_build_quads_f32(const uint8_t* __restrict__ src, int w, int h, float4* __restrict__ dst)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ix >= w + 2 || iy >= h + 2) return;
    const int x0 = min(max(ix - 1, 0), w - 1), x1 = min(max(ix, 0), w - 1);
    const int y0 = min(max(iy - 1, 0), h - 1), y1 = min(max(iy, 0), h - 1);
    // UNORM8 -> float, correctly rounded, and the differences the bilinear filter forms first
    const float t00 = (float)src[y0 * w + x0] / 255.0f, t10 = (float)src[y0 * w + x1] / 255.0f;
    const float t01 = (float)src[y1 * w + x0] / 255.0f, t11 = (float)src[y1 * w + x1] / 255.0f;
    dst[(size_t)iy * (w + 2) + ix] = make_float4(t00, t10 - t00, t01, t11 - t01);
}

// Tables of the tile pass's fast variant (DevTex::fast): (w+3) x (h+3) clamp-addressed entries per level.
__global__ void k_build_fast_r8(const uint8_t* __restrict__ src, int w, int h, float4* __restrict__ dst)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ix >= w + 3 || iy >= h + 3) return;
    const int x0 = min(max(ix - 1, 0), w - 1), x1 = min(max(ix, 0), w - 1);
    const int y0 = min(max(iy - 1, 0), h - 1), y1 = min(max(iy, 0), h - 1);
    const float t00 = (float)src[y0 * w + x0] / 255.0f, t10 = (float)src[y0 * w + x1] / 255.0f;
    const float t01 = (float)src[y1 * w + x0] / 255.0f, t11 = (float)src[y1 * w + x1] / 255.0f;
    dst[(size_t)iy * (w + 3) + ix] = make_float4(t00, t10 - t00, t01, t11 - t01);
}
__global__ void k_build_fast_srgba8(const uint32_t* __restrict__ src, int w, int h, const float* __restrict__ lut, float4* __restrict__ dst)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ix >= w + 3 || iy >= h + 3) return;
    const uint32_t p = src[(size_t)min(max(iy - 1, 0), h - 1) * w + min(max(ix - 1, 0), w - 1)];
    dst[(size_t)iy * (w + 3) + ix] = make_float4(lut[p & 255u], lut[(p >> 8) & 255u], lut[(p >> 16) & 255u], 0.0f);
}

__global__ void k_decode_rgbf(const uint32_t* __restrict__ src, size_t n, const float* __restrict__ lut, float* __restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t p = src[i];
        reinterpret_cast<float4*>(dst)[i] = make_float4(lut[p & 255u], lut[(p >> 8) & 255u], lut[(p >> 16) & 255u], 0.0f);
    }
}

int vr_tex_upload_and_mip(vr_context* ctx, const uint8_t* host, int w, int h, int tb, DevTex* out, uint8_t** out_mem, uint64_t* out_bytes)
{
    VR_REQUIRE(host && w > 0 && h > 0 && w <= 16384 && h <= 16384, "bad texture");
    // The tile pass reads DECODED copies (16 B per heightmap footprint and per albedo texel) through 32-bit byte offsets:
    // a chain and its decoded tables must stay below 2 GB, which 8192 x 8192 texels (2^26) still do and more do not.
    VR_REQUIRE((size_t)w * (size_t)h <= ((size_t)1 << 26), "textures above 2^26 texels (8192 x 8192) are not supported (decoded tables are addressed with 32-bit offsets)");
    int levels = 1; { int m = w > h ? w : h; while (m > 1) { m >>= 1; levels++; } }
    VR_REQUIRE(levels <= kMaxLevels, "too many mip levels");
    uint32_t off[kMaxLevels] = { 0 };
    size_t total = 0;
    for (int l = 0; l < levels; l++) {
        int lw = (w >> l) > 1 ? (w >> l) : 1, lh = (h >> l) > 1 ? (h >> l) : 1;
        off[l] = (uint32_t)total;
        total += ((size_t)lw * lh * tb + 255) / 256 * 256;
    }
    size_t table_off = total;
    total += sizeof(uint32_t) * kMaxLevels;
    // quad (bilinear footprint) tables for R8 textures
    uint32_t qoff[kMaxLevels] = { 0 };
    size_t qtable_off = 0, quad_off = 0, quad_dwords = 0, quadf_off = 0;
    if (tb == 1) {
        for (int l = 0; l < levels; l++) {
            int lw = (w >> l) > 1 ? (w >> l) : 1, lh = (h >> l) > 1 ? (h >> l) : 1;
            qoff[l] = (uint32_t)quad_dwords;
            quad_dwords += ((size_t)(lw + 2) * (lh + 2) + 63) / 64 * 64;
        }
        qtable_off = total; total += sizeof(uint32_t) * kMaxLevels;
        total = (total + 255) / 256 * 256;
        quad_off = total; total += quad_dwords * sizeof(uint32_t);
        total = (total + 255) / 256 * 256;
        quadf_off = total; total += quad_dwords * sizeof(float4);
    }
    size_t rgbf_off = 0;
    if (tb == 4) { total = (total + 255) / 256 * 256; rgbf_off = total; total += table_off / 4 * 16; }   // decoded copy of the whole chain
    // tables of the tile pass's fast variant: (w+3) x (h+3) entries of 16 B per level, levels 256-B aligned
    uint32_t fast_lv[kMaxLevels][4];
    size_t fast_entry[kMaxLevels];
    memset(fast_lv, 0, sizeof(fast_lv));
    total = (total + 255) / 256 * 256;
    const size_t fastlv_off = total; total += sizeof(fast_lv);
    total = (total + 255) / 256 * 256;
    const size_t fast_off = total;
    size_t fast_bytes = 0;
    for (int l = 0; l < levels; l++) {
        const int lw = (w >> l) > 1 ? (w >> l) : 1, lh = (h >> l) > 1 ? (h >> l) : 1;
        fast_entry[l] = fast_bytes / 16;
        const uint32_t row = (uint32_t)(lw + 3) * 16u;
        const float wf = (float)lw, hf = (float)lh;
        fast_lv[l][0] = (uint32_t)fast_bytes + row + 16u; fast_lv[l][1] = row;
        memcpy(&fast_lv[l][2], &wf, 4); memcpy(&fast_lv[l][3], &hf, 4);
        fast_bytes += ((size_t)(lw + 3) * (lh + 3) * 16 + 255) / 256 * 256;
    }
    total += fast_bytes;
    // every table is read through its own buffer resource with 32-bit byte offsets: each must stay below 2 GB
    VR_REQUIRE(quad_dwords * sizeof(float4) < ((size_t)1 << 31) && table_off * 4 < ((size_t)1 << 31) && fast_bytes < ((size_t)1 << 31),
               "texture too large");

    uint8_t* mem = nullptr;
    VR_HIP(hipMalloc(&mem, total));
    if (out_bytes) *out_bytes += total;
    hipStream_t s = ctx->stream;
    // on any failure below: wait for the copies that read this function's stack arrays, free, report
#define VR_TEX_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipStreamSynchronize(s); (void)hipFree(mem); \
        vr_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
        return e_ == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP; } } while (0)
    VR_TEX_TRY(hipMemcpyAsync(mem, host, (size_t)w * h * tb, hipMemcpyHostToDevice, s));
    VR_TEX_TRY(hipMemcpyAsync(mem + table_off, off, sizeof(off), hipMemcpyHostToDevice, s));
    int sw = w, sh = h;
    for (int l = 1; l < levels; l++) {
        int dw = sw > 1 ? sw >> 1 : 1, dh = sh > 1 ? sh >> 1 : 1;
        dim3 blk(32, 8), grd((dw + 31) / 32, (dh + 7) / 8);
        if (tb == 1) hipLaunchKernelGGL(k_mip_r8, grd, blk, 0, s, mem + off[l - 1], sw, sh, mem + off[l], dw, dh);
        else hipLaunchKernelGGL(k_mip_srgba8, grd, blk, 0, s, (const uint32_t*)(mem + off[l - 1]), sw, sh,
                                (uint32_t*)(mem + off[l]), dw, dh, ctx->d_srgb_lut, ctx->d_srgb_thr);
        sw = dw; sh = dh;
    }
    out->quad = nullptr; out->qoff = nullptr; out->quadf = nullptr; out->rgbf = nullptr;
    VR_TEX_TRY(hipMemcpyAsync(mem + fastlv_off, fast_lv, sizeof(fast_lv), hipMemcpyHostToDevice, s));
    for (int l = 0; l < levels; l++) {
        const int lw = (w >> l) > 1 ? (w >> l) : 1, lh = (h >> l) > 1 ? (h >> l) : 1;
        dim3 blk(32, 8), grd((lw + 3 + 31) / 32, (lh + 3 + 7) / 8);
        float4* dst = (float4*)(mem + fast_off) + fast_entry[l];
        if (tb == 1) hipLaunchKernelGGL(k_build_fast_r8, grd, blk, 0, s, mem + off[l], lw, lh, dst);
        else hipLaunchKernelGGL(k_build_fast_srgba8, grd, blk, 0, s, (const uint32_t*)(mem + off[l]), lw, lh, ctx->d_srgb_lut, dst);
    }
    out->fast = (const float4*)(mem + fast_off); out->fast_lv = (const uint4*)(mem + fastlv_off); out->fast_bytes = (uint32_t)fast_bytes; out->pad2 = 0;
    if (tb == 4) {       // (padding texels between levels are decoded too; nothing reads them)
        hipLaunchKernelGGL(k_decode_rgbf, dim3(4096), dim3(256), 0, s, (const uint32_t*)mem, table_off / 4, ctx->d_srgb_lut, (float*)(mem + rgbf_off));
        out->rgbf = (const float*)(mem + rgbf_off);
    }
    if (tb == 1) {
        VR_TEX_TRY(hipMemcpyAsync(mem + qtable_off, qoff, sizeof(qoff), hipMemcpyHostToDevice, s));
        for (int l = 0; l < levels; l++) {
            int lw = (w >> l) > 1 ? (w >> l) : 1, lh = (h >> l) > 1 ? (h >> l) : 1;
            dim3 blk(32, 8), grd((lw + 2 + 31) / 32, (lh + 2 + 7) / 8);
            hipLaunchKernelGGL(k_build_quads, grd, blk, 0, s, mem + off[l], lw, lh, (uint32_t*)(mem + quad_off) + qoff[l]);
            hipLaunchKernelGGL(k_build_quads_f32, grd, blk, 0, s, mem + off[l], lw, lh, (float4*)(mem + quadf_off) + qoff[l]);
        }
        out->quadf = (const float4*)(mem + quadf_off);
        out->quad = (const uint32_t*)(mem + quad_off); out->qoff = (const uint32_t*)(mem + qtable_off);
    }
    VR_TEX_TRY(hipGetLastError());
    VR_TEX_TRY(hipStreamSynchronize(s));   // host source buffer is caller-owned: finish the copy before returning
#undef VR_TEX_TRY
    out->base = mem; out->off = (const uint32_t*)(mem + table_off); out->levels = levels; out->w0 = w; out->h0 = h;
    out->chain_bytes = (uint32_t)table_off; out->quad_bytes = (uint32_t)(quad_dwords * sizeof(uint32_t)); out->pad = 0;
    *out_mem = mem;
    return VR_OK;
}

// ---- synthetic inputs: integer-hash value-noise fBm (5 octaves) -> R8; height-banded
// albedo + per-texel noise -> SRGBA8.  Integer arithmetic only. ------------------------
__device__ __forceinline__ uint32_t vr_hash32(uint32_t x, uint32_t y, uint32_t s)
{
    uint32_t h = (x * 0x9E3779B1u) ^ (y * 0x85EBCA77u) ^ (s * 0xC2B2AE3Du);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
__device__ uint32_t vr_value_noise16(uint32_t x, uint32_t y, uint32_t period, uint32_t seed)
{
    uint32_t ix = x / period, iy = y / period;
    uint32_t fx = ((x % period) << 16) / period, fy = ((y % period) << 16) / period;
    uint64_t sx = ((((uint64_t)fx * fx) >> 16) * (uint64_t)(3u * 65536u - 2u * fx)) >> 16;
    uint64_t sy = ((((uint64_t)fy * fy) >> 16) * (uint64_t)(3u * 65536u - 2u * fy)) >> 16;
    uint64_t h00 = vr_hash32(ix, iy, seed) >> 16, h10 = vr_hash32(ix + 1, iy, seed) >> 16;
    uint64_t h01 = vr_hash32(ix, iy + 1, seed) >> 16, h11 = vr_hash32(ix + 1, iy + 1, seed) >> 16;
    uint64_t top = (h00 * (65536u - sx) + h10 * sx) >> 16;
    uint64_t bot = (h01 * (65536u - sx) + h11 * sx) >> 16;
    return (uint32_t)((top * (65536u - sy) + bot * sy) >> 16);
}
__global__ void k_synth_height(int size, uint32_t seed, uint8_t* out)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= size || y >= size) return;
    const uint32_t wgt[5] = { 16, 8, 4, 2, 1 };
    uint32_t total = 0;
#pragma unroll
    for (int o = 0; o < 5; o++) {
        uint32_t period = (uint32_t)size >> (2 + o); if (period < 1) period = 1;
        total += wgt[o] * vr_value_noise16((uint32_t)x, (uint32_t)y, period, seed + (uint32_t)o);
    }
    uint32_t v16 = total / 31u;
    int32_t s = ((int32_t)v16 - 9000) * 3 / 2;
    s = s < 0 ? 0 : (s > 65535 ? 65535 : s);
    uint32_t h16 = ((uint32_t)s * (uint32_t)s) >> 16;
    out[(size_t)y * size + x] = (uint8_t)(h16 >> 8);
}
__global__ void k_synth_albedo(int size, uint32_t seed, const uint8_t* height, uint32_t* out)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= size || y >= size) return;
    const int32_t hs[6] = { 0, 20, 40, 110, 180, 255 };
    const int32_t cs[6][3] = { { 40, 70, 110 }, { 60, 90, 120 }, { 180, 165, 120 }, { 70, 120, 50 }, { 110, 100, 90 }, { 235, 235, 240 } };
    int32_t hgt = height[(size_t)y * size + x];
    int seg = 0;
    while (seg < 4 && hgt >= hs[seg + 1]) seg++;
    int32_t h0 = hs[seg], h1 = hs[seg + 1];
    uint32_t n = vr_hash32((uint32_t)x, (uint32_t)y, seed) & 255u;
    uint32_t o = 0xff000000u;
    for (int c = 0; c < 3; c++) {
        int32_t v = (cs[seg][c] * (h1 - hgt) + cs[seg + 1][c] * (hgt - h0)) / (h1 - h0);
        v += (int32_t)(n >> 4) - 8;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        o |= (uint32_t)v << (8 * c);
    }
    out[(size_t)y * size + x] = o;
}

extern "C" VR_API int vr_synth_heightmap(vr_context* ctx, int32_t size, uint32_t seed, uint8_t* out)
{
    VR_REQUIRE(ctx && out && size >= 4 && size <= 16384, "bad arguments");
    VR_HIP(hipSetDevice(ctx->device));
    uint8_t* d = nullptr;
    VR_HIP(hipMalloc(&d, (size_t)size * size));
    dim3 blk(32, 8), grd((size + 31) / 32, (size + 7) / 8);
    hipLaunchKernelGGL(k_synth_height, grd, blk, 0, ctx->stream, size, seed, d);
    hipError_t e = hipMemcpyAsync(out, d, (size_t)size * size, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    VR_HIP(e);
    return VR_OK;
}
extern "C" VR_API int vr_synth_albedo(vr_context* ctx, int32_t size, uint32_t seed, const uint8_t* height, uint8_t* out)
{
    VR_REQUIRE(ctx && out && height && size >= 4 && size <= 16384, "bad arguments");
    VR_HIP(hipSetDevice(ctx->device));
    uint8_t* dh = nullptr; uint32_t* dc = nullptr;
    VR_HIP(hipMalloc(&dh, (size_t)size * size));
    hipError_t e = hipMalloc(&dc, (size_t)size * size * 4);
    if (e != hipSuccess) { (void)hipFree(dh); VR_HIP(e); }
    e = hipMemcpyAsync(dh, height, (size_t)size * size, hipMemcpyHostToDevice, ctx->stream);
    dim3 blk(32, 8), grd((size + 31) / 32, (size + 7) / 8);
    hipLaunchKernelGGL(k_synth_albedo, grd, blk, 0, ctx->stream, size, seed, dh, dc);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dc, (size_t)size * size * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dh); (void)hipFree(dc);
    VR_HIP(e);
    return VR_OK;
}

// ---- test helper ---------------------------------------------------------------------------
__global__ void k_debug_srgb_encode(const float* __restrict__ in, size_t n, uint8_t* __restrict__ out, const float* __restrict__ thr_g,
                                    const uint8_t* __restrict__ tab_g)
{
    __shared__ float thr[kThrTabSize];
    __shared__ uint8_t tab[kEncTabSize];
    thr[threadIdx.x] = thr_g[threadIdx.x];
    if (threadIdx.x == 0) thr[256] = __uint_as_float(0x7fc00000u);   // NaN: no x is >= it, not even +inf
    for (int i = threadIdx.x; i < kEncTabSize; i += blockDim.x) tab[i] = tab_g[i];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (uint8_t)vr_srgb_encode_fast(in[i], thr, tab);
}
extern "C" VR_API int vr_debug_srgb_encode(vr_context* ctx, const float* in, size_t n, uint8_t* out)
{
    VR_REQUIRE(ctx && in && out && n > 0, "bad arguments");
    VR_HIP(hipSetDevice(ctx->device));
    float* din = nullptr; uint8_t* dout = nullptr;
    VR_HIP(hipMalloc(&din, n * sizeof(float)));
    hipError_t e = hipMalloc(&dout, n);
    if (e != hipSuccess) { (void)hipFree(din); VR_HIP(e); }
    e = hipMemcpyAsync(din, in, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    hipLaunchKernelGGL(k_debug_srgb_encode, dim3(2048), dim3(256), 0, ctx->stream, din, n, dout, ctx->d_srgb_thr, ctx->d_enc_tab);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(din); (void)hipFree(dout);
    VR_HIP(e);
    return VR_OK;
}

// Sweeps every float with an exponent in [-60, 60) (both signs for the reciprocal) and counts the values for which the
// tile pass's short sequences differ from 1.0f / x and sqrtf(x).  out[0] = reciprocal mismatches, out[1] = square-root
// mismatches, out[2] = values tested.
__global__ __launch_bounds__(256) void k_debug_fastmath(unsigned long long* __restrict__ out)
{
    const uint32_t lo = (uint32_t)(127 - 60) << 23, hi = (uint32_t)(127 + 60) << 23;
    unsigned long long bad_r = 0, bad_s = 0, n = 0;
    for (uint64_t b = (uint64_t)lo + (uint64_t)blockIdx.x * 256 + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * 256) {
        const float x = __uint_as_float((uint32_t)b);
        bad_r += __float_as_uint(vr_rcp_exact(x)) != __float_as_uint(1.0f / x);
        bad_r += __float_as_uint(vr_rcp_exact(-x)) != __float_as_uint(1.0f / -x);
        bad_s += __float_as_uint(vr_sqrt_exact(x)) != __float_as_uint(sqrtf(x));
        n++;
    }
    atomicAdd(&out[0], bad_r); atomicAdd(&out[1], bad_s); atomicAdd(&out[2], n);
}
extern "C" VR_API int vr_debug_fastmath_check(vr_context* ctx, unsigned long long out[3])
{
    VR_REQUIRE(ctx && out, "NULL argument");
    VR_HIP(hipSetDevice(ctx->device));
    unsigned long long* d = nullptr;
    VR_HIP(hipMalloc(&d, 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_debug_fastmath, dim3(8192), dim3(256), 0, ctx->stream, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    VR_HIP(e);
    return VR_OK;
}
