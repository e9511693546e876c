// Tone mapping HdrColor -> LdrColor.
//
// Replaces donut::render::ToneMappingPass as the reference uses it: default CreateParameters
// (Renderer.cpp:256-257), AdvanceFrame (:188-189), SimpleRender(cmd, ToneMappingParameters(), view,
// HdrColor) into the SRGBA8 LdrColor target (:430-431, Renderer.h:81-95).  Donut's source is absent
// from the reference checkout; the stages restate it from recollection:
// luminance histogram -> percentile-windowed average log luminance -> eye
// adaptation -> extended Reinhard on luminance -> sRGB encode.
//
// All three streaming kernels are bandwidth kernels (8 B/px in for the histogram, 8 in + 4 out for the
// tone map; 6 in + 3 out on packed multi-GPU tiles).  Arithmetic is fp32 in the oracle's order, no
// FMA (this file is built with -ffp-contract=off like the rest), log2/exp2 are the pinned
// polynomials, so histogram bins and LDR bytes are bit-exact against the oracle.
#include "vr_internal.h"
#include "vr_tex_dev.h"

#include <math.h>
#include <string.h>

struct vr_tonemap {
    vr_context* ctx;
    uint32_t* d_hist;        // VR_TONEMAP_BINS
    float* d_exposure;       // [0] adapted luminance
};

vr_context* vr_tonemap_context(vr_tonemap* tm) { return tm->ctx; }

struct TmArgs {
    int w, h, tiles_x;
    float scale, bias;                       // log-luminance -> [0, 1]
    float exposure_scale, wp_inv2, min_adapted;
};

__device__ __forceinline__ float tm_luminance(float r, float g, float b) { return (r * 0.2126f + g * 0.7152f) + b * 0.0722f; }

__device__ __forceinline__ float tm_log2_pinned(float x)
{
    const uint32_t bits = __float_as_uint(x);
    const int e = (int)((bits >> 23) & 255u) - 127;
    const float tt = __uint_as_float((bits & 0x7fffffu) | 0x3f800000u) - 1.0f;
    const float p = tt * (1.4208646f + tt * (-0.57725066f + tt * 0.1563861f));
    return (float)e + p;
}

__device__ __forceinline__ float tm_exp2_pinned(float x)
{
    if (!(x == x)) return 0.0f;
    if (x < -126.0f) x = -126.0f;
    if (x > 127.0f) x = 127.0f;
    const float n = floorf(x), f = x - n;
    const float p = 1.0f + f * (0.69583356f + f * (0.22606716f + f * 0.07809929f));
    return p * __uint_as_float((uint32_t)((int)n + 127) << 23);
}

// One pixel into the workgroup's LDS histogram.  Neighbouring pixels have similar luminance, so a wave's
// 64 atomics would pile onto one or two addresses: s_hist is the lane's own copy out of kHistCopies (two lanes of a
// wave per copy), and the copies are kHistStride = bins + 1 words apart so that equal bins of different copies fall
// on different LDS banks.  The grid is at most kHistBlocks workgroups: every workgroup ends with one global atomic per
// bin, and 4096 of them cost more than binning an eighth of an 8K frame (27 us of the 28 a rank's share took).
constexpr int kHistCopies = 32, kHistStride = VR_TONEMAP_BINS + 1, kHistBlocks = 1024;
__device__ __forceinline__ void tm_bin_pixel(uint32_t* __restrict__ s_hist, const TmArgs& a, float r, float g, float b)
{
    const float lum = tm_luminance(r, g, b);
    const uint32_t lb = __float_as_uint(lum);
    float t;
    if (!(lum > 0.0f)) t = 0.0f;
    else if ((lb >> 23) == 0u) t = 0.0f;
    else if ((lb >> 23) == 255u) t = 1.0f;
    else {
        t = tm_log2_pinned(lum) * a.scale + a.bias;
        t = !(t > 0.0f) ? 0.0f : (t > 1.0f ? 1.0f : t);
    }
    const float hb = t * (float)(VR_TONEMAP_BINS - 1);
    const float lf = floorf(hb);
    const int left = (int)lf;
    const uint32_t rw = (uint32_t)((hb - lf) * 64.0f), lw = 64u - rw;
    if (lw != 0u && left < VR_TONEMAP_BINS) atomicAdd(&s_hist[left], lw);
    if (rw != 0u && left + 1 < VR_TONEMAP_BINS) atomicAdd(&s_hist[left + 1], rw);
}

__device__ __forceinline__ void unpack_rgba16f_quad(const uint4& a, const uint4& b, float rgb[4][3])
{
    const uint32_t d[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        rgb[k][0] = vr_half_to_float(d[2 * k] & 0xffffu); rgb[k][1] = vr_half_to_float(d[2 * k] >> 16);
        rgb[k][2] = vr_half_to_float(d[2 * k + 1] & 0xffffu);
    }
}
// packed RGB16F quad: 24 B = r0 g0 | b0 r1 | g1 b1 | r2 g2 | b2 r3 | g3 b3
__device__ __forceinline__ void unpack_rgb16f_quad(const uint2& a, const uint2& b, const uint2& c, float rgb[4][3])
{
    const uint32_t d[6] = { a.x, a.y, b.x, b.y, c.x, c.y };
    const uint32_t h[12] = { d[0] & 0xffffu, d[0] >> 16, d[1] & 0xffffu, d[1] >> 16, d[2] & 0xffffu, d[2] >> 16,
                             d[3] & 0xffffu, d[3] >> 16, d[4] & 0xffffu, d[4] >> 16, d[5] & 0xffffu, d[5] >> 16 };
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c3 = 0; c3 < 3; c3++) rgb[k][c3] = vr_half_to_float(h[3 * k + c3]);
}

// Where a lane's 4 pixels live.  Row-major: quad index q.  Packed: block = (local tile, 8-row group), as in k_deferred.
struct QuadPos { int px0, py; size_t index; int valid; };     // index in pixels into the source / destination buffer
template <bool PACKED>
__device__ __forceinline__ QuadPos quad_pos(const TmArgs& a, const int32_t* __restrict__ owned_tiles, size_t block, int tid)
{
    QuadPos q;
    if (PACKED) {
        const int lt = (int)(block >> 4), rg = (int)(block & 15);
        const int tile = owned_tiles[lt];
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int row = rg * 8 + (tid >> 5), col = (tid & 31) * 4;
        q.px0 = tx * VR_OWNER_TILE + col; q.py = ty * VR_OWNER_TILE + row;
        q.index = ((size_t)lt * VR_OWNER_TILE + row) * VR_OWNER_TILE + col;
        q.valid = (q.px0 < a.w && q.py < a.h) ? min(4, a.w - q.px0) : 0;
    } else {
        const size_t p = (block * 256 + (size_t)tid) * 4;
        q.valid = p < (size_t)a.w * a.h ? 4 : 0;             // row-major vector path needs w % 4 == 0
        q.py = (int)(p / (size_t)a.w); q.px0 = (int)(p - (size_t)q.py * a.w);
        q.index = p;
    }
    return q;
}

template <bool PACKED>
__device__ __forceinline__ void load_quad(const void* __restrict__ src, size_t index, float rgb[4][3])
{
    if (PACKED) {
        const uint2* s = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(src) + index * 3);
        unpack_rgb16f_quad(s[0], s[1], s[2], rgb);
    } else {
        const uint4* s = reinterpret_cast<const uint4*>(reinterpret_cast<const uint2*>(src) + index);
        unpack_rgba16f_quad(s[0], s[1], rgb);
    }
}

// ---- AddFrameToHistogram ------------------------------------------------------------------
template <bool PACKED>
__global__ __launch_bounds__(256) void k_tm_histogram(TmArgs a, const void* __restrict__ src, const int32_t* __restrict__ owned_tiles,
                                                       size_t num_blocks, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_all[kHistCopies * kHistStride];
    for (int i = threadIdx.x; i < kHistCopies * kHistStride; i += 256) s_all[i] = 0u;
    uint32_t* s_hist = s_all + (threadIdx.x & (kHistCopies - 1)) * kHistStride;
    __syncthreads();
    for (size_t blk = blockIdx.x; blk < num_blocks; blk += gridDim.x) {
        const QuadPos q = quad_pos<PACKED>(a, owned_tiles, blk, (int)threadIdx.x);
        if (q.valid == 0) continue;
        float rgb[4][3];
        load_quad<PACKED>(src, q.index, rgb);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k < q.valid) tm_bin_pixel(s_hist, a, rgb[k][0], rgb[k][1], rgb[k][2]);
    }
    __syncthreads();
    uint32_t v = 0u;
#pragma unroll
    for (int c = 0; c < kHistCopies; c++) v += s_all[c * kHistStride + threadIdx.x];
    if (v != 0u) atomicAdd(&hist[threadIdx.x], v);
}

// any width: one pixel per lane, row-major RGBA16F
__global__ __launch_bounds__(256) void k_tm_histogram_scalar(TmArgs a, const uint2* __restrict__ src, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_all[kHistCopies * kHistStride];
    for (int i = threadIdx.x; i < kHistCopies * kHistStride; i += 256) s_all[i] = 0u;
    uint32_t* s_hist = s_all + (threadIdx.x & (kHistCopies - 1)) * kHistStride;
    __syncthreads();
    const size_t n = (size_t)a.w * a.h;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (size_t)gridDim.x * 256) {
        const uint2 v = src[p];
        tm_bin_pixel(s_hist, a, vr_half_to_float(v.x & 0xffffu), vr_half_to_float(v.x >> 16), vr_half_to_float(v.y & 0xffffu));
    }
    __syncthreads();
    uint32_t v = 0u;
#pragma unroll
    for (int c = 0; c < kHistCopies; c++) v += s_all[c * kHistStride + threadIdx.x];
    if (v != 0u) atomicAdd(&hist[threadIdx.x], v);
}

// ---- ComputeExposure: 256 lanes, one per bin; same arithmetic as the oracle's (exact integer prefix
// sums, stride-halving float reductions in a fixed order) ---------------------------------------
__global__ __launch_bounds__(256) void k_tm_exposure(const uint32_t* __restrict__ hist, float* __restrict__ exposure, float scale, float bias,
                                                      float low, float high, float min_log, float min_adapted, float max_adapted, float k_up,
                                                      float k_down, int has_up, int has_down)
{
    __shared__ unsigned long long pre[VR_TONEMAP_BINS];
    __shared__ float acc[VR_TONEMAP_BINS], wgt[VR_TONEMAP_BINS];
    const int i = threadIdx.x;
    pre[i] = hist[i];
    __syncthreads();
    for (int d = 1; d < VR_TONEMAP_BINS; d <<= 1) {          // inclusive scan (integers: any order gives the same sums)
        const unsigned long long v = i >= d ? pre[i - d] : 0ull;
        __syncthreads();
        pre[i] += v;
        __syncthreads();
    }
    const float ftotal = (float)pre[VR_TONEMAP_BINS - 1];
    const float lo = ftotal * low, hi = ftotal * high;
    const float below = i > 0 ? (float)pre[i - 1] : 0.0f, running = (float)pre[i];
    const float ca = running < lo ? lo : (running > hi ? hi : running);
    const float cb = below < lo ? lo : (below > hi ? hi : below);
    const float w = ca - cb;
    const float log_lum = ((float)i / (float)(VR_TONEMAP_BINS - 1) - bias) / scale;
    acc[i] = log_lum * w; wgt[i] = w;
    __syncthreads();
    for (int stride = VR_TONEMAP_BINS / 2; stride >= 1; stride >>= 1) {
        if (i < stride) { acc[i] = acc[i] + acc[i + stride]; wgt[i] = wgt[i] + wgt[i + stride]; }
        __syncthreads();
    }
    if (i != 0) return;
    const float accum = acc[0], wsum = wgt[0];
    const float avg_log = wsum > 0.0f ? accum / wsum : min_log;
    float target = tm_exp2_pinned(avg_log);
    if (target < min_adapted) target = min_adapted;
    if (target > max_adapted) target = max_adapted;
    const float old_lum = exposure[0];
    float out = target;
    if (old_lum > 0.0f) {
        const float diff = target - old_lum;
        const bool up = diff > 0.0f;
        if (up ? has_up : has_down) out = old_lum + diff * (up ? k_up : k_down);
    }
    exposure[0] = out;
}

// ---- Render: extended Reinhard on luminance, SRGBA8 out ------------------------------------
__device__ __forceinline__ uint32_t tm_pixel(const TmArgs& a, float inv_adapted, const float c[3], const float* __restrict__ thr,
                                             const uint8_t* __restrict__ enc)
{
    const float src = tm_luminance(c[0], c[1], c[2]);
    if (!(src > 0.0f)) return 0u;
    const float scaled = (a.exposure_scale * src) * inv_adapted;
    const float k = (scaled * (1.0f + scaled * a.wp_inv2)) / ((1.0f + scaled) * src);     // mapped / src as one division
    return vr_srgb_encode_fast(c[0] * k, thr, enc) | (vr_srgb_encode_fast(c[1] * k, thr, enc) << 8) | (vr_srgb_encode_fast(c[2] * k, thr, enc) << 16);
}

template <bool PACKED>
__global__ __launch_bounds__(256) void k_tonemap(TmArgs a, const void* __restrict__ src, const int32_t* __restrict__ owned_tiles,
                                                  const float* __restrict__ exposure, void* __restrict__ dst,
                                                  const float* __restrict__ thr_g, const uint8_t* __restrict__ enc_g)
{
    __shared__ float thr[kThrTabSize];
    __shared__ __attribute__((aligned(4))) uint8_t enc[(kEncTabSize + 3) / 4 * 4];
    thr[threadIdx.x] = thr_g[threadIdx.x];
    if (threadIdx.x == 0) thr[256] = __uint_as_float(0x7fc00000u);   // NaN: no x is >= it, not even +inf
    for (int i = threadIdx.x; i < (kEncTabSize + 3) / 4; i += 256) reinterpret_cast<uint32_t*>(enc)[i] = reinterpret_cast<const uint32_t*>(enc_g)[i];
    __syncthreads();
    const QuadPos q = quad_pos<PACKED>(a, owned_tiles, blockIdx.x, (int)threadIdx.x);
    if (q.valid == 0) return;
    float adapted = exposure[0];
    if (!(adapted > 0.0f)) adapted = a.min_adapted;
    adapted = 1.0f / adapted;                        // used as the reciprocal from here on
    float rgb[4][3];
    load_quad<PACKED>(src, q.index, rgb);
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = tm_pixel(a, adapted, rgb[k], thr, enc);
    if (PACKED) {
        // RGB8, 12 B per quad: r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
        uint32_t* d = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(dst) + q.index * 3);
        d[0] = o[0] | (o[1] << 24);
        d[1] = (o[1] >> 8) | (o[2] << 16);
        d[2] = (o[2] >> 16) | (o[3] << 8);
    } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(dst) + q.index) =
            make_uint4(o[0] | 0xff000000u, o[1] | 0xff000000u, o[2] | 0xff000000u, o[3] | 0xff000000u);
    }
}

__global__ __launch_bounds__(256) void k_tonemap_scalar(TmArgs a, const uint2* __restrict__ src, const float* __restrict__ exposure,
                                                         uint32_t* __restrict__ dst, const float* __restrict__ thr_g,
                                                         const uint8_t* __restrict__ enc_g)
{
    __shared__ float thr[kThrTabSize];
    __shared__ __attribute__((aligned(4))) uint8_t enc[(kEncTabSize + 3) / 4 * 4];
    thr[threadIdx.x] = thr_g[threadIdx.x];
    if (threadIdx.x == 0) thr[256] = __uint_as_float(0x7fc00000u);   // NaN: no x is >= it, not even +inf
    for (int i = threadIdx.x; i < (kEncTabSize + 3) / 4; i += 256) reinterpret_cast<uint32_t*>(enc)[i] = reinterpret_cast<const uint32_t*>(enc_g)[i];
    __syncthreads();
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)a.w * a.h) return;
    float adapted = exposure[0];
    if (!(adapted > 0.0f)) adapted = a.min_adapted;
    adapted = 1.0f / adapted;                        // used as the reciprocal from here on
    const uint2 v = src[p];
    const float c[3] = { vr_half_to_float(v.x & 0xffffu), vr_half_to_float(v.x >> 16), vr_half_to_float(v.y & 0xffffu) };
    dst[p] = tm_pixel(a, adapted, c, thr, enc) | 0xff000000u;
}

// gathered = world_size packed RGB8 buffers back to back; one lane expands 4 pixels (12 B -> 16 B)
__global__ __launch_bounds__(256) void k_detile_ldr(const uint8_t* __restrict__ gathered, uint4* __restrict__ frame, int w, int h,
                                                     int tiles_x, const int32_t* __restrict__ tile_slot)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;     // quad index
    const size_t p = i * 4;
    if (p >= (size_t)w * h) return;
    const int y = (int)(p / (size_t)w), x = (int)(p - (size_t)y * w);
    const int tx = x / VR_OWNER_TILE, ty = y / VR_OWNER_TILE;
    const int slot = tile_slot[ty * tiles_x + tx];
    const size_t src = ((size_t)slot * VR_OWNER_TILE + (y - ty * VR_OWNER_TILE)) * VR_OWNER_TILE + (x - tx * VR_OWNER_TILE);
    const uint32_t* g = reinterpret_cast<const uint32_t*>(gathered + src * 3);     // src is a multiple of 4 pixels: 4-byte aligned
    const uint32_t d0 = g[0], d1 = g[1], d2 = g[2];
    frame[i] = make_uint4((d0 & 0xffffffu) | 0xff000000u, (d0 >> 24) | ((d1 & 0xffffu) << 8) | 0xff000000u,
                          (d1 >> 16) | ((d2 & 0xffu) << 16) | 0xff000000u, (d2 >> 8) | 0xff000000u);
}

// ---- host -----------------------------------------------------------------------------------
extern "C" VR_API void vr_tonemap_default_params(vr_tonemap_params* p)
{
    if (!p) return;
    p->histogram_low_percentile = 0.8f; p->histogram_high_percentile = 0.95f;
    p->eye_adaptation_speed_up = 1.0f; p->eye_adaptation_speed_down = 0.5f;
    p->min_adapted_luminance = 0.02f; p->max_adapted_luminance = 0.5f;
    p->exposure_bias = -0.5f; p->white_point = 3.0f;
    p->min_log_luminance = -10.0f; p->max_log_luminance = 4.0f;
}

extern "C" VR_API int vr_tonemap_create(vr_context* ctx, vr_tonemap** out)
{
    VR_REQUIRE(ctx && out, "NULL argument");
    VR_HIP(hipSetDevice(ctx->device));
    vr_tonemap* tm = new vr_tonemap();
    tm->ctx = ctx; tm->d_hist = nullptr; tm->d_exposure = nullptr;
    if (hipMalloc(&tm->d_hist, sizeof(uint32_t) * VR_TONEMAP_BINS) != hipSuccess || hipMalloc(&tm->d_exposure, sizeof(float) * 4) != hipSuccess) {
        vr_tonemap_destroy(tm);
        vr_set_error("hipMalloc failed for the tone-mapping buffers");
        return VR_ERR_OUT_OF_MEMORY;
    }
    VR_HIP(hipMemsetAsync(tm->d_hist, 0, sizeof(uint32_t) * VR_TONEMAP_BINS, ctx->stream));
    VR_HIP(hipMemsetAsync(tm->d_exposure, 0, sizeof(float) * 4, ctx->stream));
    *out = tm;
    return VR_OK;
}

extern "C" VR_API void vr_tonemap_destroy(vr_tonemap* tm)
{
    if (!tm) return;
    (void)hipSetDevice(tm->ctx->device);
    (void)hipStreamSynchronize(tm->ctx->stream);
    (void)hipFree(tm->d_hist); (void)hipFree(tm->d_exposure);
    delete tm;
}

extern "C" VR_API int vr_tonemap_reset_exposure(vr_tonemap* tm, float adapted_luminance)
{
    VR_REQUIRE(tm, "NULL argument");
    VR_HIP(hipSetDevice(tm->ctx->device));
    uint32_t bits; memcpy(&bits, &adapted_luminance, 4);
    VR_HIP(hipMemsetD32Async((hipDeviceptr_t)tm->d_exposure, (int)bits, 1, tm->ctx->stream));
    return VR_OK;
}

extern "C" VR_API int vr_tonemap_reset_histogram(vr_tonemap* tm)
{
    VR_REQUIRE(tm, "NULL argument");
    VR_HIP(hipSetDevice(tm->ctx->device));
    VR_HIP(hipMemsetAsync(tm->d_hist, 0, sizeof(uint32_t) * VR_TONEMAP_BINS, tm->ctx->stream));
    return VR_OK;
}

static int check_params(const vr_tonemap_params* p)
{
    VR_REQUIRE(p, "NULL argument");
    VR_REQUIRE(p->max_log_luminance > p->min_log_luminance, "max_log_luminance must exceed min_log_luminance");
    VR_REQUIRE(p->white_point > 0.0f, "white_point must be positive");
    VR_REQUIRE(p->min_adapted_luminance > 0.0f && p->max_adapted_luminance >= p->min_adapted_luminance, "bad adapted-luminance range");
    VR_REQUIRE(p->histogram_low_percentile >= 0.0f && p->histogram_high_percentile <= 1.0f
               && p->histogram_high_percentile >= p->histogram_low_percentile, "bad histogram percentiles");
    return VR_OK;
}

static TmArgs make_args(const vr_tonemap_params* p, int w, int h)
{
    TmArgs a;
    a.w = w; a.h = h; a.tiles_x = (w + VR_OWNER_TILE - 1) / VR_OWNER_TILE;
    a.scale = 1.0f / (p->max_log_luminance - p->min_log_luminance);
    a.bias = (0.0f - p->min_log_luminance) * a.scale;
    a.exposure_scale = exp2f(p->exposure_bias);
    a.wp_inv2 = 1.0f / (p->white_point * p->white_point);
    a.min_adapted = p->min_adapted_luminance;
    return a;
}

// size checks of a (possibly packed) HDR source; returns the number of 256-lane blocks of the vector path
static int source_blocks(vr_context* ctx, vr_image* hdr, int w, int h, const vr_partition* part, size_t* blocks, const PartTables** tables)
{
    VR_REQUIRE(w > 0 && h > 0, "bad frame size");
    *tables = nullptr;
    if (part) {
        int rc = vr_partition_tables(ctx, w, h, part, tables);
        if (rc) return rc;
        VR_REQUIRE((size_t)(*tables)->max_owned * VR_OWNER_TILE * VR_OWNER_TILE * 6 <= hdr->capacity_bytes, "hdr is smaller than vr_partition_packed_bytes()");
        *blocks = (size_t)(*tables)->num_owned * 16;
    } else {
        VR_REQUIRE((size_t)w * h * 8 <= hdr->capacity_bytes, "hdr is smaller than the frame");
        *blocks = ((size_t)w * h / 4 + 255) / 256;
    }
    return VR_OK;
}

extern "C" VR_API int vr_tonemap_add_frame_to_histogram(vr_tonemap* tm, const vr_tonemap_params* p, vr_image* hdr, int32_t w, int32_t h,
                                                         const vr_partition* part)
{
    VR_REQUIRE(tm && hdr, "NULL argument");
    int rc = check_params(p); if (rc) return rc;
    vr_context* ctx = tm->ctx;
    VR_REQUIRE(hdr->ctx->device == ctx->device, "image and tone-mapping pass live on different devices");
    VR_HIP(hipSetDevice(ctx->device));
    size_t blocks = 0;
    const PartTables* pt = nullptr;
    if ((rc = source_blocks(ctx, hdr, w, h, part, &blocks, &pt))) return rc;
    const TmArgs a = make_args(p, w, h);
    VrKernelScope ks(ctx, VR_K_TM_HISTOGRAM);
    if (part) {
        if (blocks > 0)
            hipLaunchKernelGGL(k_tm_histogram<true>, dim3((unsigned)(blocks < (size_t)kHistBlocks ? blocks : (size_t)kHistBlocks)), dim3(256), 0, ctx->stream, a, (const void*)hdr->data,
                               pt->d_owned_tiles, blocks, tm->d_hist);
    } else if (w % 4 == 0) {
        hipLaunchKernelGGL(k_tm_histogram<false>, dim3((unsigned)(blocks < (size_t)kHistBlocks ? blocks : (size_t)kHistBlocks)), dim3(256), 0, ctx->stream, a, (const void*)hdr->data,
                           (const int32_t*)nullptr, blocks, tm->d_hist);
    } else {
        const size_t b = ((size_t)w * h + 255) / 256;
        hipLaunchKernelGGL(k_tm_histogram_scalar, dim3((unsigned)(b < (size_t)kHistBlocks ? b : (size_t)kHistBlocks)), dim3(256), 0, ctx->stream, a, (const uint2*)hdr->data, tm->d_hist);
    }
    VR_HIP(hipGetLastError());
    return VR_OK;
}

extern "C" VR_API void* vr_tonemap_histogram_device_ptr(vr_tonemap* tm) { return tm ? tm->d_hist : nullptr; }

extern "C" VR_API int vr_tonemap_compute_exposure(vr_tonemap* tm, const vr_tonemap_params* p, float frame_time)
{
    VR_REQUIRE(tm, "NULL argument");
    int rc = check_params(p); if (rc) return rc;
    vr_context* ctx = tm->ctx;
    VR_HIP(hipSetDevice(ctx->device));
    const TmArgs a = make_args(p, 4, 4);
    const float k_up = (float)(1.0 - exp(-(double)frame_time * (double)p->eye_adaptation_speed_up));
    const float k_down = (float)(1.0 - exp(-(double)frame_time * (double)p->eye_adaptation_speed_down));
    VrKernelScope ks(ctx, VR_K_TM_EXPOSURE);
    hipLaunchKernelGGL(k_tm_exposure, dim3(1), dim3(256), 0, ctx->stream, (const uint32_t*)tm->d_hist, tm->d_exposure, a.scale, a.bias,
                       p->histogram_low_percentile, p->histogram_high_percentile, p->min_log_luminance, p->min_adapted_luminance,
                       p->max_adapted_luminance, k_up, k_down, p->eye_adaptation_speed_up > 0.0f ? 1 : 0, p->eye_adaptation_speed_down > 0.0f ? 1 : 0);
    VR_HIP(hipGetLastError());
    return VR_OK;
}

extern "C" VR_API size_t vr_partition_packed_bytes_ldr(int32_t w, int32_t h, int32_t world)
{
    return vr_partition_packed_bytes(w, h, world) / 2;         // RGB8 instead of RGB16F
}

extern "C" VR_API int vr_tonemap_render(vr_tonemap* tm, const vr_tonemap_params* p, vr_image* hdr, int32_t w, int32_t h, void* ldr,
                                         size_t ldr_capacity, const vr_partition* part)
{
    VR_REQUIRE(tm && hdr && ldr, "NULL argument");
    int rc = check_params(p); if (rc) return rc;
    vr_context* ctx = tm->ctx;
    VR_REQUIRE(hdr->ctx->device == ctx->device, "image and tone-mapping pass live on different devices");
    VR_HIP(hipSetDevice(ctx->device));
    size_t blocks = 0;
    const PartTables* pt = nullptr;
    if ((rc = source_blocks(ctx, hdr, w, h, part, &blocks, &pt))) return rc;
    const TmArgs a = make_args(p, w, h);
    VrKernelScope ks(ctx, VR_K_TONEMAP);
    if (part) {
        VR_REQUIRE((size_t)pt->max_owned * VR_OWNER_TILE * VR_OWNER_TILE * 3 <= ldr_capacity, "ldr buffer is smaller than vr_partition_packed_bytes_ldr()");
        if (blocks > 0)
            hipLaunchKernelGGL(k_tonemap<true>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, (const void*)hdr->data, pt->d_owned_tiles,
                               (const float*)tm->d_exposure, ldr, ctx->d_srgb_thr, ctx->d_enc_tab);
    } else {
        VR_REQUIRE((size_t)w * h * 4 <= ldr_capacity, "ldr buffer is smaller than the frame");
        if (w % 4 == 0) {
            hipLaunchKernelGGL(k_tonemap<false>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, (const void*)hdr->data, (const int32_t*)nullptr,
                               (const float*)tm->d_exposure, ldr, ctx->d_srgb_thr, ctx->d_enc_tab);
        } else {
            const size_t b = ((size_t)w * h + 255) / 256;
            hipLaunchKernelGGL(k_tonemap_scalar, dim3((unsigned)b), dim3(256), 0, ctx->stream, a, (const uint2*)hdr->data, (const float*)tm->d_exposure,
                               (uint32_t*)ldr, ctx->d_srgb_thr, ctx->d_enc_tab);
        }
    }
    VR_HIP(hipGetLastError());
    return VR_OK;
}

extern "C" VR_API int vr_tonemap_simple_render(vr_tonemap* tm, const vr_tonemap_params* p, float frame_time, vr_image* hdr, void* ldr,
                                                size_t ldr_capacity)
{
    VR_REQUIRE(tm && hdr, "NULL argument");
    int rc;
    if ((rc = vr_tonemap_reset_histogram(tm))) return rc;
    if ((rc = vr_tonemap_add_frame_to_histogram(tm, p, hdr, hdr->w, hdr->h, nullptr))) return rc;
    if ((rc = vr_tonemap_compute_exposure(tm, p, frame_time))) return rc;
    return vr_tonemap_render(tm, p, hdr, hdr->w, hdr->h, ldr, ldr_capacity, nullptr);
}

extern "C" VR_API int vr_tonemap_download(vr_tonemap* tm, uint32_t histogram[VR_TONEMAP_BINS], float* adapted)
{
    VR_REQUIRE(tm, "NULL argument");
    VR_HIP(hipSetDevice(tm->ctx->device));
    VR_HIP(hipStreamSynchronize(tm->ctx->stream));
    if (histogram) VR_HIP(hipMemcpy(histogram, tm->d_hist, sizeof(uint32_t) * VR_TONEMAP_BINS, hipMemcpyDeviceToHost));
    if (adapted) VR_HIP(hipMemcpy(adapted, tm->d_exposure, sizeof(float), hipMemcpyDeviceToHost));
    return VR_OK;
}

extern "C" VR_API int vr_frame_detile_ldr(vr_context* ctx, const void* gathered, int32_t world, int32_t w, int32_t h, void* frame)
{
    VR_REQUIRE(ctx && gathered && frame, "NULL argument");
    VR_REQUIRE(w % 4 == 0 && w > 0 && h > 0 && world >= 1, "frame width must be a multiple of 4");
    VR_HIP(hipSetDevice(ctx->device));
    const PartTables* pt = nullptr;
    { int rc = vr_partition_slot_tables(ctx, w, h, world, &pt); if (rc) return rc; }
    const size_t quads = (size_t)w * h / 4;
    VrKernelScope ks(ctx, VR_K_DETILE_LDR);
    hipLaunchKernelGGL(k_detile_ldr, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t*)gathered,
                       (uint4*)frame, w, h, (w + VR_OWNER_TILE - 1) / VR_OWNER_TILE, pt->d_tile_slot);
    VR_HIP(hipGetLastError());
    return VR_OK;
}
