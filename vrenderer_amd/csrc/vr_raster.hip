// TerrainPass::Render as compute: vertex transform (terrain_vs.hlsl), triangle
// setup + screen-tile binning, near/guard-band clipper, and a tile rasteriser with
// the pixel shader (terrain_ps.hlsl) fused into its resolve phase.
//
// Design (sort-middle, visibility-buffer in LDS):
//   k_vertex : one lane per (instance, grid vertex); heightmap taps are byte loads
//              served by L2/Infinity Cache; output 48 B/vertex (clip + snapped).
//   k_setup  : one lane per triangle; cull (frustum, back-face, no-sample) and count
//              the 64x64 raster tiles it touches (global atomics, ~1 per triangle).
//              Triangles that cross the near plane /
//              leave the guard band go to a small list for the clipper, which emits explicit
//              sub-triangles (k_clip).
//   k_scan   : a slice of the entry array per tile (workgroup-local scans + one atomic each)
//              and the tile pass's launch order.
//   k_fill   : second pass of the bin fill.  Order inside a bin is
//              arbitrary — depth ties are resolved by a draw-order key, not by arrival order.
//   k_raster : one workgroup per raster tile.  A 64x64 x u64 visibility buffer
//              (32 KiB LDS) holds (depth bits << 32 | ~draw order) per pixel; LessOrEqual
//              in-order depth testing == 64-bit ds_min.  Small triangles are rasterised by
//              their own lane; big ones are broadcast with v_readlane and swept by all
//              64 lanes of the wave in 8x8 pixel blocks.  The resolve phase shades each
//              pixel's winner once (no overdraw shading) and leaves the tile with
//              16-byte coalesced stores per plane (28 B/pixel).
// Edge functions are exact (24.8 fixed point, 64-bit integers) with the D3D top-left
// rule; all float math is in fixed order with contraction off, so the G-buffer is
// bit-identical to the CPU oracle.
#include "vr_internal.h"
#include "vr_tex_dev.h"
#include "vr_deferred_dev.h"
#include "vr_experiments.h"
#include <type_traits>

#include <stdlib.h>
#include <string.h>

struct VertexArgs {
    float w2v[16], v2c[16];
    float cam_x, cam_z;
    float lod_ranges[VR_MAX_LODS];
    float morph_start, world_size, max_height;
    float vp_x, vp_y, vp_w, vp_h;
};

struct RasterArgs {
    int w, h;                       // render target size
    int vx0, vy0, vx1, vy1;         // inclusive pixel bounds (viewport ∩ target)
    int rtx, rty;                   // raster tiles per row / column
    int tile_shift;                 // log2 of the raster tile edge (5 or 6)
    int mirrored;
    int world, rank;                // partition (world <= 1: whole frame)
    uint32_t world_magic;           // ceil(2^16 / world): (v * magic) >> 16 == v / world for the owner-tile sums v < 1024 (world <= 64)
    int depth_only, assume_cleared;
    int wireframe;                  // RasterFillMode::Wireframe: triangle edges as aliased lines
    float world_size, inv_world_size;
    int ws_pow2;                    // world_size is a power of two: x / ws == x * (1 / ws) exactly
    // fast variant (power-of-two world size): what the implicit LOD multiplies the world-space derivatives by - texels per world
    // unit, (float)w0 * inv_world_size: (d * inv_ws) * w0 == d * (inv_ws * w0) bit for bit, the scaling by a power of two commutes
    // with the rounding (where d * inv_ws would be a denormal both forms give a LOD far below 0, clamped to 0) - and the last mip
    // level as a float (the tile pass otherwise converts three integers per pixel: the compiler rematerialises them)
    float lod_w, lod_h, max_level_f;
    uint32_t bin_capacity;
    uint32_t extra_vert_base, extra_vert_cap, hard_cap;
    float vp_x, vp_y, vp_w, vp_h;
};

// counters[]: 0 selected nodes, 1 status flags, 2 hard sub-triangles, 3 extra verts,
//             4 hard-list length, 5 total bin entries, 8..15 raster tiles per bin-length class (k_scan)
// (2..15 are reset by k_vertex, the first kernel of a chain that uses them)
enum { C_COUNT = 0, C_FLAGS = 1, C_HARDTRIS = 2, C_XVERTS = 3, C_HARDLIST = 4, C_BINTOTAL = 5, C_CLASS0 = 8 };


// One bilinear tap through the quad (footprint) table, split into address and filter so that the
// loads of all taps of a pixel can be issued back to back before the first one is consumed.
struct QuadTap { uint32_t idx; float fx, fy; };
__device__ __forceinline__ QuadTap quad_tap(int w, int h, float u, float v)
{
    const float x = __builtin_fmaf(u, (float)w, -0.5f), y = __builtin_fmaf(v, (float)h, -0.5f);
    float xf = floorf(x), yf = floorf(y);
    QuadTap q; q.fx = x - xf; q.fy = y - yf;
    xf = vr_min(vr_max(xf, -1.0f), (float)w); yf = vr_min(vr_max(yf, -1.0f), (float)h);
    q.idx = (uint32_t)(__mul24((int)yf + 1, w + 2) + ((int)xf + 1));
    return q;
}
__device__ __forceinline__ float quad_filter(uint32_t e, const QuadTap& q, const float* __restrict__ r8)
{
    const float t00 = r8[e & 255u], t10 = r8[(e >> 8) & 255u], t01 = r8[(e >> 16) & 255u], t11 = r8[e >> 24];
    const float top = __builtin_fmaf(t10 - t00, q.fx, t00), bot = __builtin_fmaf(t11 - t01, q.fx, t01);   // the sampler's lerps are fused (oracle: tex_bilinear)
    return __builtin_fmaf(bot - top, q.fy, top);
}
// ---------------------------------------------------------------------------------------
// vertex stage (terrain_vs.hlsl:35-62)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void snap_vertex(DevVert& v, float vp_x, float vp_y, float vp_w, float vp_h)
{
    float iw = 1.0f / v.cw;
    float nx = v.cx * iw, ny = v.cy * iw;
    float sx = (nx * 0.5f + 0.5f) * vp_w + vp_x;
    float sy = (ny * -0.5f + 0.5f) * vp_h + vp_y;
    float fx = floorf(sx * 256.0f + 0.5f), fy = floorf(sy * 256.0f + 0.5f);
    // vertices outside the guard band are never rasterised from these values (k_setup
    // sends their triangles to the clipper); clamp so the conversion is defined
    fx = vr_min(vr_max(fx, -1.0e9f), 1.0e9f); fy = vr_min(vr_max(fy, -1.0e9f), 1.0e9f);
    v.X = (int32_t)fx; v.Y = (int32_t)fy;
    v.z = v.cz * iw; v.iw = iw;
}

__global__ __launch_bounds__(256) void k_vertex(VertexArgs a, DevTex hm, const vr_instance* __restrict__ inst,
                                                 uint32_t* __restrict__ counters, DevVert* __restrict__ verts)
{
    VR_GEOMETRY_PRIORITY();
    // first kernel of every frame: reset the frame's work counters (k_setup is the first to use them); words 6 and 7 are
    // k_select's (the selection's size before truncation, NodeSelect's count) and stay
    if (blockIdx.x == 0 && threadIdx.x < 14 && (threadIdx.x < 4 || threadIdx.x >= 6)) counters[2 + threadIdx.x] = 0u;
    __shared__ float r8[256];
    __shared__ uint32_t s_qoff[kMaxLevels];
    r8[threadIdx.x] = (float)threadIdx.x / 255.0f;     // UNORM8 -> float, correctly rounded
    if (threadIdx.x < kMaxLevels) s_qoff[threadIdx.x] = hm.qoff[threadIdx.x];
    __syncthreads();
    const uint32_t total = counters[C_COUNT] * (uint32_t)kVertsPerInst;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gridDim.x * blockDim.x) {
        const uint32_t i = v / kVertsPerInst, r = v - i * kVertsPerInst;
        const int vz = (int)(r / kSide), vx = (int)(r - (uint32_t)vz * kSide);
        const float* M = inst[i].transform;
        const float half = (float)(kGrid / 2), gs = (float)kGrid;
        const float p0 = (float)(vx - kGrid / 2) / half, p1 = 0.0f, p2 = (float)(vz - kGrid / 2) / half;   // TerrainPass.cpp:63
        float world[4];
#pragma unroll
        for (int k = 0; k < 3; k++) world[k] = ((M[k * 4 + 0] * p0 + M[k * 4 + 1] * p1) + M[k * 4 + 2] * p2) + M[k * 4 + 3] * 1.0f;   // :44
        world[3] = 1.0f;
        const float dx = world[0] - a.cam_x, dz = world[2] - a.cam_z;
        const float distance = sqrtf(dx * dx + dz * dz);                                            // :46
        const float gridExtents = 2.0f * sqrtf((M[0] * M[0] + M[4] * M[4]) + M[8] * M[8]);          // :47
        // computeMorphK (:16-25): int(log2(x)) == exponent for x >= 1, clamps to 0 below
        int lod = gridExtents > 0.0f ? (int)((__float_as_uint(gridExtents) >> 23) & 255u) - 127 : 0;
        lod = lod < 0 ? 0 : (lod > 11 ? 11 : lod);
        const float start = a.lod_ranges[lod] * a.morph_start, end = a.lod_ranges[lod];
        const float delta = end - start;
        const float morphK = vr_saturate((distance - start) / delta);
        // morphVertex (:10-14)
        const float gp0 = (p0 + 1.0f) * 0.5f, gp1 = (p2 + 1.0f) * 0.5f;                               // :49
        {
            float q = gp0 * gs * 0.5f; float fr = (q - floorf(q)) * 2.0f / gs;
            world[0] = world[0] - fr * gridExtents * morphK;
            q = gp1 * gs * 0.5f; fr = (q - floorf(q)) * 2.0f / gs;
            world[2] = world[2] - fr * gridExtents * morphK;
        }
        // sampleHeight (:27-33)
        const float halfSize = a.world_size * 0.5f;
        const float u = (world[0] + halfSize) / a.world_size, w_ = (world[2] + halfSize) / a.world_size;
        {   // SampleLevel(linearClamp, uv, 0.1): levels 0 and 1 fetched together
            const LodSplit ls = vr_lod_split(hm.levels, 0.1f);
            const int l1 = min(ls.l0 + 1, hm.levels - 1);
            const int w0 = max(1, hm.w0 >> ls.l0), h0 = max(1, hm.h0 >> ls.l0), w1 = max(1, hm.w0 >> l1), h1 = max(1, hm.h0 >> l1);
            const QuadTap t0 = quad_tap(w0, h0, u, w_), t1 = quad_tap(w1, h1, u, w_);
            const uint32_t e0 = hm.quad[s_qoff[ls.l0] + t0.idx], e1 = hm.quad[s_qoff[l1] + t1.idx];
            const float s0 = quad_filter(e0, t0, r8), s1 = quad_filter(e1, t1, r8);
            const float hv = ls.f > 0.0f ? __builtin_fmaf(s1 - s0, ls.f, s0) : s0;
            world[1] = hv * a.max_height;                                                           // :51
        }
        float viewPos[4], clip[4];
#pragma unroll
        for (int j = 0; j < 4; j++) viewPos[j] = ((world[0] * a.w2v[0 * 4 + j] + world[1] * a.w2v[1 * 4 + j]) + world[2] * a.w2v[2 * 4 + j]) + world[3] * a.w2v[3 * 4 + j];   // :60
#pragma unroll
        for (int j = 0; j < 4; j++) clip[j] = ((viewPos[0] * a.v2c[0 * 4 + j] + viewPos[1] * a.v2c[1 * 4 + j]) + viewPos[2] * a.v2c[2 * 4 + j]) + viewPos[3] * a.v2c[3 * 4 + j];   // :61
        DevVert o;
        o.cx = clip[0]; o.cy = clip[1]; o.cz = clip[2]; o.cw = clip[3];
        o.wx = world[0]; o.wz = world[2];
        o.pad0 = 0; o.pad1 = 0;
        if (o.cw > 0.0f) snap_vertex(o, a.vp_x, a.vp_y, a.vp_w, a.vp_h);
        else { o.X = 0; o.Y = 0; o.z = 0.0f; o.iw = 0.0f; }
        verts[v] = o;
    }
}

// ---------------------------------------------------------------------------------------
// triangle setup shared by binning and rasterisation
// ---------------------------------------------------------------------------------------
struct ScreenVert { int32_t X, Y; float z, iw; float wx, wz; };

__device__ __forceinline__ ScreenVert load_sv(const DevVert* __restrict__ verts, uint32_t idx)
{
    const float4 b = *reinterpret_cast<const float4*>(&verts[idx].wx);   // wx, wz, X, Y
    const float2 c = *reinterpret_cast<const float2*>(&verts[idx].z);    // z, iw
    ScreenVert s;
    s.wx = b.x; s.wz = b.y; s.X = __float_as_int(b.z); s.Y = __float_as_int(b.w); s.z = c.x; s.iw = c.y;
    return s;
}

__device__ __forceinline__ void regular_tri_indices(uint32_t tri, uint32_t& i0, uint32_t& i1, uint32_t& i2)
{
    // index buffer (TerrainPass.cpp:68-87): per cell (BL,TL,TR), (BL,TR,BR); row = z, column = x
    const uint32_t inst = tri >> 11, rem = tri & 2047u, cell = rem >> 1, second = rem & 1u;
    const uint32_t ci = cell >> 5, cj = cell & 31u;
    const uint32_t bl = inst * kVertsPerInst + ci * kSide + cj, tl = bl + kSide, tr = tl + 1, br = bl + 1;
    i0 = bl; i1 = second ? tr : tl; i2 = second ? br : tr;
}

struct TriSetup {
    int32_t A0, B0, A1, B1, A2, B2;   // E_i(PX,PY) = A_i*PX + B_i*PY + C_i
    int64_t C0, C1, C2;
    int32_t bias0, bias1, bias2;      // 0 for top-left edges, else 1
    int64_t area2;                    // twice the area, 24.8 x 24.8 units, > 0 (clockwise)
    int x0, y0, x1, y1;               // inclusive pixel bounds after clamping; (x0, y0) is the anchor of the planes
    bool visible;
};

__device__ __forceinline__ bool is_top_left(int32_t dx, int32_t dy) { return (dy < 0) || (dy == 0 && dx > 0); }

// Orders the vertices clockwise on the (y-down) target, applies the cull state of
// TerrainPass::CreateGraphicsPipeline (TerrainPass.cpp:474-476), and returns edge
// functions + the pixel bounding box clamped to [bx0..bx1] x [by0..by1].
__device__ __forceinline__ TriSetup tri_setup(ScreenVert& s0, ScreenVert& s1, ScreenVert& s2, int mirrored,
                                              int bx0, int by0, int bx1, int by1, bool wire = false)
{
    TriSetup t;
    t.visible = false;
    int64_t area2 = (int64_t)(s1.X - s0.X) * (int64_t)(s2.Y - s0.Y) - (int64_t)(s2.X - s0.X) * (int64_t)(s1.Y - s0.Y);
    if (area2 == 0) return t;
    const bool cw = area2 > 0;
    const bool front = mirrored ? !cw : cw;
    if (!front) return t;
    if (!cw) { ScreenVert tmp = s1; s1 = s2; s2 = tmp; area2 = -area2; }
    const int32_t minX = min(s0.X, min(s1.X, s2.X)), maxX = max(s0.X, max(s1.X, s2.X));
    const int32_t minY = min(s0.Y, min(s1.Y, s2.Y)), maxY = max(s0.Y, max(s1.Y, s2.Y));
    if (wire) {     // a line pixel is the one that contains the line point, not one whose centre is inside the box
        t.x0 = max(minX >> 8, bx0); t.x1 = min(maxX >> 8, bx1);
        t.y0 = max(minY >> 8, by0); t.y1 = min(maxY >> 8, by1);
    } else {
        t.x0 = max((minX - 128 + 255) >> 8, bx0); t.x1 = min((maxX - 128) >> 8, bx1);
        t.y0 = max((minY - 128 + 255) >> 8, by0); t.y1 = min((maxY - 128) >> 8, by1);
    }
    if (t.x0 > t.x1 || t.y0 > t.y1) return t;
    // edge(a,b)(p) = (bX-aX)*(py-aY) - (bY-aY)*(px-aX)
    t.B0 = s2.X - s1.X; t.A0 = -(s2.Y - s1.Y); t.C0 = -(int64_t)t.B0 * s1.Y - (int64_t)t.A0 * s1.X;
    t.B1 = s0.X - s2.X; t.A1 = -(s0.Y - s2.Y); t.C1 = -(int64_t)t.B1 * s2.Y - (int64_t)t.A1 * s2.X;
    t.B2 = s1.X - s0.X; t.A2 = -(s1.Y - s0.Y); t.C2 = -(int64_t)t.B2 * s0.Y - (int64_t)t.A2 * s0.X;
    t.bias0 = is_top_left(s2.X - s1.X, s2.Y - s1.Y) ? 0 : 1;
    t.bias1 = is_top_left(s0.X - s2.X, s0.Y - s2.Y) ? 0 : 1;
    t.bias2 = is_top_left(s1.X - s0.X, s1.Y - s0.Y) ? 0 : 1;
    t.area2 = area2;
    t.visible = true;
    return t;
}

__device__ __forceinline__ bool tile_owned(const RasterArgs& a, int tx, int ty)
{
    if (a.world <= 1) return true;
    const int sub_shift = 7 - a.tile_shift;                     // VR_OWNER_TILE = 128 = raster tile << sub_shift
    // (tx + ty) % world without the ~40-instruction division by a run-time value: on 32-pixel tiles most triangles of a large
    // frame touch several tiles and this test runs per tile in three kernels
    const uint32_t v = (uint32_t)((tx >> sub_shift) + (ty >> sub_shift));
    return v - ((v * a.world_magic) >> 16) * (uint32_t)a.world == (uint32_t)a.rank;
}

__device__ __forceinline__ uint64_t pack_rect(int tx0, int ty0, int tx1, int ty1)
{
    return (uint64_t)(uint32_t)tx0 | ((uint64_t)(uint32_t)ty0 << 16) | ((uint64_t)(uint32_t)tx1 << 32) | ((uint64_t)(uint32_t)ty1 << 48);
}

// Fixed-function interpolation as per-triangle plane equations (oracle: planes_setup, raster model revision 3): depth z,
// q = 1/w and the attributes over w are affine in the pixel position, P(x, y) = p0 + px (x - ax) + py (y - ay) about the
// anchor pixel (ax, ay) = the first pixel of the triangle's viewport-clamped box.  The coefficients are set up once per
// triangle in double precision from the exact integer edge functions - the same expressions, in the same order, as the
// oracle - and rounded to float; a pixel evaluates fma(py, dy, fma(px, dx, p0)).
struct Plane { float p0, px, py; };
__device__ __forceinline__ Plane plane_of(const double l[3], const double lx[3], const double ly[3], double a0, double a1, double a2)
{
    Plane p;
    p.p0 = (float)((l[0] * a0 + l[1] * a1) + l[2] * a2);
    p.px = (float)((lx[0] * a0 + lx[1] * a1) + lx[2] * a2);
    p.py = (float)((ly[0] * a0 + ly[1] * a1) + ly[2] * a2);
    return p;
}
__device__ __forceinline__ float plane_at(float p0, float px, float py, float dx, float dy) { return __builtin_fmaf(py, dy, __builtin_fmaf(px, dx, p0)); }

// Triangle record: everything the tile pass needs of a triangle that survived culling, written once by
// k_setup (regular triangles, index = triangle id) or k_clip (clipper output, index = kRecHardBase-relative)
// and read by both phases of k_raster, so neither re-derives the set-up from the vertices.  Eight 16-byte
// groups (128 B):
//   0: A0 B0 C0(lo hi)        edge 0            (coverage)
//   1: A1 B1 C1(lo hi)        edge 1            (coverage)
//   2: A2 B2 C2(lo hi)        edge 2            (coverage)
//   3: Z0 Zx Zy flags         depth plane       (coverage); flags: bias0..2, bit 3 = small
//   4: box0 box1 - -          pixel box (x | y << 16, inclusive, clamped to the viewport); box0 is the planes' anchor
//   5: Q0 Qx Qy box0          1/w plane + anchor        (resolve)
//   6: NX0 NXx NXy -          (world x) / w plane       (resolve)
//   7: NZ0 NZx NZy -          (world z) / w plane       (resolve)
// small = every |A_i|, |B_i| <= 2^14 (edges up to 64 pixels): every edge value at a pixel of a tile the triangle
// touches then fits 31 bits (|E(P)| <= 2^29 inside its box, + 64 pixels * 256 * (|A|+|B|) <= 2^29 to any pixel of
// the tile), and 256*B fits 24 bits, so the tile pass runs such triangles in int32 / mad24.
constexpr int32_t kSmallEdge = 1 << 14;

__device__ __forceinline__ void write_tri_rec(uint4* __restrict__ dst, const TriSetup& t, const ScreenVert& s0, const ScreenVert& s1,
                                              const ScreenVert& s2)
{
    // barycentrics of the anchor's centre and their steps per pixel: l1 = E1 / area2, l2 = E2 / area2 (E1 = edge(s2, s0), E2 = edge(s0, s1))
    const double inv = 1.0 / (double)t.area2;
    const int64_t PX = (int64_t)t.x0 * 256 + 128, PY = (int64_t)t.y0 * 256 + 128;
    const int64_t E1 = (int64_t)t.A1 * PX + ((int64_t)t.B1 * PY + t.C1), E2 = (int64_t)t.A2 * PX + ((int64_t)t.B2 * PY + t.C2);
    double l[3], lx[3], ly[3];
    l[1] = (double)E1 * inv; l[2] = (double)E2 * inv; l[0] = (1.0 - l[1]) - l[2];
    lx[1] = (double)((int64_t)t.A1 * 256) * inv; lx[2] = (double)((int64_t)t.A2 * 256) * inv; lx[0] = (0.0 - lx[1]) - lx[2];
    ly[1] = (double)((int64_t)t.B1 * 256) * inv; ly[2] = (double)((int64_t)t.B2 * 256) * inv; ly[0] = (0.0 - ly[1]) - ly[2];
    const Plane z = plane_of(l, lx, ly, (double)s0.z, (double)s1.z, (double)s2.z);
    const Plane q = plane_of(l, lx, ly, (double)s0.iw, (double)s1.iw, (double)s2.iw);
    const Plane nx = plane_of(l, lx, ly, (double)s0.wx * (double)s0.iw, (double)s1.wx * (double)s1.iw, (double)s2.wx * (double)s2.iw);
    const Plane nz = plane_of(l, lx, ly, (double)s0.wz * (double)s0.iw, (double)s1.wz * (double)s1.iw, (double)s2.wz * (double)s2.iw);
    const int32_t m = max(max(max(abs(t.A0), abs(t.B0)), max(abs(t.A1), abs(t.B1))), max(abs(t.A2), abs(t.B2)));
    const uint32_t flags = (uint32_t)t.bias0 | ((uint32_t)t.bias1 << 1) | ((uint32_t)t.bias2 << 2) | (m <= kSmallEdge ? 8u : 0u);
    const uint32_t box0 = (uint32_t)t.x0 | ((uint32_t)t.y0 << 16), box1 = (uint32_t)t.x1 | ((uint32_t)t.y1 << 16);
#define F2U(x) __float_as_uint(x)
    dst[0] = make_uint4((uint32_t)t.A0, (uint32_t)t.B0, (uint32_t)(uint64_t)t.C0, (uint32_t)((uint64_t)t.C0 >> 32));
    dst[1] = make_uint4((uint32_t)t.A1, (uint32_t)t.B1, (uint32_t)(uint64_t)t.C1, (uint32_t)((uint64_t)t.C1 >> 32));
    dst[2] = make_uint4((uint32_t)t.A2, (uint32_t)t.B2, (uint32_t)(uint64_t)t.C2, (uint32_t)((uint64_t)t.C2 >> 32));
    dst[3] = make_uint4(F2U(z.p0), F2U(z.px), F2U(z.py), flags);
    dst[4] = make_uint4(box0, box1, 0u, 0u);
    dst[5] = make_uint4(F2U(q.p0), F2U(q.px), F2U(q.py), box0);
    dst[6] = make_uint4(F2U(nx.p0), F2U(nx.px), F2U(nx.py), 0u);
    dst[7] = make_uint4(F2U(nz.p0), F2U(nz.px), F2U(nz.py), 0u);
#undef F2U
}

// Sets a triangle up against the viewport: its raster-tile rectangle, or ~0 when it is culled; a surviving
// triangle leaves its record at `rec`.
__device__ __forceinline__ uint64_t triangle_rect(const RasterArgs& a, ScreenVert s0, ScreenVert s1, ScreenVert s2, uint4* __restrict__ rec)
{
    TriSetup t = tri_setup(s0, s1, s2, a.mirrored, a.vx0, a.vy0, a.vx1, a.vy1, a.wireframe != 0);
    if (!t.visible) return ~0ull;
    const int tx0 = t.x0 >> a.tile_shift, ty0 = t.y0 >> a.tile_shift, tx1 = t.x1 >> a.tile_shift, ty1 = t.y1 >> a.tile_shift;
    if (a.world > 1) {
        // A rank of an N-way split draws 1/N of the tiles but sets up all the geometry: a triangle that touches none of this
        // rank's tiles needs neither a record (its plane set-up in double precision is most of this kernel) nor a bin.
        bool mine = false;
        for (int ty = ty0; ty <= ty1 && !mine; ty++)
            for (int tx = tx0; tx <= tx1; tx++) if (tile_owned(a, tx, ty)) { mine = true; break; }
        if (!mine) return ~0ull;
    }
    write_tri_rec(rec, t, s0, s1, s2);          // tri_setup has put the vertices into clockwise order
    return pack_rect(tx0, ty0, tx1, ty1);
}

// The coverage part of a triangle record (groups 0-4) as k_fill holds it while it writes the triangle's bin entries.
struct TriCov {
    int32_t A0, B0, A1, B1, A2, B2;
    int64_t C0, C1, C2;
    float z0, zx, zy;
    uint32_t flags;                    // bias0..2, bit 3 = small
    int bx0, by0, bx1, by1;            // pixel box (viewport-clamped); (bx0, by0) = the planes' anchor
};
__device__ __forceinline__ int64_t edge_eval(int32_t A, int32_t B, int64_t C, int32_t PX, int32_t PY);
__device__ __forceinline__ int64_t rec_c(const uint4& g);
__device__ __forceinline__ uint32_t order_of(uint32_t key);
__device__ __forceinline__ TriCov load_tri_cov(const uint4* __restrict__ rp)
{
    const uint4 g0 = rp[0], g1 = rp[1], g2 = rp[2], g3 = rp[3], g4 = rp[4];
    TriCov c;
    c.A0 = (int32_t)g0.x; c.B0 = (int32_t)g0.y; c.A1 = (int32_t)g1.x; c.B1 = (int32_t)g1.y; c.A2 = (int32_t)g2.x; c.B2 = (int32_t)g2.y;
    c.C0 = rec_c(g0); c.C1 = rec_c(g1); c.C2 = rec_c(g2);
    c.z0 = __uint_as_float(g3.x); c.zx = __uint_as_float(g3.y); c.zy = __uint_as_float(g3.z); c.flags = g3.w;
    c.bx0 = (int)(g4.x & 0xffffu); c.by0 = (int)(g4.x >> 16); c.bx1 = (int)(g4.y & 0xffffu); c.by1 = (int)(g4.y >> 16);
    return c;
}
// The triangle as raster tile (tx, ty) sees it (TileEntry, vr_internal.h): what every tile-pass workgroup used to derive per lane
// from the record - the same integer expressions, so coverage is bit for bit what it was.
__device__ __forceinline__ void write_tile_entry(const RasterArgs& a, TileEntry* __restrict__ dst, const TriCov& c, uint32_t key, int tx, int ty)
{
    const int tile = 1 << a.tile_shift;
    const int ox = tx << a.tile_shift, oy = ty << a.tile_shift;
    const int32_t PX0 = ox * 256 + 128, PY0 = oy * 256 + 128;                       // centre of the tile's pixel (0, 0)
    const int64_t e0 = edge_eval(c.A0, c.B0, c.C0, PX0, PY0), e1 = edge_eval(c.A1, c.B1, c.C1, PX0, PY0), e2 = edge_eval(c.A2, c.B2, c.C2, PX0, PY0);
    const int lx0 = max(ox, a.vx0), ly0 = max(oy, a.vy0), lx1 = min(ox + tile - 1, a.vx1), ly1 = min(oy + tile - 1, a.vy1);
    const int x0 = max(c.bx0, lx0) - ox, y0 = max(c.by0, ly0) - oy, x1 = min(c.bx1, lx1) - ox, y1 = min(c.by1, ly1) - oy;
    const bool valid = x0 <= x1 && y0 <= y1;            // (never empty for a binned triangle; checked all the same)
    bool fits32 = (c.flags & 8u) != 0u;                  // small: every edge value over the tile fits 31 bits (write_tri_rec)
    if (!fits32) {
        const int64_t lim = (int64_t)1 << 30;
        const int64_t w0 = (int64_t)(kRasterTile + 8) * 256 * (llabs((int64_t)c.A0) + llabs((int64_t)c.B0));
        const int64_t w1 = (int64_t)(kRasterTile + 8) * 256 * (llabs((int64_t)c.A1) + llabs((int64_t)c.B1));
        const int64_t w2 = (int64_t)(kRasterTile + 8) * 256 * (llabs((int64_t)c.A2) + llabs((int64_t)c.B2));
        fits32 = llabs(e0) + w0 < lim && llabs(e1) + w1 < lim && llabs(e2) + w2 < lim;
    }
#define MUL256(v) ((uint32_t)(v) * 256u)      // (unsigned: the products of a triangle that does not fit are never used, but must not be undefined)
    uint4* __restrict__ q = reinterpret_cast<uint4*>(dst);
    q[0] = make_uint4((uint32_t)e0, (uint32_t)e1, (uint32_t)e2, MUL256(c.A0));
    q[1] = make_uint4(MUL256(c.B0), MUL256(c.A1), MUL256(c.B1), MUL256(c.A2));
    q[2] = make_uint4(MUL256(c.B2), valid ? ((uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)x1 << 16) | ((uint32_t)y1 << 24)) : 0u,
                      __float_as_uint(c.z0), __float_as_uint(c.zx));
    q[3] = make_uint4(__float_as_uint(c.zy), ((uint32_t)(ox - c.bx0) & 0xffffu) | ((uint32_t)(oy - c.by0) << 16), order_of(key),
                      (c.flags & 7u) | (fits32 ? kTeFits32 : 0u) | (valid ? kTeValid : 0u) | ((c.flags & 8u) ? kTeSmall : 0u));
#undef MUL256
}

// Adds `r`'s triangle to the per-tile counters (FILL = false) or claims its bin slots and writes
// its entries (FILL = true: `cov` = the triangle's coverage record, `entry` = its draw-order key).  Neighbouring triangles of a terrain tile mostly land in the same raster
// tile, so the common single-tile case is aggregated per wave: lanes that target the same tile elect
// a leader (ballot + readlane), which issues ONE atomic for all of them.  Must be called by every
// lane that is still in the caller's loop (wave-level operations inside).
template <bool FILL>
__device__ __forceinline__ void bin_rect(const RasterArgs& a, uint64_t r, uint32_t entry, uint32_t* __restrict__ counters_or_cursor,
                                         TileEntry* __restrict__ entries, const TriCov& cov)
{
    const int tx0 = (int)(r & 0xffffu), ty0 = (int)((r >> 16) & 0xffffu), tx1 = (int)((r >> 32) & 0xffffu), ty1 = (int)((r >> 48) & 0xffffu);
    const bool live = r != ~0ull;
    const bool single = live && tx0 == tx1 && ty0 == ty1;
    const int tile = (single && tile_owned(a, tx0, ty0)) ? ty0 * a.rtx + tx0 : -1;
    const int lane = (int)(threadIdx.x & 63u);
    unsigned long long todo = __ballot(tile >= 0);
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const int t = __builtin_amdgcn_readlane(tile, lead);
        const unsigned long long same = __ballot(tile == t);
        const uint32_t n = (uint32_t)__popcll(same);
        uint32_t base = 0;
        if (lane == lead) base = atomicAdd(&counters_or_cursor[t], n);
        if (FILL) {
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, lead);
            if (tile == t) {
                const uint32_t pos = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                if (pos < a.bin_capacity) write_tile_entry(a, entries + pos, cov, entry, tx0, ty0);
            }
        }
        todo &= ~same;
    }
    if (live && !single) {
        if (FILL && tx1 - tx0 <= 1 && ty1 - ty0 <= 1) {
            // up to 2 x 2 tiles (most of what is left: a terrain cell is about a tile wide): the slots of all four are
            // claimed before the first one is used - four atomics in flight instead of four round trips in a row
            uint32_t pos[4]; bool use[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int tx = tx0 + (q & 1), ty = ty0 + (q >> 1);
                use[q] = tx <= tx1 && ty <= ty1 && tile_owned(a, tx, ty);
                pos[q] = use[q] ? atomicAdd(&counters_or_cursor[ty * a.rtx + tx], 1u) : 0u;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) if (use[q] && pos[q] < a.bin_capacity) write_tile_entry(a, entries + pos[q], cov, entry, tx0 + (q & 1), ty0 + (q >> 1));
        } else
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++)
                if (tile_owned(a, tx, ty)) {
                    const uint32_t pos = atomicAdd(&counters_or_cursor[ty * a.rtx + tx], 1u);
                    if (FILL && pos < a.bin_capacity) write_tile_entry(a, entries + pos, cov, entry, tx, ty);
                }
    }
}

// ---------------------------------------------------------------------------------------
// clipper: triangles crossing z = 0 or leaving the guard band (rare)
// ---------------------------------------------------------------------------------------
struct ClipVert { float c[4]; float wx, wz; };

__device__ int clip_poly(ClipVert* poly, int n, int plane)
{
    ClipVert out[12];
    float d[12];
    int m = 0;
    for (int i = 0; i < n; i++) {
        const float* c = poly[i].c;
        switch (plane) {
        case 0: d[i] = c[2]; break;
        case 1: d[i] = kGuardBand * c[3] + c[0]; break;
        case 2: d[i] = kGuardBand * c[3] - c[0]; break;
        case 3: d[i] = kGuardBand * c[3] + c[1]; break;
        default: d[i] = kGuardBand * c[3] - c[1]; break;
        }
    }
    for (int i = 0; i < n; i++) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        const bool ini = d[i] >= 0.0f, inj = d[j] >= 0.0f;
        if (ini) out[m++] = poly[i];
        if (ini != inj) {
            // always interpolate from the inside vertex towards the outside vertex so both
            // triangles that share the edge create the identical vertex
            const ClipVert& va = ini ? poly[i] : poly[j];
            const ClipVert& vb = ini ? poly[j] : poly[i];
            const float da = ini ? d[i] : d[j], db = ini ? d[j] : d[i];
            const float tt = da / (da - db);
            ClipVert nv;
            for (int k = 0; k < 4; k++) nv.c[k] = va.c[k] + (vb.c[k] - va.c[k]) * tt;
            nv.wx = va.wx + (vb.wx - va.wx) * tt;
            nv.wz = va.wz + (vb.wz - va.wz) * tt;
            out[m++] = nv;
        }
    }
    for (int i = 0; i < m; i++) poly[i] = out[i];
    return m;
}

// Entries first .. n_hard (step `step`) of the hard list.
__device__ __forceinline__ void clip_hard_list(const RasterArgs& a, DevVert* __restrict__ verts, uint32_t* __restrict__ counters,
                                            const uint32_t* __restrict__ hard_list, HardTriRec* __restrict__ hard_tris,
                                            uint32_t* __restrict__ hard_first, uint32_t* __restrict__ tile_count,
                                            uint4* __restrict__ hard_recs, uint32_t first, uint32_t n_hard, uint32_t step)
{
    for (uint32_t h = first; h < n_hard; h += step) {
        const uint32_t tri = hard_list[h];
        uint32_t idx[3];
        regular_tri_indices(tri, idx[0], idx[1], idx[2]);
        ClipVert poly[12];
        bool need_near = false;
        for (int k = 0; k < 3; k++) {
            const DevVert& v = verts[idx[k]];
            poly[k].c[0] = v.cx; poly[k].c[1] = v.cy; poly[k].c[2] = v.cz; poly[k].c[3] = v.cw;
            poly[k].wx = v.wx; poly[k].wz = v.wz;
            need_near |= v.cz < 0.0f;
        }
        int n = 3;
        if (need_near) n = clip_poly(poly, n, 0);
        if (n >= 3) {
            bool need_guard = false;
            for (int i = 0; i < n; i++) {
                const float g = kGuardBand * poly[i].c[3];
                need_guard |= (poly[i].c[0] < -g) || (poly[i].c[0] > g) || (poly[i].c[1] < -g) || (poly[i].c[1] > g);
            }
            if (need_guard) for (int pl = 1; pl <= 4 && n >= 3; pl++) n = clip_poly(poly, n, pl);
        }
        if (n < 3) { hard_first[tri] = 0xffffffffu; continue; }
        const uint32_t nsub = (uint32_t)(n - 2);
        const uint32_t vbase = atomicAdd(&counters[C_XVERTS], (uint32_t)n);
        const uint32_t tbase = atomicAdd(&counters[C_HARDTRIS], nsub);
        if (vbase + (uint32_t)n > a.extra_vert_cap || tbase + nsub > a.hard_cap * 4u) {
            atomicOr(&counters[C_FLAGS], 2u); hard_first[tri] = 0xffffffffu; continue;
        }
        for (int i = 0; i < n; i++) {
            DevVert o;
            o.cx = poly[i].c[0]; o.cy = poly[i].c[1]; o.cz = poly[i].c[2]; o.cw = poly[i].c[3];
            o.wx = poly[i].wx; o.wz = poly[i].wz; o.pad0 = 0; o.pad1 = 0;
            snap_vertex(o, a.vp_x, a.vp_y, a.vp_w, a.vp_h);
            verts[a.extra_vert_base + vbase + (uint32_t)i] = o;
        }
        hard_first[tri] = tbase;
        for (uint32_t s = 0; s < nsub; s++) {
            HardTriRec rec;
            rec.v0 = a.extra_vert_base + vbase; rec.v1 = rec.v0 + s + 1; rec.v2 = rec.v0 + s + 2;
            rec.order_key = (tri << 4) | (s << 1) | 1u;
            const uint64_t r = triangle_rect(a, load_sv(verts, rec.v0), load_sv(verts, rec.v1), load_sv(verts, rec.v2),
                                             hard_recs + (size_t)(tbase + s) * kRecGroups);
            if (r != ~0ull) {       // rare path, divergent loop: plain per-tile atomics
                const int qx0 = (int)(r & 0xffffu), qy0 = (int)((r >> 16) & 0xffffu), qx1 = (int)((r >> 32) & 0xffffu), qy1 = (int)((r >> 48) & 0xffffu);
                for (int ty = qy0; ty <= qy1; ty++)
                    for (int tx = qx0; tx <= qx1; tx++)
                        if (tile_owned(a, tx, ty)) atomicAdd(&tile_count[ty * a.rtx + tx], 1u);
            }
            rec.rect_lo = (uint32_t)r; rec.rect_hi = (uint32_t)(r >> 32); rec.pad0 = 0; rec.pad1 = 0;
            hard_tris[tbase + s] = rec;
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_scan: every raster tile's slice of the bin-entry array, and the tile pass's launch order
// ---------------------------------------------------------------------------------------
// A tile needs a slice of `entries` as long as its count - not the slice an ordered prefix sum would give it.  So the
// scan is local: a workgroup takes 1024 tiles, scans their counts in registers / shuffles / a few words of LDS, and
// claims the space for all of them with ONE atomic on the frame's entry total; nothing waits for a neighbour.
// (Round 2: one workgroup of 1024 threads walked the whole frame, 33 us alone and twice that beside the tile pass,
// whose workgroups left no CU with room for sixteen waves at once.)
// It also orders the tiles the tile pass will launch over (all of them, or this rank's `cand` list) by falling bin
// length (eight classes; empty bins - the sky - last): workgroups are handed out in launch order, so the
// expensive tiles start first and the cheap ones fill the tail of the launch (measured: -2.5 % on the 8K tile pass).
// Every class has its own region of `order` (class c from c * stride); a workgroup claims its tiles' places in each
// with one atomic per class, and the tile pass finds workgroup i's tile from the eight class totals (tile_of_block).
constexpr int kScanClasses = 8;
constexpr int kScanPerThread = 4, kScanPerGroup = 256 * kScanPerThread;
__device__ __forceinline__ int scan_class(uint32_t cnt)
{
    if (cnt == 0u) return 7;
    const int k = 31 - __clz((int)cnt);                  // floor(log2): >= 256 entries -> 0, 128.. -> 1, 64.. -> 2, 32.. -> 3, 16.. -> 4, 4.. -> 5, 1.. -> 6
    return k >= 8 ? 0 : (k >= 4 ? 8 - k : (k >= 2 ? 5 : 6));
}
__global__ __launch_bounds__(256) void k_scan(int n_tiles, uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_offset,
                                               uint32_t* __restrict__ tile_cursor, uint32_t* __restrict__ counters, uint32_t capacity,
                                               const int32_t* __restrict__ cand, int n_cand, int32_t* __restrict__ order)
{
    VR_GEOMETRY_PRIORITY();
    constexpr int NW = 4;
    __shared__ uint32_t s_wsum[NW];                       // per wave: sum of its threads' counts -> first entry of the wave's tiles
    __shared__ uint32_t s_cls[kScanClasses * NW];         // per (class, wave): tiles of that class -> first place in `order`
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // A rank's bins are the candidates' only (k_setup / k_fill count owned tiles alone): scan that list, not the frame.
    const int n = cand ? n_cand : n_tiles;
    const int b = min((int)blockIdx.x * kScanPerGroup + tid * kScanPerThread, n), e = min(b + kScanPerThread, n);
    int t[kScanPerThread]; uint32_t cnt[kScanPerThread];
#pragma unroll
    for (int j = 0; j < kScanPerThread; j++) t[j] = b + j < e ? (cand ? cand[b + j] : b + j) : -1;
#pragma unroll
    for (int j = 0; j < kScanPerThread; j++) cnt[j] = t[j] >= 0 ? tile_count[t[j]] : 0u;
    uint32_t s = 0, cc[kScanClasses];
#pragma unroll
    for (int c = 0; c < kScanClasses; c++) cc[c] = 0u;
#pragma unroll
    for (int j = 0; j < kScanPerThread; j++) {
        if (t[j] < 0) continue;
        s += cnt[j];
        const int cl = scan_class(cnt[j]);
#pragma unroll
        for (int c = 0; c < kScanClasses; c++) cc[c] += cl == c ? 1u : 0u;
    }
    // inclusive scans inside the wave
    uint32_t incl = s, ci[kScanClasses];
#pragma unroll
    for (int c = 0; c < kScanClasses; c++) ci[c] = cc[c];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += v;
#pragma unroll
        for (int c = 0; c < kScanClasses; c++) { const uint32_t w = (uint32_t)__shfl_up((int)ci[c], d); if (lane >= d) ci[c] += w; }
    }
    if (lane == 63) {
        s_wsum[wave] = incl;
#pragma unroll
        for (int c = 0; c < kScanClasses; c++) s_cls[c * NW + wave] = ci[c];
    }
    __syncthreads();
    // nine lanes, one quantity each: the workgroup's total (entries / tiles of a class), its share of the frame's (one
    // atomic), and the four waves' first places
    if (tid <= kScanClasses) {
        uint32_t* __restrict__ w = tid < kScanClasses ? &s_cls[tid * NW] : s_wsum;
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], total = (w0 + w1) + (w2 + w3);
        uint32_t base = total ? atomicAdd(&counters[tid < kScanClasses ? C_CLASS0 + tid : C_BINTOTAL], total) : 0u;
        if (tid == kScanClasses) { if (base + total > capacity) atomicOr(&counters[C_FLAGS], 2u); }
        else base += (uint32_t)tid * (uint32_t)n_tiles;   // the class's own region of `order`
        w[0] = base; w[1] = base + w0; w[2] = base + (w0 + w1); w[3] = base + (w0 + w1) + w2;
    }
    __syncthreads();
    uint32_t run = s_wsum[wave] + (incl - s);
    uint32_t slot[kScanClasses];
#pragma unroll
    for (int c = 0; c < kScanClasses; c++) slot[c] = s_cls[c * NW + wave] + (ci[c] - cc[c]);
    // the counts are consumed here: zero them for the next frame (saves a memset per frame); the
    // rasteriser gets a bin's length from cursor - offset once k_fill has run
#pragma unroll
    for (int j = 0; j < kScanPerThread; j++) {
        if (t[j] < 0) continue;
        tile_offset[t[j]] = run; tile_cursor[t[j]] = run; run += cnt[j]; tile_count[t[j]] = 0u;
        const int cl = scan_class(cnt[j]);
        uint32_t pos = 0u;
#pragma unroll
        for (int c = 0; c < kScanClasses; c++) if (cl == c) { pos = slot[c]; slot[c] = pos + 1u; }
        order[pos] = t[j];
    }
}
// The tile of the tile pass's workgroup `i`: classes in falling bin length, class c's tiles at order[c * stride ...].
template <typename OP, typename CP>
__device__ __forceinline__ int tile_of_block(int i, OP order, CP class_totals, int stride)
{
    int c = 0;
#pragma unroll
    for (int q = 0; q < kScanClasses - 1; q++) { const int k = (int)class_totals[q]; if (c == q && i >= k) { i -= k; c = q + 1; } }
    return order[(size_t)c * stride + i];
}

// ---------------------------------------------------------------------------------------
// k_setup: regular triangles
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_setup(RasterArgs a, const DevVert* __restrict__ verts, uint32_t* __restrict__ counters,
                                                uint64_t* __restrict__ rect, uint32_t* __restrict__ hard_list,
                                                uint32_t* __restrict__ tile_count, uint4* __restrict__ recs)
{
    VR_GEOMETRY_PRIORITY();
    const uint32_t total = counters[C_COUNT] * (uint32_t)kTrisPerInst;
    for (uint32_t tri = blockIdx.x * blockDim.x + threadIdx.x; tri < total; tri += gridDim.x * blockDim.x) {
        uint32_t i0, i1, i2;
        regular_tri_indices(tri, i0, i1, i2);
        const float4 c0 = *reinterpret_cast<const float4*>(&verts[i0].cx);
        const float4 c1 = *reinterpret_cast<const float4*>(&verts[i1].cx);
        const float4 c2 = *reinterpret_cast<const float4*>(&verts[i2].cx);
        // trivial reject against the six clip planes
        const bool out_l = (c0.x < -c0.w) && (c1.x < -c1.w) && (c2.x < -c2.w);
        const bool out_r = (c0.x > c0.w) && (c1.x > c1.w) && (c2.x > c2.w);
        const bool out_b = (c0.y < -c0.w) && (c1.y < -c1.w) && (c2.y < -c2.w);
        const bool out_t = (c0.y > c0.w) && (c1.y > c1.w) && (c2.y > c2.w);
        const bool out_n = (c0.z < 0.0f) && (c1.z < 0.0f) && (c2.z < 0.0f);
        const bool out_f = (c0.z > c0.w) && (c1.z > c1.w) && (c2.z > c2.w);
        uint64_t r = ~0ull;
        if (!(out_l || out_r || out_b || out_t || out_n || out_f)) {
            const bool need_near = (c0.z < 0.0f) || (c1.z < 0.0f) || (c2.z < 0.0f);
            const float g0 = kGuardBand * c0.w, g1 = kGuardBand * c1.w, g2 = kGuardBand * c2.w;
            const bool need_guard = (c0.x < -g0) || (c0.x > g0) || (c0.y < -g0) || (c0.y > g0)
                                 || (c1.x < -g1) || (c1.x > g1) || (c1.y < -g1) || (c1.y > g1)
                                 || (c2.x < -g2) || (c2.x > g2) || (c2.y < -g2) || (c2.y > g2);
            if (need_near || need_guard) {
                const uint32_t slot = atomicAdd(&counters[C_HARDLIST], 1u);
                if (slot < a.hard_cap) hard_list[slot] = tri; else atomicOr(&counters[C_FLAGS], 2u);
            } else {
                r = triangle_rect(a, load_sv(verts, i0), load_sv(verts, i1), load_sv(verts, i2), recs + (size_t)tri * kRecGroups);
            }
        }
        bin_rect<false>(a, r, 0u, tile_count, nullptr, TriCov());
        rect[tri] = r;
    }
}

// ---------------------------------------------------------------------------------------
// k_clip: the clipper over the hard list
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_clip(RasterArgs a, DevVert* __restrict__ verts, uint32_t* __restrict__ counters,
                                              const uint32_t* __restrict__ hard_list, HardTriRec* __restrict__ hard_tris,
                                              uint32_t* __restrict__ hard_first, uint32_t* __restrict__ tile_count,
                                              uint4* __restrict__ hard_recs)
{
    VR_GEOMETRY_PRIORITY();
    const uint32_t n_hard = min(counters[C_HARDLIST], a.hard_cap);
    clip_hard_list(a, verts, counters, hard_list, hard_tris, hard_first, tile_count, hard_recs, blockIdx.x * blockDim.x + threadIdx.x, n_hard,
                   gridDim.x * blockDim.x);
}

// ---------------------------------------------------------------------------------------
// k_fill: write bin entries - one TileEntry per (triangle, raster tile) pair: the triangle set up against that tile, with its
// draw-order key (triangle id << 4) | (sub << 1) | hard.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill(RasterArgs a, const uint32_t* __restrict__ counters, const uint64_t* __restrict__ rect,
                                               const HardTriRec* __restrict__ hard_tris, const uint4* __restrict__ recs, uint32_t rec_hard_base,
                                               uint32_t* __restrict__ tile_cursor, TileEntry* __restrict__ entries, uint32_t* __restrict__ status)
{
    VR_GEOMETRY_PRIORITY();
    // the chain's counters (node count, status flags, work-list lengths: all final before this launch) into the terrain's pinned
    // host mirror: the host reads them without a wait once the chain's event has completed (vr_terrain_poll)
    if (blockIdx.x == 0 && threadIdx.x < 8) { __hip_atomic_store(&status[threadIdx.x], counters[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    const uint32_t n_reg = counters[C_COUNT] * (uint32_t)kTrisPerInst;
    const uint32_t n_hard = min(counters[C_HARDTRIS], a.hard_cap * 4u);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_reg + n_hard; i += gridDim.x * blockDim.x) {
        uint64_t r; uint32_t entry; size_t rec;
        if (i < n_reg) { r = rect[i]; entry = i << 4; rec = i; }
        else {
            const HardTriRec h = hard_tris[i - n_reg];
            r = (uint64_t)h.rect_lo | ((uint64_t)h.rect_hi << 32); entry = h.order_key;
            rec = (size_t)rec_hard_base + (i - n_reg);       // the clipper's records, in the order of its sub-triangles (clip_hard_list)
        }
        TriCov cov = TriCov();
        if (r != ~0ull) cov = load_tri_cov(recs + rec * kRecGroups);
        bin_rect<true>(a, r, entry, tile_cursor, entries, cov);
    }
}

// ---------------------------------------------------------------------------------------
// k_raster: one workgroup per 64x64 (or 32x32) raster tile
// ---------------------------------------------------------------------------------------
// record index of a bin entry: the triangle id, or - for clipper output - its slot in the records' extra region
template <typename HF>
__device__ __forceinline__ size_t rec_index(uint32_t key, HF hard_first, uint32_t rec_hard_base)
{
    const uint32_t tri = key >> 4;
    if (key & 1u) return (size_t)rec_hard_base + hard_first[tri] + ((key >> 1) & 7u);
    return (size_t)tri;
}

template <typename HF>
__device__ __forceinline__ void entry_vertices(uint32_t key, const HardTriRec* __restrict__ hard_tris, HF hard_first,
                                               uint32_t& i0, uint32_t& i1, uint32_t& i2)
{
    const uint32_t tri = key >> 4;
    if (key & 1u) {
        const HardTriRec rec = hard_tris[hard_first[tri] + ((key >> 1) & 7u)];
        i0 = rec.v0; i1 = rec.v1; i2 = rec.v2;
    } else {
        regular_tri_indices(tri, i0, i1, i2);
    }
}

__device__ __forceinline__ int64_t edge_eval(int32_t A, int32_t B, int64_t C, int32_t PX, int32_t PY)
{
    return (int64_t)A * PX + ((int64_t)B * PY + C);
}
__device__ __forceinline__ int64_t rec_c(const uint4& g) { return (int64_t)(((uint64_t)g.w << 32) | (uint64_t)g.z); }

struct Attr { float wx, wz, dwxdx, dwzdx, dwxdy, dwzdy; };

// (x + half) / world_size; when world_size is a power of two the division is an exact
// scaling, so the multiplication by its reciprocal gives the identical float.
__device__ __forceinline__ float div_ws(const RasterArgs& a, float s)
{
    return a.ws_pow2 ? s * a.inv_world_size : s / a.world_size;
}
__device__ __forceinline__ float to_uv(const RasterArgs& a, float x) { return div_ws(a, x + a.world_size * 0.5f); }

// One axis of a bilinear footprint: texel-space coordinate -> integer floor (clamped to [-1, n], as the quad table is
// indexed) and fraction.  Same operations as quad_tap / the oracle's tex_bilinear (u * n - 0.5 fused), once per distinct coordinate.
struct Axis { int i; float f; };
__device__ __forceinline__ Axis tap_axis(int n, float t)
{
    const float x = __builtin_fmaf(t, (float)n, -0.5f);
    float xf = floorf(x);
    Axis r; r.f = x - xf;
    r.i = (int)vr_clampf(xf, -1.0f, (float)n);
    return r;
}
// NaN never reaches the normal's encode (the vector is normalised from a length >= 0.2): clamp, scale, round half away
__device__ __forceinline__ uint32_t snorm16_finite(float v)
{
    const float s = vr_clampf(v, -1.0f, 1.0f) * 32767.0f;
    return (uint32_t)(int)(s + copysignf(0.5f, s)) & 0xffffu;      // == (s >= 0 ? s + 0.5 : s - 0.5) for every finite s
}

// A component of a NORMALISED vector (|v| <= 1 + a few ulps): v * 32767 rounds below 32767.5 without the clamp, so
// (int)(s + copysign(0.5, s)) is vr_snorm16's result bit for bit; POSITIVE: the component is known to be > 0 (the normal's y).
template <bool POSITIVE>
__device__ __forceinline__ uint32_t snorm16_unit(float v)
{
    const float s = v * 32767.0f;
    return (uint32_t)(int)(POSITIVE ? s + 0.5f : s + copysignf(0.5f, s)) & 0xffffu;
}

// One mip level of main_ps's five taps (terrain_ps.hlsl:59-61, :68): four height taps at uv +- 0.1 through the quad
// table and the albedo footprint at uv.  SAME: heightmap and albedo have the same size (the reference's case), so the
// albedo footprint is the floor / fraction pair the height taps at the unshifted u and v already computed.
template <bool SAME>
__device__ __forceinline__ void sample_level(const DevTex& hm, const DevTex& al, __amdgpu_buffer_rsrc_t rq, __amdgpu_buffer_rsrc_t rc,
                                             const uint32_t* __restrict__ qoff, const uint32_t* __restrict__ aoff, int lvl_h, int lvl_c, float ua, float ub, float va, float vb,
                                             float u0, float v0, float u, float v, float hgt[4], float col[3])
{
    const int w = max(1, hm.w0 >> lvl_h), h = max(1, hm.h0 >> lvl_h);
    const uint32_t q = qoff[lvl_h], c = aoff[lvl_c];             // dword offset of the quad table, byte offset of the albedo level
    const Axis xa = tap_axis(w, ua), xb = tap_axis(w, ub), x0 = tap_axis(w, u0), y0 = tap_axis(h, v0), ya = tap_axis(h, va), yb = tap_axis(h, vb);
    const int r0 = __mul24(y0.i + 1, w + 2) + 1, ra = __mul24(ya.i + 1, w + 2) + 1, rb = __mul24(yb.i + 1, w + 2) + 1;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDQ(i) __builtin_amdgcn_raw_buffer_load_b128(rq, (q + (uint32_t)(i)) << 4, 0, 0)
    const u32x4 e0 = LDQ(r0 + xa.i), e1 = LDQ(r0 + xb.i), e2 = LDQ(ra + x0.i), e3 = LDQ(rb + x0.i);
    const uint32_t c4 = c << 2;                                   // the decoded chain: 16 B per texel, levels at 4 x the byte offset
#define LDC(i) __builtin_amdgcn_raw_buffer_load_b128(rc, ((uint32_t)(i) << 4) + c4, 0, 0)
    u32x4 p00, p10, p01, p11;
    float cfx, cfy;
    if (SAME) {
        const int cx0 = vr_clampi(x0.i, 0, w - 1), cx1 = vr_clampi(x0.i + 1, 0, w - 1), cy0 = vr_clampi(y0.i, 0, h - 1), cy1 = vr_clampi(y0.i + 1, 0, h - 1);
        const int q0 = __mul24(cy0, w), q1 = __mul24(cy1, w);
        p00 = LDC(q0 + cx0); p10 = LDC(q0 + cx1); p01 = LDC(q1 + cx0); p11 = LDC(q1 + cx1);
        cfx = x0.f; cfy = y0.f;
    } else {
        const BilinearSetup s = vr_bilinear_setup(max(1, al.w0 >> lvl_c), max(1, al.h0 >> lvl_c), u, v);
        p00 = LDC(s.i00); p10 = LDC(s.i10); p01 = LDC(s.i01); p11 = LDC(s.i11);
        cfx = s.fx; cfy = s.fy;
    }
#undef LDQ
#undef LDC
    // entry = (t00, t10 - t00, t01, t11 - t01): top = fma(t10 - t00, fx, t00), bot likewise, then across y (fused lerps, oracle: tex_bilinear)
#define QF(e, fx, fy) ({ const float top_ = __builtin_fmaf(__uint_as_float((e).y), (fx), __uint_as_float((e).x)), \
                                     bot_ = __builtin_fmaf(__uint_as_float((e).w), (fx), __uint_as_float((e).z)); \
                         __builtin_fmaf(bot_ - top_, (fy), top_); })
    hgt[0] = QF(e0, xa.f, y0.f); hgt[1] = QF(e1, xb.f, y0.f); hgt[2] = QF(e2, x0.f, ya.f); hgt[3] = QF(e3, x0.f, yb.f);
#undef QF
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float t00 = __uint_as_float(p00[k]), t10 = __uint_as_float(p10[k]), t01 = __uint_as_float(p01[k]), t11 = __uint_as_float(p11[k]);
        const float top = __builtin_fmaf(t10 - t00, cfx, t00), bot = __builtin_fmaf(t11 - t01, cfx, t01);
        col[k] = __builtin_fmaf(bot - top, cfy, top);
    }
}

// main_ps (terrain_ps.hlsl:45-82) -> encoded render-target texels.  qoff / aoff: LDS copies of the
// per-level offset tables (no dependent global load in front of a texel fetch).  The finer level of
// the four height taps and of the albedo tap is fetched as one batch of 8 loads; the coarser level
// (only when a LOD fraction is non-zero) as a second batch.
// Texels are fetched through buffer resources (rq: the quad tables, rc: the albedo chain): a fetch's address is one
// 32-bit byte offset (1 VALU) instead of a 64-bit pointer sum (3), and an out-of-range offset reads 0 instead of faulting.
// POW2: the world size is a power of two (known at compile time in the fast variant of the tile pass): "/ world_size" is one
// multiplication, with no run-time choice between the two forms - six scalar branches per pixel otherwise.
template <bool SAME, bool POW2>
__device__ __forceinline__ void pixel_shader(const RasterArgs& a, const DevTex& hm, const DevTex& al, __amdgpu_buffer_rsrc_t rq,
                                             __amdgpu_buffer_rsrc_t rc, const float* __restrict__ thr, const uint8_t* __restrict__ enc,
                                             const uint32_t* __restrict__ qoff, const uint32_t* __restrict__ aoff, const Attr& p,
                                             uint32_t& diffuse, uint32_t& n01, uint32_t& n23)
{
#define DIVWS(x) (POW2 ? (x) * a.inv_world_size : div_ws(a, (x)))
    const float half_ws = a.world_size * 0.5f;
    const float u = DIVWS(p.wx + half_ws), v = DIVWS(p.wz + half_ws);                        // :12-13, :20-21
    const float dudx = DIVWS(p.dwxdx), dvdx = DIVWS(p.dwzdx), dudy = DIVWS(p.dwxdy), dvdy = DIVWS(p.dwzdy);
#undef DIVWS
    const float lod_h = vr_lod_from_derivs(dudx, dvdx, dudy, dvdy, hm.w0, hm.h0);
    const LodSplit lh = vr_lod_split(hm.levels, lod_h);
    LodSplit lc = lh;                                            // same size and level count: the same LOD
    if (!SAME) lc = vr_lod_split(al.levels, vr_lod_from_derivs(dudx, dvdx, dudy, dvdy, al.w0, al.h0));
    const float offset = 0.1f;                                                              // :59
    // (uv + float2(offset, 0.0)'s "+ 0.0" only turns -0 into +0, which no later operation can tell apart: fma(+-0, n, -0.5) = -0.5)
    const float ua = u + offset, ub = u + (-offset), va = v + offset, vb = v + (-offset), u0 = u, v0 = v;
    float hgt[4], col[3];
    sample_level<SAME>(hm, al, rq, rc, qoff, aoff, lh.l0, lc.l0, ua, ub, va, vb, u0, v0, u, v, hgt, col);
    // Wave-uniform branch (ballot): where every pixel of the wave is magnified (LOD 0 - the near half of an 8K frame)
    // the whole second level is skipped; a per-lane condition gets if-converted and every pixel pays for both levels.
    if (__any(lh.f > 0.0f || lc.f > 0.0f)) {
        // blending with a zero fraction returns the finer sample exactly, so one branch serves both textures
        const int l1h = min(lh.l0 + 1, hm.levels - 1), l1c = min(lc.l0 + 1, al.levels - 1);
        // The coordinates pass through an empty asm so that nothing of this level can be hoisted above the branch
        // (its loads and arithmetic are speculatable; left alone the compiler may run both levels for every pixel).
        float xa = ua, xb = ub, ya = va, yb = vb, x0 = u0, y0 = v0, xu = u, yv = v;
        asm volatile("" : "+v"(xa), "+v"(xb), "+v"(ya), "+v"(yb), "+v"(x0), "+v"(y0), "+v"(xu), "+v"(yv));
        float g[4], cb[3];
        sample_level<SAME>(hm, al, rq, rc, qoff, aoff, l1h, l1c, xa, xb, ya, yb, x0, y0, xu, yv, g, cb);
        // fma(b - a, f, a): with f == 0 this is a exactly (b - a is finite), as the oracle's "f > 0" branch leaves it
#pragma unroll
        for (int k = 0; k < 4; k++) hgt[k] = __builtin_fmaf(g[k] - hgt[k], lh.f, hgt[k]);
#pragma unroll
        for (int k = 0; k < 3; k++) col[k] = __builtin_fmaf(cb[k] - col[k], lc.f, col[k]);
    }
    const float hDx = hgt[0] - hgt[1], hDy = hgt[2] - hgt[3];                                // :60-61
    float nx = -hDx, ny = 2.0f * offset, nz = -hDy;                                          // :63
    // normalize(): 1.0f / sqrtf(fma(nz, nz, fma(nx, nx, ny * ny))), length in [0.2, 1.43]
    const float inv = vr_rcp_exact(vr_sqrt_exact(__builtin_fmaf(nz, nz, __builtin_fmaf(nx, nx, ny * ny))));
    nx *= inv; ny *= inv; nz *= inv;
    diffuse = vr_srgb_encode_fast(col[0], thr, enc) | (vr_srgb_encode_fast(col[1], thr, enc) << 8) | (vr_srgb_encode_fast(col[2], thr, enc) << 16)
            | 0xff000000u;                                                                  // :68, :73-75
    n01 = snorm16_finite(nx) | (snorm16_finite(ny) << 16);                                  // :78
    n23 = snorm16_finite(nz) | (32767u << 16);                                              // :79 roughness = 1
}

// Phase clocks for experiments (tools/build_variant.py prof -DVR_RASTER_PROFILE; needs -DVR_EXPERIMENT_BUILD, vr_experiments.h); not in the product build.
#ifdef VR_RASTER_PROFILE
constexpr int kProfBlocks = 16384;
constexpr int kProfSlots = 16;
__device__ unsigned long long g_raster_prof[kProfBlocks * kProfSlots];
// every wave's lane 0 adds its phase time to its block's slot (LDS-free, 4 waves per address)
#define VR_PROF_MARK(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < kProfBlocks) { const unsigned long long now_ = __builtin_readcyclecounter(); \
                                 atomicAdd(&g_raster_prof[blockIdx.x * kProfSlots + (i)], now_ - prof_t_); prof_t_ = __builtin_readcyclecounter(); } } while (0)
#define VR_PROF_BEGIN unsigned long long prof_t_ = __builtin_readcyclecounter()
#define VR_PROF_PARAM , unsigned long long& prof_t_
#define VR_PROF_ARG , prof_t_
#define VR_PROF_DRAIN asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define VR_PROF_ADD(i, v) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < kProfBlocks) atomicAdd(&g_raster_prof[blockIdx.x * kProfSlots + (i)], (unsigned long long)(v)); } while (0)
extern "C" VR_API int vr_debug_raster_prof(unsigned long long out[16], int reset)
{
    static unsigned long long host[kProfBlocks * kProfSlots];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_raster_prof), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < kProfSlots; i++) out[i] = 0;
    for (int b = 0; b < kProfBlocks; b++) for (int i = 0; i < kProfSlots; i++) out[i] += host[b * kProfSlots + i];
    if (reset) { memset(host, 0, sizeof(host)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_raster_prof), host, sizeof(host)) != hipSuccess) return -1; }
    return 0;
}
#else
#define VR_PROF_MARK(i) do { } while (0)
#define VR_PROF_BEGIN do { } while (0)
#define VR_PROF_PARAM
#define VR_PROF_ARG
#define VR_PROF_DRAIN do { } while (0)
#define VR_PROF_ADD(i, v) do { } while (0)
#endif

// ---- the fast variant's pixel shader -------------------------------------------------------------------------------
// Same arithmetic as pixel_shader<true, true> (bit for bit), arranged for what this part's vector pipes issue quickly
// (tools/micro/valu_cost.hip, profiles/r03_valu_issue_costs.txt: at four waves per SIMD a v_mul / v_add / v_sub / v_fmac /
// v_and / v_add_u32 costs ~1.2 cycles, a three-operand v_fma 2, and conversions, v_floor, v_med3 / min / max, shifts,
// 24-bit mads, compares and selects ~3.2 - with two wait states between a compare and the select that reads it):
//   - heights and albedo are read from the clamp-padded tables of DevTex::fast: one set of byte offsets serves the four
//     height taps and the albedo footprint, built from three row products and fast adds; no clamp, no shift - an index
//     is scaled by 16 as a FLOAT (exact) before its conversion, right / lower neighbours are immediate offsets;
//   - what a pixel needs of a mip level (table offset, row bytes, float sizes) is one 16-byte LDS read;
//   - the implicit LOD has no compare / select (oracle: lod_from_derivs), its clamp is one v_med3;
//   - no range check in front of the short reciprocal: a pixel inside its triangle interpolates 1/w between its vertices'.
__device__ __forceinline__ uint32_t srgb_encode_nonneg(float x, const float* __restrict__ thr, const uint8_t* __restrict__ tab)
{
    // vr_srgb_encode_fast for x in [0, 1] (a bilinear blend of decoded texels): no sign / NaN case, no upper clamp
    const int b = max((int)(__float_as_uint(x) >> 16) - kEncTabBase, 0);
    const uint32_t g = tab[b];
    return g + (x >= thr[g + 1u] ? 1u : 0u);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct FastTaps { u32x4 e0, e1, e2, e3, p00, p10, p01, p11; float fxa, fxb, fx0, fy0, fya, fyb; };

// One axis: texel-space coordinate -> fraction and the clamped floor, scaled by `scale` (16 for columns: a byte offset;
// 1 for rows) while it is still a float (exact: |floor| <= 16385).
#define FAST_AXIS(f_, i_, t_, n_, scale_) do { const float x_ = __builtin_fmaf((t_), (n_), -0.5f); const float xf_ = floorf(x_); (f_) = x_ - xf_; \
                                                (i_) = (int)(vr_clampf(xf_, -1.0f, (n_)) * (scale_)); } while (0)

// The same for the UNSHIFTED coordinates u, v of a pixel of the terrain: the interpolated world position lies inside its
// triangle, i.e. on the terrain, so u, v are in [0, 1] to within rounding and floor(u * n - 0.5) in [-1, n - 1] - already inside
// the clamp's range [-1, n] (it would take u < -0.5 / n or u >= 1 + 0.5 / n: 2e-4 beyond the terrain's edge on the finest level).
#define FAST_AXIS_INSIDE(f_, i_, t_, n_, scale_) do { const float x_ = __builtin_fmaf((t_), (n_), -0.5f); const float xf_ = floorf(x_); (f_) = x_ - xf_; \
                                                       (i_) = (int)(xf_ * (scale_)); } while (0)

__device__ __forceinline__ FastTaps issue_level_fast(__amdgpu_buffer_rsrc_t rq, __amdgpu_buffer_rsrc_t rc, const uint4* __restrict__ s_lv, int lvl16,
                                                     float ua, float ub, float va, float vb, float u0, float v0)
{
    FastTaps t;
    const uint4 L = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(s_lv) + lvl16);     // { offset of entry (1, 1), row bytes, (float)w, (float)h }
    const float wf = __uint_as_float(L.z), hf = __uint_as_float(L.w);
    int xa, xb, x0, y0, ya, yb;
    FAST_AXIS(t.fxa, xa, ua, wf, 16.0f); FAST_AXIS(t.fxb, xb, ub, wf, 16.0f); FAST_AXIS_INSIDE(t.fx0, x0, u0, wf, 16.0f);
    FAST_AXIS_INSIDE(t.fy0, y0, v0, hf, 1.0f); FAST_AXIS(t.fya, ya, va, hf, 1.0f); FAST_AXIS(t.fyb, yb, vb, hf, 1.0f);
    const uint32_t row0 = L.x + (uint32_t)__mul24(y0, (int)L.y), rowa = L.x + (uint32_t)__mul24(ya, (int)L.y), rowb = L.x + (uint32_t)__mul24(yb, (int)L.y);
    // (kExp*: timing experiments, all false in a product build - vr_experiments.h)
    const uint32_t am = kExpSameAddr ? 0xf0u : ~0u;
#define LD(r_, o_) __builtin_amdgcn_raw_buffer_load_b128((r_), (o_) & am, 0, 0)
    const uint32_t a00 = row0 + (uint32_t)x0, a01 = a00 + L.y;
    t.e0 = LD(rq, row0 + (uint32_t)xa);
    if (!kExpOneHeight) { t.e1 = LD(rq, row0 + (uint32_t)xb); t.e2 = LD(rq, rowa + (uint32_t)x0); t.e3 = LD(rq, rowb + (uint32_t)x0); }
    t.p00 = LD(rc, a00);
    if (!(kExpOneHeight || kExpOneAlbedo)) { t.p10 = LD(rc, a00 + 16u); t.p01 = LD(rc, a01); t.p11 = LD(rc, a01 + 16u); }
    if (kExpOneHeight) { t.e1 = t.e0; t.e2 = t.e0; t.e3 = t.e0; }
    if (kExpOneHeight || kExpOneAlbedo) { t.p10 = t.p00; t.p01 = t.p00; t.p11 = t.p00; }
#undef LD
    return t;
}

__device__ __forceinline__ void filter_level_fast(const FastTaps& t, float hgt[4], float col[3])
{
#define QF(e, fx, fy) ({ const float top_ = __builtin_fmaf(__uint_as_float((e).y), (fx), __uint_as_float((e).x)), \
                                     bot_ = __builtin_fmaf(__uint_as_float((e).w), (fx), __uint_as_float((e).z)); \
                         __builtin_fmaf(bot_ - top_, (fy), top_); })
    hgt[0] = QF(t.e0, t.fxa, t.fy0); hgt[1] = QF(t.e1, t.fxb, t.fy0); hgt[2] = QF(t.e2, t.fx0, t.fya); hgt[3] = QF(t.e3, t.fx0, t.fyb);
#undef QF
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float t00 = __uint_as_float(t.p00[k]), t10 = __uint_as_float(t.p10[k]), t01 = __uint_as_float(t.p01[k]), t11 = __uint_as_float(t.p11[k]);
        const float top = __builtin_fmaf(t10 - t00, t.fx0, t00), bot = __builtin_fmaf(t11 - t01, t.fx0, t01);
        col[k] = __builtin_fmaf(bot - top, t.fy0, top);
    }
}

__device__ __forceinline__ void pixel_shader_fast(const RasterArgs& a, __amdgpu_buffer_rsrc_t rq, __amdgpu_buffer_rsrc_t rc,
                                                  const float* __restrict__ thr, const uint8_t* __restrict__ enc, const uint4* __restrict__ s_lv, const Attr& p,
                                                  uint32_t& diffuse, uint32_t& n01, uint32_t& n23 VR_PROF_PARAM)
{
    const float half_ws = a.world_size * 0.5f, iws = a.inv_world_size;
    const float u = (p.wx + half_ws) * iws, v = (p.wz + half_ws) * iws;                       // :12-13, :20-21 (power-of-two world size)
    const float max_level = a.max_level_f;
    const float lod = vr_lod_from_derivs_f(p.dwxdx, p.dwzdx, p.dwxdy, p.dwzdy, a.lod_w, a.lod_h);     // (RasterArgs::lod_w: the division by the world size folded in)
    const float lc = vr_clampf(lod, 0.0f, max_level);            // == the sampler's clamp (vr_lod_split): lod is never NaN
    const float lf = floorf(lc), frac = lc - lf;
    const int l16 = (int)(lf * 16.0f);                           // the level as a byte offset into the LDS level table
    const float offset = 0.1f;                                                              // :59
    const float ua = u + offset, ub = u + (-offset), va = v + offset, vb = v + (-offset);
    float hgt[4], col[3];
    {
        const FastTaps t0 = issue_level_fast(rq, rc, s_lv, l16, ua, ub, va, vb, u, v);
        VR_PROF_MARK(10);
        VR_PROF_DRAIN;
        VR_PROF_MARK(11);
        filter_level_fast(t0, hgt, col);
        VR_PROF_MARK(12);
    }
    // Wave-uniform branch (ballot): where every pixel of the wave is magnified (LOD 0 - the near half of an 8K frame)
    // the whole second level is skipped; a per-lane condition gets if-converted and every pixel pays for both levels.
    if (kExpNoLevel1 ? a.w < 0 : __any(frac > 0.0f)) {
        const int l16b = (int)(fminf(lf + 1.0f, max_level) * 16.0f);
        // The coordinates pass through an empty asm so that nothing of this level can be hoisted above the branch
        float xa = ua, xb = ub, ya = va, yb = vb, x0 = u, y0 = v;
        asm volatile("" : "+v"(xa), "+v"(xb), "+v"(ya), "+v"(yb), "+v"(x0), "+v"(y0));
        float g[4], cb[3];
        const FastTaps t1 = issue_level_fast(rq, rc, s_lv, l16b, xa, xb, ya, yb, x0, y0);
        filter_level_fast(t1, g, cb);
        // fma(b - a, f, a): with f == 0 this is a exactly (b - a is finite), as the oracle's "f > 0" branch leaves it
#pragma unroll
        for (int k = 0; k < 4; k++) hgt[k] = __builtin_fmaf(g[k] - hgt[k], frac, hgt[k]);
#pragma unroll
        for (int k = 0; k < 3; k++) col[k] = __builtin_fmaf(cb[k] - col[k], frac, col[k]);
    }
    VR_PROF_MARK(13);
    const float hDx = hgt[0] - hgt[1], hDy = hgt[2] - hgt[3];                                // :60-61
    float nx = -hDx, ny = 2.0f * offset, nz = -hDy;                                          // :63
    const float inv = vr_rcp_exact(vr_sqrt_exact(__builtin_fmaf(nz, nz, __builtin_fmaf(nx, nx, ny * ny))));
    nx *= inv; ny *= inv; nz *= inv;
    if (kExpNoEncode) diffuse = (__float_as_uint(col[0]) >> 20) | ((__float_as_uint(col[1]) >> 20) << 8) | ((__float_as_uint(col[2]) >> 20) << 16) | 0xff000000u;
    else
    diffuse = srgb_encode_nonneg(col[0], thr, enc) | (srgb_encode_nonneg(col[1], thr, enc) << 8) | (srgb_encode_nonneg(col[2], thr, enc) << 16)
            | 0xff000000u;                                                                  // :68, :73-75
    n01 = snorm16_unit<false>(nx) | (snorm16_unit<true>(ny) << 16);                         // :78
    n23 = snorm16_unit<false>(nz) | (32767u << 16);                                         // :79 roughness = 1
    VR_PROF_MARK(14);
}

// Low half of a visibility-buffer word: ~(key + 1), so a later bin entry (larger key) gives a SMALLER word and
// wins ties at equal depth (ds_min), and no entry - not even key 0, the first triangle of the first node -
// collides with 0xffffffff, the "nothing drawn" value.
__device__ __forceinline__ uint32_t order_of(uint32_t key) { return ~(key + 1u); }
__device__ __forceinline__ uint32_t key_of(uint32_t order) { return ~order - 1u; }

constexpr int kSmallArea = 16;     // triangles whose tile-clipped bbox has <= 16 pixels are rasterised by one lane
constexpr int kDenseWave = 32, kSmallAreaDense = 128;   // ... <= 128 pixels when at least half of the wave's lanes hold an entry
#ifndef VR_ROW_MIN
#define VR_ROW_MIN 8
#endif
constexpr int kRowMin = VR_ROW_MIN;  // the row hand-out needs this many eligible triangles in a wave (its scan + fetches are a fixed cost)
#ifndef VR_SPARSE_MAX
#define VR_SPARSE_MAX 32
#endif
constexpr int kSparseMax = VR_SPARSE_MAX;   // bins up to this long are walked entry by entry with scalar loads (k_raster's coverage) and keep their planes in LDS
static_assert(kSparseMax <= (1 << kTeSlotBits), "an entry's slot must fit the visibility word's low bits");
constexpr int kSweepW = 8;          // cooperative sweep block: 8 x 8 pixels (4x16, 16x4, 32x2, 64x1 measured 2-20 % slower)

// The depth plane of a triangle as the coverage sweeps carry it: coefficients + the tile's origin relative to the anchor
// (offx = ox - ax, offy = oy - ay: the plane's dx of tile-local pixel lx is lx + offx, an exact small integer in float).
struct ZPlane { float z0, zx, zy; int offx, offy; };

// One covered sample: depth from the triangle's plane at the pixel centre (oracle: fragment()), depth clip, LessOrEqual
// in draw order == 64-bit minimum of (depth bits, ~order).  fx, fy = the pixel's offset from the plane's anchor.
template <int TILE>
__device__ __forceinline__ void cover_pixel(unsigned long long* __restrict__ vis, int lx, int ly, float fx, float fy, const ZPlane& zp, uint32_t ord)
{
    float z = plane_at(zp.z0, zp.zx, zp.zy, fx, fy);
    if (!(z >= 0.0f && z <= 1.0f)) return;                       // depth clip
    z = z + 0.0f;                                                // canonical +0
    atomicMin(&vis[ly * TILE + lx], ((unsigned long long)__float_as_uint(z) << 32) | ord);
}

// Tile-relative edge functions: E_i(lx, ly) = e_i + sx_i*lx + sy_i*ly for the pixel (lx, ly)
// of the tile (centre sampled), bias_i = 0 on top-left edges else 1 (inside <=> E_i - bias_i >= 0).
// When every value over the tile fits in 32 bits the sweep runs in int32, else in int64.
template <typename T, int TILE>
__device__ __forceinline__ void sweep_small(unsigned long long* __restrict__ vis, T e0, T e1, T e2, T sx0, T sy0, T sx1, T sy1, T sx2, T sy2,
                                            int b0, int b1, int b2, int x0, int y0, int x1, int y1, const ZPlane& zp, uint32_t ord)
{
    for (int y = y0; y <= y1; y++) {
        T r0 = e0 + sy0 * (T)y - (T)b0 + sx0 * (T)x0, r1 = e1 + sy1 * (T)y - (T)b1 + sx1 * (T)x0, r2 = e2 + sy2 * (T)y - (T)b2 + sx2 * (T)x0;
        const float fy = (float)(y + zp.offy);
        float fx = (float)(x0 + zp.offx);
        for (int x = x0; x <= x1; x++) {
            if ((r0 | r1 | r2) >= 0) cover_pixel<TILE>(vis, x, y, fx, fy, zp, ord);
            r0 += sx0; r1 += sx1; r2 += sx2; fx += 1.0f;
        }
    }
}

// All 64 lanes sweep one (wave-uniform) triangle in 8x8 pixel blocks; blocks that lie
// completely outside an edge are skipped with scalar arithmetic only.
template <typename T, int TILE>
__device__ __forceinline__ void sweep_big(unsigned long long* __restrict__ vis, int lane, T e0, T e1, T e2, T sx0, T sy0, T sx1, T sy1, T sx2, T sy2,
                                          int b0, int b1, int b2, int x0, int y0, int x1, int y1, const ZPlane& zp, uint32_t ord)
{
    constexpr int BW = kSweepW, BH = 64 / kSweepW;                                                              // block of 64 pixels, one per lane
    const int lx = lane & (BW - 1), ly = lane / BW;
    const T l0 = sx0 * (T)lx + sy0 * (T)ly, l1 = sx1 * (T)lx + sy1 * (T)ly, l2 = sx2 * (T)lx + sy2 * (T)ly;    // per-lane offsets
    const T m0 = (sx0 > 0 ? sx0 * (BW - 1) : (T)0) + (sy0 > 0 ? sy0 * (BH - 1) : (T)0);                        // max offset inside a block
    const T m1 = (sx1 > 0 ? sx1 * (BW - 1) : (T)0) + (sy1 > 0 ? sy1 * (BH - 1) : (T)0);
    const T m2 = (sx2 > 0 ? sx2 * (BW - 1) : (T)0) + (sy2 > 0 ? sy2 * (BH - 1) : (T)0);
    const int lox = lx + zp.offx, loy = ly + zp.offy;
    for (int yb = y0 & ~(BH - 1); yb <= y1; yb += BH) {
        const T r0 = e0 - (T)b0 + sy0 * (T)yb, r1 = e1 - (T)b1 + sy1 * (T)yb, r2 = e2 - (T)b2 + sy2 * (T)yb;
        for (int xb = x0 & ~(BW - 1); xb <= x1; xb += BW) {
            const T o0 = r0 + sx0 * (T)xb, o1 = r1 + sx1 * (T)xb, o2 = r2 + sx2 * (T)xb;                        // block origin (uniform)
            if (((o0 + m0) | (o1 + m1) | (o2 + m2)) < 0) continue;                                              // block outside an edge
            const T v0 = o0 + l0, v1 = o1 + l1, v2 = o2 + l2;
            const int x = xb + lx, y = yb + ly;
            if ((v0 | v1 | v2) < 0 || x < x0 || x > x1 || y < y0 || y > y1) continue;
            cover_pixel<TILE>(vis, x, y, (float)(xb + lox), (float)(yb + loy), zp, ord);
        }
    }
}

// A long triangle (one side of its box >= kWalkMin pixels - at 8K most survivors are slivers of ~100 x 7 pixels): the lanes
// take the positions along the box's LONG axis and walk the short one, one row (two rows of a 32-pixel tile) of the tile
// per step.  ~22 instructions per step with every lane of the long side busy, against ~28 per 8x8 block whose lanes a
// sliver mostly misses.
#ifndef VR_WALK_MIN
#define VR_WALK_MIN 28
#endif
constexpr int kWalkMin = VR_WALK_MIN;
template <int TILE>
__device__ __forceinline__ void sweep_walk(unsigned long long* __restrict__ vis, int lane, int32_t e0, int32_t e1, int32_t e2, int32_t sx0, int32_t sy0,
                                           int32_t sx1, int32_t sy1, int32_t sx2, int32_t sy2, int b0, int b1, int b2, int x0, int y0, int x1, int y1,
                                           const ZPlane& zp, uint32_t ord)
{
    constexpr int PER = 64 / TILE;                               // short-axis positions covered per step: 1, or 2 for 32-pixel tiles
    const bool horiz = (x1 - x0) >= (y1 - y0);                   // (wave-uniform)
    const int la = lane & (TILE - 1), lb = lane / TILE;          // place on the long axis; which of the PER short-axis rows
    const int l0 = horiz ? x0 : y0, l1 = horiz ? x1 : y1, s0 = (horiz ? y0 : x0) + lb, s1 = horiz ? y1 : x1;
    if (la < l0 || la > l1) return;
    const int32_t a0 = horiz ? sx0 : sy0, a1 = horiz ? sx1 : sy1, a2 = horiz ? sx2 : sy2;     // step along the long axis
    const int32_t c0 = horiz ? sy0 : sx0, c1 = horiz ? sy1 : sx1, c2 = horiz ? sy2 : sx2;     // step along the short axis
    int32_t v0 = (e0 - b0) + a0 * la + c0 * s0, v1 = (e1 - b1) + a1 * la + c1 * s0, v2 = (e2 - b2) + a2 * la + c2 * s0;
    const float fla = (float)(la + (horiz ? zp.offx : zp.offy));             // this lane's offset from the anchor along the long axis
    float fsp = (float)(s0 + (horiz ? zp.offy : zp.offx));                   // ... and along the short one (exact small integers)
    for (int sp = s0; sp <= s1; sp += PER) {
        if ((v0 | v1 | v2) >= 0) cover_pixel<TILE>(vis, horiz ? la : sp, horiz ? sp : la, horiz ? fla : fsp, horiz ? fsp : fla, zp, ord);
        v0 += c0 * PER; v1 += c1 * PER; v2 += c2 * PER; fsp += (float)PER;
    }
}

__device__ __forceinline__ int64_t floor_div64(int64_t num, int64_t den)
{
    if (den < 0) { num = -num; den = -den; }
    int64_t q = num / den;
    if (num % den < 0) q--;
    return q;
}

// Wireframe: aliased line a-b, one pixel per column (x-major) or row (y-major) whose centre lies in
// [min, max) of the major axis: the pixel that contains the exact line point there.  The pixel is a
// sample of the triangle's plane at its centre (depth, attributes), so the resolve below is shared
// with fill mode.  Debug mode: one lane per triangle, no wave cooperation.
template <int TILE>
__device__ __noinline__ void wire_edge(unsigned long long* __restrict__ vis, int32_t aX, int32_t aY, int32_t bX, int32_t bY, const ZPlane& zp,
                                        int ox, int oy, int bx0, int by0, int bx1, int by1, uint32_t ord)
{
    const int64_t dX = (int64_t)bX - aX, dY = (int64_t)bY - aY;
    if (dX == 0 && dY == 0) return;
    const bool xmajor = llabs(dX) >= llabs(dY);
    const int32_t ca = xmajor ? aX : aY, cb = xmajor ? bX : bY;
    const int32_t lo = min(ca, cb), hi = max(ca, cb);
    const int p0 = max((lo - 128 + 255) >> 8, xmajor ? bx0 : by0), p1 = min(((hi - 128 + 255) >> 8) - 1, xmajor ? bx1 : by1);
    for (int p = p0; p <= p1; p++) {
        const int64_t P = (int64_t)p * 256 + 128;
        int px, py;
        if (xmajor) {
            const int64_t q = floor_div64((int64_t)aY * dX + (P - aX) * dY, dX * 256);
            if (q < by0 || q > by1) continue;
            px = p; py = (int)q;
        } else {
            const int64_t q = floor_div64((int64_t)aX * dY + (P - aY) * dX, dY * 256);
            if (q < bx0 || q > bx1) continue;
            px = (int)q; py = p;
        }
        cover_pixel<TILE>(vis, px - ox, py - oy, (float)(px - ox + zp.offx), (float)(py - oy + zp.offy), zp, ord);
    }
}

// Workgroup size of the tile pass.  The visibility buffer (32 KB for a 64-pixel tile) admits four workgroups per CU;
// how many waves that is per SIMD follows from the workgroup's size (256 threads: 4, 320: 5, 384: 6) if the kernel's
// registers allow as many (<= 128 / 96 / 80).
#ifndef VR_RASTER_THREADS
#define VR_RASTER_THREADS 256
#endif
constexpr int kRT = VR_RASTER_THREADS, kRW = kRT / 64;
#ifndef VR_RASTER_WAVES_PER_EU
#define VR_RASTER_WAVES_PER_EU kRW
#endif

// Variants of the tile pass's resolve (the coverage phase is the same in all of them):
//   RM_FAST    the reference's case, with everything the host knows about it compiled in: heightmap and albedo of one size,
//              a power-of-two world size, G-buffer planes within 4 GB of each other (one buffer resource), shaded;
//   RM_DEPTH   depth only (the shadow map's pass): no record fetch, no texel fetch, one store per pixel;
//   RM_GENERIC everything else, decided at run time.
enum { RM_GENERIC = 0, RM_FAST = 1, RM_DEPTH = 2 };
// RANGES (fast variant only): the depth range of every 32x32 light tile is left at `ranges` for the tiled lighting pass.
// NOEMI (fast variant only): the emissive plane is known to hold zeros already (vr_gbuffer::emissive_zero) and is not rewritten -
// main_ps's o_channel3 = 0 (terrain_ps.hlsl:80) changes nothing there; 8 of the 28 bytes a pixel sends through the CU's store path.
// LIT (fast variant only; vr_terrain_render_lit): the fused pass of SURVEY 7 step 6 - the resolve encodes the pixel to the
// G-buffer's formats in registers, decodes and shades it with the lighting pass's own arithmetic (shade_pixel, vr_deferred_dev.h)
// and stores depth + HdrColor only: 12 bytes per pixel leave the pass instead of 28, and the lighting pass's 36 are not moved at
// all.  Same bits as vr_terrain_render + vr_deferred_light.  Opt-in; the unfused pair stays the default and the graded path.
struct LitArgs {
    DeferredArgs da;                   // the lighting pass's constants
    const float* lut_g;                // sRGB8 -> linear (256 floats)
    uint2* hdr;                        // HdrColor, row-major RGBA16F - or, with a partition, this rank's packed RGB16F tiles
    const int32_t* tile_slot;          // partition: owner tile -> rank * max_owned + local index (NULL: whole frame, row-major)
    int slot_base, owner_tiles_x;
};
struct LitNone { int unused; };
template <bool WIRE, int TILE, int MODE, bool RANGES = false, bool NOEMI = false, bool LIT = false>
// A 32-pixel tile's workgroup needs 11 KB of LDS: registers, not LDS, decide how many fit a CU.  Asked for six waves per SIMD
// the compiler fits every 32-pixel variant into 80 VGPRs without a spill (84-90 otherwise: five waves): 5120x2880 frame
// 0.288 -> 0.279 ms, 4K 0.203 -> 0.2015, the rank of an 8-way split 0.128 -> 0.126 (profiles/r03_tile32_waves.txt).
#ifndef VR_RASTER_WAVES_32
#define VR_RASTER_WAVES_32 6
#endif
__global__ __launch_bounds__(kRT, (LIT ? 4 : TILE == 32 ? VR_RASTER_WAVES_32 : VR_RASTER_WAVES_PER_EU)) void k_raster(RasterArgs a, DevTex hm, DevTex al, const DevVert* __restrict__ verts,
                                                 const HardTriRec* __restrict__ hard_tris, const uint32_t* __restrict__ hard_first,
                                                 const uint4* __restrict__ recs, uint32_t rec_hard_base,
                                                 const uint32_t* __restrict__ tile_cursor, const uint32_t* __restrict__ tile_offset,
                                                 const TileEntry* __restrict__ entries, const int32_t* __restrict__ tile_list,
                                                 const uint32_t* __restrict__ tile_classes,
                                                 float* __restrict__ g_depth, uint32_t* __restrict__ g_diff, uint32_t* __restrict__ g_spec,
                                                 uint2* __restrict__ g_nrm, uint2* __restrict__ g_emi,
                                                 const float* __restrict__ thr_g,
                                                 const uint8_t* __restrict__ enc_g, uint32_t spec_const, uint2* __restrict__ ranges,
                                                 uint8_t* __restrict__ region,
                                                 typename std::conditional<LIT, LitArgs, LitNone>::type lit)
{
    // Region states (plane-state tracking, vr_gbuffer::d_region; NULL: off): one byte per wave of a 32-pixel tile = its 8 rows x 32
    // pixels.  kRegionClear: every pixel of the region holds the clear values in all planes; kRegionSpec: every pixel holds
    // spec_const in the specular plane (main_ps writes one constant there, terrain_ps.hlsl:76).  A region that is all sky and
    // known clear is not written at all, a region that is all terrain and known constant keeps its specular plane: the bytes in
    // memory are the same either way.
    constexpr bool TRACK = MODE == RM_FAST && TILE == 32 && !LIT && !WIRE;
    static_assert(!LIT || (MODE == RM_FAST && !RANGES), "the fused variant is the fast variant");
    static_assert((!RANGES && !NOEMI) || MODE == RM_FAST, "depth ranges and the emissive skip come with the fast variant");
    __shared__ unsigned long long vis[TILE * TILE];
    __shared__ __attribute__((aligned(4))) uint8_t enc[(kEncTabSize + 3) / 4 * 4];
    __shared__ float thr[kThrTabSize];
    __shared__ uint32_t s_qoff[kMaxLevels], s_aoff[kMaxLevels];
    __shared__ uint4 s_lv[kMaxLevels];                 // fast variant: DevTex::fast_lv (the same for both textures)
    // The interpolation planes of a sparse bin's triangles (record groups 5-7), 48 B per entry: the resolve reads a pixel's winner
    // from here by the slot in its visibility word - three LDS reads instead of three fetches per pixel and lane (tile pass alone
    // -2.5 %; profiles/r04_tile_pass_experiments.txt).  Ninety-six lanes fetch them while the sweeps run.
    __shared__ uint4 s_rec[MODE != RM_DEPTH ? 3 * (1 << kTeSlotBits) : 1];
    __shared__ float s_lut[LIT ? 256 : 1];             // fused variant: the lighting pass's sRGB8 -> linear table
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    VR_PROF_BEGIN;
    const int tile = tile_list ? tile_of_block((int)blockIdx.x, tile_list, tile_classes, a.rtx * a.rty) : (int)blockIdx.x;
    if (tile < 0 || tile >= a.rtx * a.rty) return;              // never index the bins or the targets with a foreign tile id
    // Everything the workgroup needs from memory before its first sweep is requested HERE, at once, and used after the
    // visibility buffer's set-up: the small tables and the bin's bounds.  Written in program order (table, wait, LDS store,
    // next table, ..., barrier, bounds, entries) the prologue was eight dependent round trips per workgroup,
    // a fifth of a 32-pixel tile's time (4K: init + record fetch = 21 % of the wave cycles).
    uint4 lv_r = make_uint4(0, 0, 0, 0);
    if (MODE == RM_FAST && tid < kMaxLevels) lv_r = hm.fast_lv[tid];
    uint32_t qoff_r = 0, aoff_r = 0;
    if (MODE != RM_FAST && tid < kMaxLevels) { qoff_r = hm.qoff[tid]; aoff_r = al.off[tid]; }
    constexpr int kEncWords = (kEncTabSize + 3) / 4, kEncPer = (kEncWords + kRT - 1) / kRT;
    uint32_t enc_r[kEncPer];
#pragma unroll
    for (int q = 0; q < kEncPer; q++) enc_r[q] = (!kExpNoTables && tid + q * kRT < kEncWords) ? reinterpret_cast<const uint32_t*>(enc_g)[tid + q * kRT] : 0u;
    const float thr_r = (!kExpNoTables && tid < 256) ? thr_g[tid] : 0.0f;
    float lut_r = 0.0f;
    if constexpr (LIT) lut_r = lit.lut_g[tid & 255];
    const uint32_t off = tile_offset[tile], n_all = tile_cursor[tile] - off;     // bin = entries[off .. off + n_all)
    const uint32_t n = min(n_all, off < a.bin_capacity ? a.bin_capacity - off : 0u);   // (an overflowing frame drops the entries beyond the capacity: VR_ERR_OVERFLOW)
    const TileEntry* __restrict__ bin = entries + off;
    if constexpr (TRACK) {
        // nothing to draw and the whole tile known to hold the clear values already: done (workgroup-uniform)
        if (region != nullptr && a.assume_cleared && n_all == 0u
            && reinterpret_cast<const uint32_t*>(region)[tile] == kRegionClear * 0x01010101u) return;
    }
    const bool lds_recs = !WIRE && MODE != RM_DEPTH && n <= (uint32_t)kSparseMax;      // (workgroup-uniform)
    uint4 rec_r = make_uint4(0, 0, 0, 0);                         // requested now, stored behind the sweeps: its latency is theirs
    if (lds_recs && (uint32_t)tid < n * 3u) {
        const uint32_t ord_j = bin[(uint32_t)tid / 3u].order;     // entry -> key -> record (two dependent fetches, under the buffer's set-up and the sweeps)
        rec_r = recs[rec_index(key_of(ord_j), hard_first, rec_hard_base) * kRecGroups + 5u + (uint32_t)tid % 3u];
    }
    // consecutive bin entries go to different waves so that a short list still uses all of them
    const uint32_t idx0 = (uint32_t)(lane * kRW + wave);
    const int tyi = tile / a.rtx, txi = tile - tyi * a.rtx;
    const int ox = txi * TILE, oy = tyi * TILE;
    // visibility buffer: existing depth (or the clear value) with the "nothing drawn" key
    if (a.assume_cleared && ox + TILE <= a.w && oy + TILE <= a.h) {       // interior tile of a cleared target: one constant
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 cv = { 0x3f800000ffffffffull, 0x3f800000ffffffffull };
        for (int i = tid; i < TILE * TILE / 2; i += kRT) reinterpret_cast<u64x2*>(vis)[i] = cv;
    } else
    for (int i = tid; i < TILE * TILE; i += kRT) {
        const int lx = i & (TILE - 1), ly = i / TILE;
        const int gx = ox + lx, gy = oy + ly;
        unsigned long long key = 0ull;                         // outside the target: nothing passes
        if (gx < a.w && gy < a.h) {
            const uint32_t dbits = a.assume_cleared ? 0x3f800000u : __float_as_uint(g_depth[(size_t)gy * a.w + gx]);
            key = ((unsigned long long)dbits << 32) | 0xffffffffull;
        }
        vis[i] = key;
    }
    if (MODE == RM_FAST && tid < kMaxLevels) s_lv[tid] = lv_r;
    if (MODE != RM_FAST && tid < kMaxLevels) { s_qoff[tid] = qoff_r; s_aoff[tid] = aoff_r; }
#pragma unroll
    for (int q = 0; q < kEncPer; q++) if (tid + q * kRT < kEncWords) reinterpret_cast<uint32_t*>(enc)[tid + q * kRT] = enc_r[q];
    if (tid < 256) thr[tid] = thr_r;
    if (tid == 0) thr[256] = __uint_as_float(0x7fc00000u);   // NaN: no x is >= it, not even +inf
    if (LIT && tid < 256) s_lut[tid] = lut_r;
    __syncthreads();
    VR_PROF_MARK(0);

    const int32_t PX0 = ox * 256 + 128, PY0 = oy * 256 + 128;   // centre of the tile's pixel (0,0)
    // ---- coverage.  A bin entry is the triangle already set up against THIS tile (TileEntry, written by k_fill): edge values at
    // the tile's first pixel, steps, box, depth plane, draw-order word.  Until round 4 every workgroup derived all that per lane
    // from the triangle's record - five 16-byte fetches behind the key's, ~250 vector instructions per wave whether the wave held
    // two entries or sixty - and broadcast each swept triangle with ~25 v_readlane.
    // A wide triangle (edge values beyond 31 bits over the tile: kTeFits32 clear; rare) is redone in 64 bits from its record.
    auto sweep_wide = [&](uint32_t ord, uint32_t key_ord, uint32_t bx, uint32_t flags, const ZPlane& zz) {          // (wave-uniform arguments)
        const uint32_t key = key_of(key_ord);
        const uint4* __restrict__ rp = recs + rec_index(key, hard_first, rec_hard_base) * kRecGroups;
        const uint4 g0 = rp[0], g1 = rp[1], g2 = rp[2];
        const int64_t e0 = edge_eval((int32_t)g0.x, (int32_t)g0.y, rec_c(g0), PX0, PY0), e1 = edge_eval((int32_t)g1.x, (int32_t)g1.y, rec_c(g1), PX0, PY0),
                      e2 = edge_eval((int32_t)g2.x, (int32_t)g2.y, rec_c(g2), PX0, PY0);
        sweep_big<int64_t, TILE>(vis, lane, e0, e1, e2, (int64_t)(int32_t)g0.x * 256, (int64_t)(int32_t)g0.y * 256, (int64_t)(int32_t)g1.x * 256,
                                 (int64_t)(int32_t)g1.y * 256, (int64_t)(int32_t)g2.x * 256, (int64_t)(int32_t)g2.y * 256, (int)(flags & 1u), (int)((flags >> 1) & 1u),
                                 (int)((flags >> 2) & 1u), (int)(bx & 255u), (int)((bx >> 8) & 255u), (int)((bx >> 16) & 255u), (int)(bx >> 24), zz, ord);
    };
    // One (wave-uniform) entry swept by all 64 lanes: along its long axis when that is most of a tile, else in 8x8 blocks.
    auto sweep_uniform = [&](int32_t e0, int32_t e1, int32_t e2, int32_t sx0, int32_t sy0, int32_t sx1, int32_t sy1, int32_t sx2, int32_t sy2,
                             uint32_t bx, float z0, float zx, float zy, uint32_t bo, uint32_t ord, uint32_t flags, uint32_t key_ord) {
        ZPlane bz;
        bz.z0 = z0; bz.zx = zx; bz.zy = zy;
        bz.offx = (int)(int16_t)(bo & 0xffffu); bz.offy = (int)(int16_t)(bo >> 16);
        if (!(flags & kTeFits32)) { sweep_wide(ord, key_ord, bx, flags, bz); return; }
        const int b0 = flags & 1u, b1 = (flags >> 1) & 1u, b2 = (flags >> 2) & 1u;
        const int qx0 = bx & 255u, qy0 = (bx >> 8) & 255u, qx1 = (bx >> 16) & 255u, qy1 = bx >> 24;
        if (max(qx1 - qx0, qy1 - qy0) + 1 >= min(kWalkMin, TILE - 4))
            sweep_walk<TILE>(vis, lane, e0, e1, e2, sx0, sy0, sx1, sy1, sx2, sy2, b0, b1, b2, qx0, qy0, qx1, qy1, bz, ord);
        else
            sweep_big<int32_t, TILE>(vis, lane, e0, e1, e2, sx0, sy0, sx1, sy1, sx2, sy2, b0, b1, b2, qx0, qy0, qx1, qy1, bz, ord);
    };
    if (!WIRE && n <= (uint32_t)kSparseMax) {
        // Sparse bin (the rule at 8K: ~6 entries per 32-pixel tile): every wave walks its share of the entries, and because the entry
        // index is wave-uniform the 64 bytes arrive through the scalar cache straight in SGPRs - the next entry requested before
        // the current one is swept.
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
        const uint4* __restrict__ q = reinterpret_cast<const uint4*>(bin);
        uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0, n2 = n0, n3 = n0;
        if (w0 < n) { n0 = q[w0 * kTeGroups + 0]; n1 = q[w0 * kTeGroups + 1]; n2 = q[w0 * kTeGroups + 2]; n3 = q[w0 * kTeGroups + 3]; }
        for (uint32_t j = w0; j < n; j += kRW) {
            const uint4 t0 = n0, t1 = n1, t2 = n2, t3 = n3;
            const uint32_t jn = j + kRW;
            if (jn < n) { n0 = q[jn * kTeGroups + 0]; n1 = q[jn * kTeGroups + 1]; n2 = q[jn * kTeGroups + 2]; n3 = q[jn * kTeGroups + 3]; }
            if (!(t3.w & kTeValid)) continue;
            // the visibility word of a sparse bin: draw order (27 bits, later = smaller, as ever) above the entry's slot
            sweep_uniform((int32_t)t0.x, (int32_t)t0.y, (int32_t)t0.z, (int32_t)t0.w, (int32_t)t1.x, (int32_t)t1.y, (int32_t)t1.z, (int32_t)t1.w, (int32_t)t2.x,
                          t2.y, __uint_as_float(t2.z), __uint_as_float(t2.w), __uint_as_float(t3.x), t3.y, MODE != RM_DEPTH ? (t3.z << kTeSlotBits) | j : t3.z, t3.w, t3.z);
        }
        VR_PROF_MARK(4);
    } else
    for (uint32_t base = 0; base < n; base += kRT) {
        // Dense bin (low resolutions, distant terrain): one lane per entry; tiny triangles are rasterised by their lane, small ones
        // handed out row by row, the rest broadcast and swept by the wave.
        const uint32_t idx = base + idx0;
        bool valid = idx < n;
        uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0, t2 = t0, t3 = t0;
        if (valid) { const uint4* __restrict__ q = reinterpret_cast<const uint4*>(bin + idx); t0 = q[0]; t1 = q[1]; t2 = q[2]; t3 = q[3]; }
        const int32_t e0 = (int32_t)t0.x, e1 = (int32_t)t0.y, e2 = (int32_t)t0.z;
        const int32_t sx0 = (int32_t)t0.w, sy0 = (int32_t)t1.x, sx1 = (int32_t)t1.y, sy1 = (int32_t)t1.z, sx2 = (int32_t)t1.w, sy2 = (int32_t)t2.x;
        const uint32_t box = t2.y, offs = t3.y, order = t3.z, flags = t3.w;
        const int bias0 = (int)(flags & 1u), bias1 = (int)((flags >> 1) & 1u), bias2 = (int)((flags >> 2) & 1u);
        const bool fits32 = (flags & kTeFits32) != 0u, is_small = (flags & kTeSmall) != 0u;
        ZPlane zp;
        zp.z0 = __uint_as_float(t2.z); zp.zx = __uint_as_float(t2.w); zp.zy = __uint_as_float(t3.x);
        zp.offx = (int)(int16_t)(offs & 0xffffu); zp.offy = (int)(int16_t)(offs >> 16);
        valid = valid && (flags & kTeValid) != 0u;
        const int x0 = (int)(box & 255u), y0 = (int)((box >> 8) & 255u), x1 = (int)((box >> 16) & 255u), y1 = (int)(box >> 24);      // tile-local, inclusive
        if (WIRE) {
            if (valid) {
                uint32_t i0, i1, i2;
                entry_vertices(key_of(order), hard_tris, hard_first, i0, i1, i2);
                const ScreenVert s0 = load_sv(verts, i0), s1 = load_sv(verts, i1), s2 = load_sv(verts, i2);
                const int tx0 = x0 + ox, ty0 = y0 + oy, tx1 = x1 + ox, ty1 = y1 + oy;
                wire_edge<TILE>(vis, s0.X, s0.Y, s1.X, s1.Y, zp, ox, oy, tx0, ty0, tx1, ty1, order);
                wire_edge<TILE>(vis, s1.X, s1.Y, s2.X, s2.Y, zp, ox, oy, tx0, ty0, tx1, ty1, order);
                wire_edge<TILE>(vis, s2.X, s2.Y, s0.X, s0.Y, zp, ox, oy, tx0, ty0, tx1, ty1, order);
            }
            continue;                   // covered; no fill sweeps
        }
        VR_PROF_MARK(1);
        const int bw = valid ? x1 - x0 + 1 : 0, bh = valid ? y1 - y0 + 1 : 0;
        // One lane per triangle pays when the wave's lanes are mostly busy or the box is tiny; everything else is handed out row by
        // row (small triangles) or swept by the whole wave.
        const int n_wave = __popcll(__ballot(valid));
        const bool tiny = valid && fits32 && (bw * bh <= (n_wave >= kDenseWave ? kSmallAreaDense : kSmallArea));
        if (tiny) sweep_small<int32_t, TILE>(vis, e0, e1, e2, sx0, sy0, sx1, sy1, sx2, sy2, bias0, bias1, bias2, x0, y0, x1, y1, zp, order);
        VR_PROF_MARK(2);
        bool rowp = false;
        // ---- small triangles, row by row (only when the wave holds enough of them to repay the hand-out; a few are
        // cheaper through the cooperative sweep below): the wave's rows are numbered through (prefix sum of the box heights) and
        // handed to the lanes 64 at a time; a lane fetches its row's triangle from the owning lane (ds_bpermute) and walks
        // the row's pixels.  Lane utilisation no longer depends on how many triangles the bin holds or how they are shaped.
        const bool row_ok = valid && !tiny && is_small;
        if (__popcll(__ballot(row_ok)) >= kRowMin) {
            rowp = row_ok;
            const int rows = rowp ? bh : 0;
            int incl = rows;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
            const int R = __builtin_amdgcn_readlane(incl, 63);
            if (R > 0) {        // (uniform: nothing below runs for a wave without row work)
                // biased edge values at the box's first pixel; E_i - bias_i >= 0 <=> inside
                const int32_t f0 = (e0 + __mul24(sy0, y0)) + (__mul24(sx0, x0) - bias0);
                const int32_t f1 = (e1 + __mul24(sy1, y0)) + (__mul24(sx1, x0) - bias1);
                const int32_t f2 = (e2 + __mul24(sy2, y0)) + (__mul24(sx2, x0) - bias2);
                const int start = incl - rows;
                const uint32_t xb = (uint32_t)x0 | ((uint32_t)x1 << 8) | ((uint32_t)y0 << 16);
                for (int cb = 0; cb < R; cb += 64) {
                    const int r = cb + lane;
                    int tl = 0;                                   // number of lanes whose rows end at or before r = the owning lane
#pragma unroll
                    for (int step = 32; step >= 1; step >>= 1) { const int v = __shfl(incl, tl + step - 1); if (v <= r) tl += step; }
                    const bool act = r < R;
                    tl = min(tl, 63);
                    const int j = r - __shfl(start, tl);          // row inside the triangle's box
                    const uint32_t qb = (uint32_t)__shfl((int)xb, tl);
                    const int32_t qx0 = __shfl(sx0, tl), qx1 = __shfl(sx1, tl), qx2 = __shfl(sx2, tl);
                    int32_t v0 = __shfl(f0, tl) + __mul24(__shfl(sy0, tl), j);
                    int32_t v1 = __shfl(f1, tl) + __mul24(__shfl(sy1, tl), j);
                    int32_t v2 = __shfl(f2, tl) + __mul24(__shfl(sy2, tl), j);
                    ZPlane qz;
                    qz.z0 = __shfl(zp.z0, tl); qz.zx = __shfl(zp.zx, tl); qz.zy = __shfl(zp.zy, tl);
                    const uint32_t qo = (uint32_t)__shfl((int)offs, tl);
                    qz.offx = (int)(int16_t)(qo & 0xffffu); qz.offy = (int)(int16_t)(qo >> 16);
                    const uint32_t qord = (uint32_t)__shfl((int)order, tl);
                    const int y = (int)((qb >> 16) & 255u) + j, xe = (int)((qb >> 8) & 255u);
                    int x = (int)(qb & 255u);
                    const float fy = (float)(y + qz.offy);
                    float fx = (float)(x + qz.offx);
                    while (__any(act && x <= xe)) {
                        if (act && x <= xe && (v0 | v1 | v2) >= 0) cover_pixel<TILE>(vis, x, y, fx, fy, qz, qord);
                        v0 += qx0; v1 += qx1; v2 += qx2; x++; fx += 1.0f;
                    }
                }
            }
        }
        VR_PROF_MARK(3);
        // everything else: broadcast one at a time (v_readlane -> SGPRs), all 64 lanes sweep its box
        unsigned long long big = __ballot(valid && !tiny && !rowp);
        VR_PROF_ADD(7, ((unsigned long long)__popcll(big) << 40) | ((unsigned long long)__popcll(__ballot(rowp)) << 20) | (unsigned long long)__popcll(__ballot(tiny)));
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
#define BC(v) __builtin_amdgcn_readlane((int)(v), src)
            sweep_uniform(BC(e0), BC(e1), BC(e2), BC(sx0), BC(sy0), BC(sx1), BC(sy1), BC(sx2), BC(sy2), (uint32_t)BC(box),
                          __int_as_float(BC(__float_as_int(zp.z0))), __int_as_float(BC(__float_as_int(zp.zx))), __int_as_float(BC(__float_as_int(zp.zy))),
                          (uint32_t)BC(offs), (uint32_t)BC(order), (uint32_t)BC(flags), (uint32_t)BC(order));
#undef BC
        }
        VR_PROF_MARK(4);
    }
    if (lds_recs && (uint32_t)tid < n * 3u) s_rec[tid] = rec_r;
    __syncthreads();
    VR_PROF_MARK(5);

    // ---- resolve: shade each pixel's winner once ------------------
    constexpr bool FAST = MODE == RM_FAST, DEPTH = MODE == RM_DEPTH, SAME = FAST;
    const bool depth_only = DEPTH || (!FAST && a.depth_only != 0);
    const __amdgpu_buffer_rsrc_t rq = FAST ? __builtin_amdgcn_make_buffer_rsrc((void*)hm.fast, (short)0, (int)hm.fast_bytes, 0x00020000)
                                           : __builtin_amdgcn_make_buffer_rsrc((void*)hm.quadf, (short)0, (int)(hm.quad_bytes * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = FAST ? __builtin_amdgcn_make_buffer_rsrc((void*)al.fast, (short)0, (int)al.fast_bytes, 0x00020000)
                                           : __builtin_amdgcn_make_buffer_rsrc((void*)al.rgbf, (short)0, (int)(al.chain_bytes * 4u), 0x00020000);
    // targets through one buffer resource when the planes lie within 4 GB of the depth plane (a vr_gbuffer of up to 16K x 8K;
    // the host launches the fast variant only then)
    const uint64_t gb_span = (uint64_t)(reinterpret_cast<const char*>(g_emi + (size_t)a.w * a.h) - reinterpret_cast<const char*>(g_depth));
    const uint64_t od = (uint64_t)(reinterpret_cast<const char*>(g_diff) - reinterpret_cast<const char*>(g_depth));
    const uint64_t os = (uint64_t)(reinterpret_cast<const char*>(g_spec) - reinterpret_cast<const char*>(g_depth));
    const uint64_t on = (uint64_t)(reinterpret_cast<const char*>(g_nrm) - reinterpret_cast<const char*>(g_depth));
    const uint64_t oe = (uint64_t)(reinterpret_cast<const char*>(g_emi) - reinterpret_cast<const char*>(g_depth));
    const bool gb_small = FAST || (gb_span < (1ull << 32) && od < gb_span && os < gb_span && on < gb_span && oe < gb_span);
    const int o_diff = (int)(uint32_t)od, o_spec = (int)(uint32_t)os, o_nrm = (int)(uint32_t)on, o_emi = (int)(uint32_t)oe;
    const __amdgpu_buffer_rsrc_t rgb = __builtin_amdgcn_make_buffer_rsrc((void*)g_depth, (short)0, (int)(uint32_t)gb_span, 0x00020000);
    // A lane owns a column of four pixels, the lanes of a wave are neighbouring columns: one texel fetch of the wave then
    // touches adjacent texels (8 lanes per 128-byte line where the terrain is minified) instead of every fourth one, as it
    // did when a lane owned four pixels of a row.
    // Every pixel leaves through its stores as soon as it is shaded (nothing is kept for the end of the column): a wave's
    // store covers 64 (32) neighbouring pixels of a row, 256 contiguous bytes per 4-byte plane.
    // Streaming (non-temporal) stores: the G-buffer is written once and read once by the lighting pass.  Kept out of the
    // caches it does not sit there as dirty lines that the lighting pass has to evict while it streams (8K: k_deferred
    // 226 -> 213 us), and - with the lighting pass streaming as well - the NEXT tile pass still finds its 180 MB of texel
    // tables in the Infinity Cache.  Round 2 made the choice by frame size (64-pixel tiles only: "smaller frames mostly fit
    // the cache"); measured again with this round's tile pass, streaming both passes wins or ties at every size: 4K frame
    // 0.223 -> 0.204 ms (tile pass 148 -> 123 us), 1440p 0.152 -> 0.148, 1080p and 720p unchanged (profiles/r03_nontemporal_sizes.txt).
    // The five planes of a vr_gbuffer are one allocation: one buffer resource, the plane as the scalar offset and one
    // 32-bit pixel offset per store instead of a 64-bit address per plane and pixel.
    // The hot path is kept free of taken branches: everything the fast variant knows is a template parameter, the implicit
    // LOD is computed with selects, and the two rare cases (a reciprocal outside the short sequence's range, a second mip
    // level) sit behind wave-uniform branches that fall through when they do not apply.
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        constexpr int aux = 2;                                // nt (every frame size: measured again in round 3, below)
#define ST1(ptr, v) __builtin_nontemporal_store((v), (ptr))
#define ST2(ptr, a_, b_) do { u2 v_ = { (a_), (b_) }; __builtin_nontemporal_store(v_, reinterpret_cast<u2*>(ptr)); } while (0)
    const bool whole = ox + TILE <= a.w && oy + TILE <= a.h;      // (workgroup-uniform) no pixel of this tile lies outside the target
    // The winner's planes (record groups 5..7) are fetched ONE PIXEL AHEAD, and UNCONDITIONALLY: pixel k + 1's record is
    // requested before pixel k is shaded, so its round trip (entry -> record, a gather: every lane its own triangle) runs
    // under pixel k's texel fetches instead of in front of pixel k + 1's.  Unconditionally, because a fetch that depends on
    // "is the next pixel covered" makes the hand-over a merge of two values (the new record, or the one held): the copy
    // that merge needs waits for EVERYTHING in flight (s_waitcnt vmcnt(0)) - including the five stores the previous pixel
    // has just issued, whose acknowledgements then sat on every pixel's critical path (no G-buffer stores: -19 % of the
    // pass; the same bytes into a 256-KB window: no change; profiles/r03_tile_pass_experiments.txt).  An uncovered pixel's
    // word addresses a record beyond the buffer resource's end and reads zeros.  With a plain hand-over a pixel waits
    // only for its own record (vmcnt counts in order: the stores and the next record's fetches are younger).
    // The pipeline runs across the wave's strips as well: the next strip's visibility words are read, and its first record
    // requested, under the current strip's last pixel.
    const uint32_t rec_bytes = (rec_hard_base + a.hard_cap * 4u) * (uint32_t)(kRecGroups * 16);      // < 2^32: at most 4096 instances
    const __amdgpu_buffer_rsrc_t rrec = __builtin_amdgcn_make_buffer_rsrc((void*)recs, (short)0, (int)rec_bytes, 0x00020000);
    typedef unsigned int u3 __attribute__((ext_vector_type(3)));
    struct Rec { u32x4 g5; u3 g6, g7; };
    auto fetch_rec = [&](uint32_t low) -> Rec {
        if (lds_recs) {                                          // (workgroup-uniform) a sparse bin: the winner's planes by its slot
            const char* e = reinterpret_cast<const char*>(s_rec) + (low & (uint32_t)((1 << kTeSlotBits) - 1)) * 48u;
            Rec r;
            r.g5 = *reinterpret_cast<const u32x4*>(e); r.g6 = *reinterpret_cast<const u3*>(e + 16); r.g7 = *reinterpret_cast<const u3*>(e + 32);
            return r;
        }
        const uint32_t key = key_of(low);
        uint32_t idx = key >> 4;
        const bool hard = (key & 1u) != 0u && low != 0xffffffffu;
        if (__builtin_expect(__any(hard), 0)) {                  // (wave-uniform; clipper output only)
            if (hard) idx = rec_hard_base + hard_first[idx] + ((key >> 1) & 7u);
        }
        const uint32_t off = idx << 7;                           // an uncovered pixel: 0x0fffffff << 7 = far beyond rec_bytes -> zeros
        Rec r;
        if (kExpNoRecord) { r.g5 = (u32x4){ 0x3f800000u, 0u, 0u, 0u }; r.g6 = (u3){ off, 0x3a800000u, 0u }; r.g7 = (u3){ 0x3f000000u, 0u, 0x3a800000u }; return r; }

        r.g5 = __builtin_amdgcn_raw_buffer_load_b128(rrec, off, 80, 0);
        r.g6 = __builtin_amdgcn_raw_buffer_load_b96(rrec, off, 96, 0);
        r.g7 = __builtin_amdgcn_raw_buffer_load_b96(rrec, off, 112, 0);
        return r;
    };
    constexpr int kStrips = TILE * TILE / 4;
    static_assert(kStrips >= kRT, "every thread owns at least one strip");
    unsigned long long nkeys[4];
    {
        const int lx = tid % TILE, ly0 = (tid / TILE) * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) nkeys[k] = vis[(ly0 + k) * TILE + lx];
    }
    Rec nrec;                                                     // the NEXT pixel's record
    if (!depth_only) nrec = fetch_rec((uint32_t)nkeys[0]);
    else { nrec.g5 = (u32x4){ 0u, 0u, 0u, 0u }; nrec.g6 = (u3){ 0u, 0u, 0u }; nrec.g7 = nrec.g6; }
    // RANGES: smallest / largest depth bits below 1.0 this lane has seen in the upper / lower 32 rows of the tile
    uint32_t rmin0 = 0x7f800000u, rmax0 = 0u, rmin1 = 0x7f800000u, rmax1 = 0u;
    bool skip_all = false, skip_spec = false;                     // (wave-uniform)
    if constexpr (TRACK) {
        if (region != nullptr) {
            const uint32_t ri = (uint32_t)tile * 4u + (uint32_t)__builtin_amdgcn_readfirstlane(wave);
            const uint32_t st = region[ri];
            // this lane's four pixels (those that lie on the target): all covered / none covered
            const int lx = tid % TILE, ly0 = (tid / TILE) * 4;
            bool c_all = true, c_none = true;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool on = whole || (ox + lx < a.w && oy + ly0 + k < a.h);
                const bool cov = (uint32_t)nkeys[k] != 0xffffffffu;
                c_all = c_all && (!on || cov); c_none = c_none && (!on || !cov);
            }
            const bool all_cov = __all(c_all), none_cov = __all(c_none);
            uint32_t nst;
            if (none_cov) { skip_all = a.assume_cleared ? st == kRegionClear : true; nst = a.assume_cleared ? kRegionClear : st; }
            else if (all_cov) { skip_spec = st == kRegionSpec; nst = kRegionSpec; }
            else nst = (!a.assume_cleared && st == kRegionSpec) ? kRegionSpec : 0u;
            if (nst != st && lane == 0) region[ri] = (uint8_t)nst;
        }
    }
    for (int g = skip_all ? kStrips : tid; g < kStrips; g += kRT) {
        const int lx = g % TILE, ly0 = (g / TILE) * 4;
        const int gx = ox + lx, gy0 = oy + ly0;
        const unsigned long long keys[4] = { nkeys[0], nkeys[1], nkeys[2], nkeys[3] };
        if (g + kRT < kStrips) {                                  // (uniform) the next strip's visibility words, ahead of this strip's pixels
            const int ny0 = ((g + kRT) / TILE) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) nkeys[k] = vis[(ny0 + k) * TILE + lx];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) nkeys[k] = ~0ull;
        }
        const bool inside = whole || (gy0 < a.h && gx < a.w);     // (per lane) this column lies on the target
        uint32_t pix = (uint32_t)__umul24(gy0, a.w) + (uint32_t)gx;          // < 2^28: both factors below 2^14
        // byte offset in a 4-byte plane (a row further: + 4 w)
        uint32_t pix4 = kExpTiledStores ? ((uint32_t)tile * (uint32_t)(TILE * TILE) + (uint32_t)(ly0 * TILE + lx)) << 2 : pix << 2;
        const uint32_t w4 = kExpTiledStores ? (uint32_t)TILE << 2 : (uint32_t)a.w << 2;
#pragma unroll
        for (int k = 0; k < 4; k++, pix += (uint32_t)a.w, pix4 += w4) {
            const unsigned long long key = keys[k];
            const uint32_t low = (uint32_t)key;
            const u32x4 g5 = nrec.g5; const u3 g6 = nrec.g6, g7 = nrec.g7;    // this pixel's record (requested one pixel ago)
            if (!depth_only) {
                const uint32_t nlow = (uint32_t)(k < 3 ? keys[k + 1] : nkeys[0]);
                if (kExpRecOnChange) nrec = fetch_rec(nlow != low ? nlow : 0xffffffffu);             // (timing experiments: wrong image)
                else if (kExpRecMasked) { if (nlow != low) nrec = fetch_rec(nlow); }
                else nrec = fetch_rec(nlow);
            }
            if (!inside || (!whole && gy0 + k >= a.h)) continue;
            const bool cov = low != 0xffffffffu;
            if (!cov && !a.assume_cleared) continue;               // keep what the target holds
            const uint32_t dep = (uint32_t)(key >> 32);
            uint32_t dif = 0, nn0 = 0, nn1 = 0;
            if (RANGES && cov && dep < 0x3f800000u) {
                if (TILE == 64 && ly0 >= 32) { rmin1 = min(rmin1, dep); rmax1 = max(rmax1, dep); }       // (wave-uniform: a strip lies in one half)
                else { rmin0 = min(rmin0, dep); rmax0 = max(rmax0, dep); }
            }
            VR_PROF_MARK(8);
            if (cov && !depth_only) {
                const float q0 = __uint_as_float(g5.x), qx = __uint_as_float(g5.y), qy = __uint_as_float(g5.z);
                const float nx0 = __uint_as_float(g6.x), nxx = __uint_as_float(g6.y), nxy = __uint_as_float(g6.z);
                const float nz0 = __uint_as_float(g7.x), nzx = __uint_as_float(g7.y), nzy = __uint_as_float(g7.z);
                // this pixel's offset from the planes' anchor (exact: small integers)
                const float fdx = (float)(gx - (int)(g5.w & 0xffffu)), fdy = (float)(gy0 + k - (int)(g5.w >> 16));
                // perspective-correct world xz and its screen-space derivatives (oracle: interp)
                Attr p;
                {
                    const float q = plane_at(q0, qx, qy, fdx, fdy);
                    // 1.0f / q: q = the interpolated 1/w of a point in front of the near plane, far inside the short sequence's
                    // range; anything else (a wire pixel far off its triangle's plane) is redone with the general division
                    // (the fast variant has no wire pixels: inside its triangle a pixel's 1/w lies between its vertices')
                    const float aq = fabsf(q);
                    float r = vr_rcp_exact(q);
                    if (!FAST && __builtin_expect(__any(!(aq > 0x1p-60f && aq < 0x1p60f)), 0)) r = 1.0f / q;     // (wave-uniform, practically never taken)
                    p.wx = plane_at(nx0, nxx, nxy, fdx, fdy) * r;
                    p.wz = plane_at(nz0, nzx, nzy, fdx, fdy) * r;
                    p.dwxdx = __builtin_fmaf(-p.wx, qx, nxx) * r; p.dwzdx = __builtin_fmaf(-p.wz, qx, nzx) * r;
                    p.dwxdy = __builtin_fmaf(-p.wx, qy, nxy) * r; p.dwzdy = __builtin_fmaf(-p.wz, qy, nzy) * r;
                }
                VR_PROF_MARK(9);
                if (FAST) pixel_shader_fast(a, rq, rc, thr, enc, s_lv, p, dif, nn0, nn1 VR_PROF_ARG);
                else pixel_shader<false, false>(a, hm, al, rq, rc, thr, enc, s_qoff, s_aoff, p, dif, nn0, nn1);
            }
            if constexpr (LIT) {
                // the pixel as the lighting pass would read it back from the G-buffer: the encoded texel (a cleared one where nothing
                // was drawn), the same decode, the same shading, the same half conversion
                // (a wave of uncovered pixels - cleared texels - shades to +0 in every channel: vr_deferred.hip, k_deferred; wave-uniform)
                float lrgb[3] = { 0.0f, 0.0f, 0.0f };
                if (__any(cov)) shade_pixel<false>(lit.da, s_lut, gx, gy0 + k, __uint_as_float(dep), dif, cov ? spec_const : 0u, nn0, nn1, 0u, 0u, lrgb);
                const uint32_t o0 = vr_float_to_half(lrgb[0]) | (vr_float_to_half(lrgb[1]) << 16), o1 = vr_float_to_half(lrgb[2]);
                __builtin_amdgcn_raw_buffer_store_b32(dep, rgb, pix4, 0, aux);
                if (lit.tile_slot == nullptr) {
                    const u2 ov = { o0, o1 };
                    __builtin_nontemporal_store(ov, reinterpret_cast<u2*>(lit.hdr) + pix);
                } else {
                    // packed tile-major RGB16F (vr_deferred_light's layout for a partition): [local tile][128 rows][128 px] x 6 B
                    const int otx = gx >> 7, oty = (gy0 + k) >> 7;
                    const int lt = lit.tile_slot[oty * lit.owner_tiles_x + otx] - lit.slot_base;
                    const size_t oi = ((size_t)lt * VR_OWNER_TILE + (size_t)((gy0 + k) & 127)) * VR_OWNER_TILE + (size_t)(gx & 127);
                    uint16_t* dst = reinterpret_cast<uint16_t*>(lit.hdr) + oi * 3;
                    dst[0] = (uint16_t)o0; dst[1] = (uint16_t)(o0 >> 16); dst[2] = (uint16_t)o1;
                }
                continue;
            }
            if (kExpNoStore) {        // (a dependent dummy keeps the shading alive)
                if (a.w < 0) __builtin_amdgcn_raw_buffer_store_b32(dep ^ dif ^ nn0 ^ nn1, rgb, pix4, 0, aux);
                continue;
            }
            if (gb_small) {
                __builtin_amdgcn_raw_buffer_store_b32(dep, rgb, pix4, 0, aux);
                if (!depth_only) {
                    const uint32_t pix8 = pix4 + pix4;
                    __builtin_amdgcn_raw_buffer_store_b32(dif, rgb, pix4, o_diff, aux);
                    if (!skip_spec) __builtin_amdgcn_raw_buffer_store_b32(cov ? spec_const : 0u, rgb, pix4, o_spec, aux);
                    const u2 nv = { nn0, nn1 }, zv = { 0u, 0u };
                    __builtin_amdgcn_raw_buffer_store_b64(nv, rgb, pix8, o_nrm, aux);
                    if (!NOEMI && !kExpNoEmissive) __builtin_amdgcn_raw_buffer_store_b64(zv, rgb, pix8, o_emi, aux);
                }
            } else {
                const size_t p64 = (size_t)(gy0 + k) * a.w + gx;
                ST1(g_depth + p64, __uint_as_float(dep));
                if (!depth_only) {
                    ST1(g_diff + p64, dif);
                    ST1(g_spec + p64, cov ? spec_const : 0u);
                    ST2(g_nrm + p64, nn0, nn1);
                    ST2(g_emi + p64, 0u, 0u);
                }
            }
            VR_PROF_MARK(15);
        }
    }
    if (RANGES) {
        // lanes that share a light tile: the 32-lane halves of a wave (64-pixel tiles: neighbouring columns) or the whole wave
        // (32-pixel tiles: two rows of one tile); one lane per light tile merges the wave's range into the G-buffer's
#pragma unroll
        for (int off = 1; off <= (TILE == 64 ? 16 : 32); off <<= 1) {
            rmin0 = min(rmin0, (uint32_t)__shfl_xor((int)rmin0, off)); rmax0 = max(rmax0, (uint32_t)__shfl_xor((int)rmax0, off));
            if (TILE == 64) { rmin1 = min(rmin1, (uint32_t)__shfl_xor((int)rmin1, off)); rmax1 = max(rmax1, (uint32_t)__shfl_xor((int)rmax1, off)); }
        }
        if ((lane & (TILE == 64 ? 31 : 63)) == 0) {
            const int tiles32_x = (a.w + 31) >> 5;
            const int ltx = (ox >> 5) + (TILE == 64 ? lane >> 5 : 0), lty = oy >> 5;
            // (a range exists only where a covered pixel does, i.e. inside the frame: the index is valid)
            if (rmin0 <= rmax0) { uint2* r = ranges + (size_t)lty * tiles32_x + ltx; atomicMin(&r->x, rmin0); atomicMax(&r->y, rmax0); }
            if (TILE == 64 && rmin1 <= rmax1) { uint2* r = ranges + (size_t)(lty + 1) * tiles32_x + ltx; atomicMin(&r->x, rmin1); atomicMax(&r->y, rmax1); }
        }
    }
#undef ST1
#undef ST2
}

// ---------------------------------------------------------------------------------------
// host: TerrainPass::Render (TerrainPass.cpp:143-232)
// ---------------------------------------------------------------------------------------
static int make_raster_args(vr_terrain* t, const vr_view* view, const vr_render_params* rp, int w, int h, const vr_partition* part,
                            RasterArgs& a)
{
    const int world = part ? part->world_size : 1, rank = part ? part->rank : 0;
    VR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad partition");
    memset(&a, 0, sizeof(a));
    a.w = w; a.h = h;
    a.vx0 = view->viewport_x; a.vy0 = view->viewport_y;
    a.vx1 = view->viewport_x + view->viewport_w - 1; a.vy1 = view->viewport_y + view->viewport_h - 1;
    if (a.vx0 < 0) a.vx0 = 0;
    if (a.vy0 < 0) a.vy0 = 0;
    if (a.vx1 > w - 1) a.vx1 = w - 1;
    if (a.vy1 > h - 1) a.vy1 = h - 1;
    a.tile_shift = vr_raster_tile_shift(w, h, world, t->ctx->raster_tile_force);
    { const int rt = 1 << a.tile_shift; a.rtx = (w + rt - 1) / rt; a.rty = (h + rt - 1) / rt; }
    a.mirrored = view->mirrored; a.world = world; a.rank = rank;
    VR_REQUIRE(world <= 64, "at most 64 ranks");                 // (the owner-tile test's reciprocal multiplication is exact up to there)
    a.world_magic = (65536u + (uint32_t)world - 1u) / (uint32_t)world;
    a.depth_only = rp->depth_only; a.assume_cleared = rp->assume_cleared; a.wireframe = rp->wireframe ? 1 : 0;
    a.world_size = t->p.world_size; a.inv_world_size = 1.0f / t->p.world_size;
    { uint32_t wb; memcpy(&wb, &t->p.world_size, 4); a.ws_pow2 = (wb & 0x7fffffu) == 0u && t->p.world_size >= 1.0f && t->p.world_size <= 65536.0f; }
    a.lod_w = (float)t->height.w0 * a.inv_world_size; a.lod_h = (float)t->height.h0 * a.inv_world_size; a.max_level_f = (float)(t->height.levels - 1);
    a.bin_capacity = (uint32_t)t->bin_capacity;
    a.extra_vert_base = (uint32_t)t->cap_instances * kVertsPerInst; a.extra_vert_cap = t->extra_vert_cap; a.hard_cap = t->hard_cap;
    a.vp_x = (float)view->viewport_x; a.vp_y = (float)view->viewport_y; a.vp_w = (float)view->viewport_w; a.vp_h = (float)view->viewport_h;
    return VR_OK;
}

// Grids of the three wide geometry kernels (all grid-stride loops).  Fewer, longer workgroups: the chain shares the device
// with the tile pass, which launches 32 K waves per 8K frame - with 2048 / 4096 / 4096 workgroups the chain launched 41 K of
// its own.  A/B over the grids (profiles/r03_geometry_grids.txt): 8K frame -0.5..1 %, 4K and the emulated N = 8 rank -2 %;
// smaller still (128 / 256 / 128) the kernels' own durations double without another gain.
constexpr int kGridVertex = 512, kGridSetup = 1024, kGridFill = 1024;
// select -> vertex -> setup -> clip -> scan -> fill into `g`, on the geometry stream
static int launch_geometry(vr_terrain* t, GeoSet& g, GeoSet* selection_from, const vr_view* view, const vr_render_params* rp,
                           const RasterArgs& a, const PartTables* pt)
{
    vr_context* ctx = t->ctx;
    // this chain's stream: the terrain's two geometry streams take turns.  Whatever the set did before on the other one
    // is ordered by the events below: its previous chain (consumed by a tile pass or not), the tile pass that read it, and
    // a lock_view copy of its selection into another set.
    g.stream = t->geo_streams[t->geo_turn++ & 1u];
    g.main_waited = false;
    hipStream_t s = ctx->stream, gs = g.stream;
    if (!ctx->async_geometry) {          // single-stream mode: order the geometry behind everything queued so far
        VR_HIP(hipEventRecord(t->ev_main_dep, s));
        g.main_dep_pending = true;
    }
    // after the tile pass that last read this set, and after anything the context's stream did to the terrain
    if (g.geo_recorded) VR_HIP(hipStreamWaitEvent(gs, g.ev_geo_done, 0));
    if (g.sel_read_pending) { VR_HIP(hipStreamWaitEvent(gs, g.ev_sel_read, 0)); g.sel_read_pending = false; }
    // (a stop event of an older timing epoch: the stream was synchronised when the pool was recycled - that tile pass is done)
    if (g.raster_recorded && (g.raster_done_epoch == 0 || g.raster_done_epoch == ctx->ev_epoch)) VR_HIP(hipStreamWaitEvent(gs, g.raster_done, 0));
    if (g.main_dep_pending) { VR_HIP(hipStreamWaitEvent(gs, t->ev_main_dep, 0)); g.main_dep_pending = false; }
    int rc;
    if (selection_from == nullptr) {                                   // TerrainPass.cpp:173-190
        if ((rc = vr_select_launch(t, g, view, rp->max_height, gs))) return rc;
    } else if (selection_from != &g) {
        // lockView: keep the selection of the last unlocked frame (TerrainPass.cpp:191-197); it lives in another set, whose
        // select ran on that set's stream
        VR_HIP(hipEventRecord(t->ev_sel_copy, selection_from->stream));
        VR_HIP(hipStreamWaitEvent(gs, t->ev_sel_copy, 0));
        VR_HIP(hipMemcpyAsync(g.d_node_ids, selection_from->d_node_ids, (size_t)t->p.max_instances * sizeof(uint32_t), hipMemcpyDeviceToDevice, gs));
        VR_HIP(hipMemcpyAsync(g.d_instances, selection_from->d_instances, (size_t)t->p.max_instances * sizeof(vr_instance), hipMemcpyDeviceToDevice, gs));
        VR_HIP(hipMemcpyAsync(g.d_counters, selection_from->d_counters, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, gs));
        // the source set's next writer (a select into it, on either stream) must not overtake these copies
        VR_HIP(hipEventRecord(selection_from->ev_sel_read, gs));
        selection_from->sel_read_pending = true;
        g.have_selection = true;
    }
    const int n_tiles = a.rtx * a.rty;
    if (n_tiles > g.scratch_tiles) {
        VR_HIP(hipStreamSynchronize(gs));
        VR_HIP(hipStreamSynchronize(s));
        (void)hipFree(g.d_tile_count); (void)hipFree(g.d_tile_offset); (void)hipFree(g.d_tile_cursor); (void)hipFree(g.d_tile_order);
        g.d_tile_count = g.d_tile_offset = g.d_tile_cursor = nullptr; g.d_tile_order = nullptr; g.scratch_tiles = 0;
        VR_HIP(hipMalloc(&g.d_tile_count, sizeof(uint32_t) * n_tiles));
        VR_HIP(hipMalloc(&g.d_tile_offset, sizeof(uint32_t) * n_tiles));
        VR_HIP(hipMalloc(&g.d_tile_cursor, sizeof(uint32_t) * n_tiles));
        VR_HIP(hipMalloc(&g.d_tile_order, sizeof(int32_t) * n_tiles * kScanClasses));      // a region per bin-length class
        VR_HIP(hipMemsetAsync(g.d_tile_count, 0, sizeof(uint32_t) * n_tiles, gs));   // k_scan re-zeroes it every frame
        VR_HIP(hipMemsetAsync(g.d_tile_cursor, 0, sizeof(uint32_t) * n_tiles, gs));
        VR_HIP(hipMemsetAsync(g.d_tile_offset, 0, sizeof(uint32_t) * n_tiles, gs));
        g.scratch_tiles = n_tiles;
    }
    g.last_tiles = n_tiles;
    VertexArgs va;
    for (int i = 0; i < 16; i++) { va.w2v[i] = view->world_to_view[i]; va.v2c[i] = view->view_to_clip[i]; }
    va.cam_x = view->camera_pos[0]; va.cam_z = view->camera_pos[2];
    for (int i = 0; i < VR_MAX_LODS; i++) va.lod_ranges[i] = t->lod_ranges[i];
    va.morph_start = t->p.morph_start; va.world_size = t->p.world_size; va.max_height = rp->max_height;
    va.vp_x = a.vp_x; va.vp_y = a.vp_y; va.vp_w = a.vp_w; va.vp_h = a.vp_h;
    { VrKernelScope ks(ctx, VR_K_VERTEX, gs);
    hipLaunchKernelGGL(k_vertex, dim3(kGridVertex), dim3(256), 0, gs, va, t->height, g.d_instances, g.d_counters, g.d_verts); }
    { VrKernelScope ks(ctx, VR_K_SETUP, gs);
    hipLaunchKernelGGL(k_setup, dim3(kGridSetup), dim3(256), 0, gs, a, g.d_verts, g.d_counters, g.d_rect, g.d_hard_list, g.d_tile_count, g.d_recs); }
    { VrKernelScope ks(ctx, VR_K_CLIP, gs);
    hipLaunchKernelGGL(k_clip, dim3(64), dim3(64), 0, gs, a, g.d_verts, g.d_counters, g.d_hard_list, g.d_hard_tris, g.d_hard_first, g.d_tile_count,
                       g.d_recs + (size_t)t->cap_instances * kTrisPerInst * kRecGroups); }
    { VrKernelScope ks(ctx, VR_K_SCAN, gs);
    const bool whole = pt == nullptr;
    const int n_scan = whole ? n_tiles : pt->num_raster_tiles;
    if (n_scan > 0)
        hipLaunchKernelGGL(k_scan, dim3((n_scan + kScanPerGroup - 1) / kScanPerGroup), dim3(256), 0, gs, n_tiles, g.d_tile_count, g.d_tile_offset,
                           g.d_tile_cursor, g.d_counters, a.bin_capacity, whole ? (const int32_t*)nullptr : (const int32_t*)pt->d_raster_tiles, n_scan,
                           g.d_tile_order); }
    { VrKernelScope ks(ctx, VR_K_FILL, gs);
    hipLaunchKernelGGL(k_fill, dim3(kGridFill), dim3(256), 0, gs, a, g.d_counters, g.d_rect, g.d_hard_tris, (const uint4*)g.d_recs,
                       (uint32_t)t->cap_instances * (uint32_t)kTrisPerInst, g.d_tile_cursor, g.d_bin_entries, t->d_status + (size_t)(&g - t->sets) * 8); }
    g.status_pending = true;
    VR_HIP(hipEventRecord(g.ev_geo_done, gs));
    g.geo_recorded = true;
    VR_HIP(hipGetLastError());
    return VR_OK;
}

typedef decltype(&k_raster<false, 32, RM_GENERIC>) raster_kernel_t;
template <int TILE>
static raster_kernel_t pick_raster(bool wire, bool fast, bool depth, bool ranges, bool noemi)
{
    if (wire) return k_raster<true, TILE, RM_GENERIC>;
    if (fast) return ranges ? (noemi ? k_raster<false, TILE, RM_FAST, true, true> : k_raster<false, TILE, RM_FAST, true, false>)
                            : (noemi ? k_raster<false, TILE, RM_FAST, false, true> : k_raster<false, TILE, RM_FAST, false, false>);
    return depth ? k_raster<false, TILE, RM_DEPTH> : k_raster<false, TILE, RM_GENERIC>;
}

static bool prepared_matches(const GeoSet& g, const vr_view* view, const vr_render_params* rp, int w, int h, const RasterArgs& a)
{
    return memcmp(&g.prep_view, view, sizeof(vr_view)) == 0 && g.prep_rp.max_height == rp->max_height && g.prep_rp.depth_only == rp->depth_only
        && !g.prep_rp.wireframe == !rp->wireframe && g.prep_w == w && g.prep_h == h && g.prep_rank == a.rank && g.prep_world == a.world
        && g.prep_tile_shift == a.tile_shift;      // (VR_OPT_RASTER_TILE may have changed in between: the bins are per tile size)
}

static int check_render_inputs(vr_terrain* t, const vr_view* view, vr_gbuffer* gb, const vr_render_params* rp)
{
    VR_REQUIRE(t && view && gb && rp, "NULL argument");
    VR_REQUIRE(!view->reverse_depth, "reverse depth is not supported (the reference disables it, Renderer.cpp:221)");
    VR_REQUIRE(view->viewport_w > 0 && view->viewport_h > 0 && view->viewport_w <= 16384 && view->viewport_h <= 16384, "bad viewport");
    VR_REQUIRE(gb->ctx == t->ctx, "G-buffer and terrain belong to different contexts");
    return VR_OK;
}

extern "C" VR_API int vr_terrain_prepare(vr_terrain* t, const vr_view* view, vr_gbuffer* gb, const vr_render_params* rp,
                                          const vr_partition* part)
{
    int rc = check_render_inputs(t, view, gb, rp);
    if (rc) return rc;
    VR_REQUIRE(!rp->lock_view, "vr_terrain_prepare builds a new selection; it cannot be combined with lock_view");
    VR_HIP(hipSetDevice(t->ctx->device));
    // completed chains' counters: the scratch grows here if due (a sticky condition is vr_terrain_render's to report)
    if ((rc = vr_terrain_poll(t, false))) return rc;
    RasterArgs a;
    if ((rc = make_raster_args(t, view, rp, gb->w, gb->h, part, a))) return rc;
    if ((rc = vr_terrain_reserve_bins(t, (size_t)a.rtx * a.rty / (size_t)(a.world > 1 ? a.world : 1)))) return rc;
    a.bin_capacity = (uint32_t)t->bin_capacity;
    const PartTables* pt = nullptr;       // this rank's raster tiles; unused (NULL) for the whole frame
    if (a.world > 1 && (rc = vr_partition_tables(t->ctx, gb->w, gb->h, part, &pt))) return rc;
    // already prepared for exactly these inputs (a caller may name the same future frame twice): nothing to do
    for (GeoSet& p : t->sets)
        if (p.prepared && prepared_matches(p, view, rp, gb->w, gb->h, a)) return VR_OK;
    GeoSet& g = t->sets[vr_terrain_pick_set(t)];
    g.prepared = false;
    // Start when the context's stream starts the tile pass queued last: the host runs frames ahead of the
    // device, and without this the geometry would become runnable one pass earlier and share the device
    // with the previous frame's lighting pass (bandwidth-bound) instead of with a tile pass (which leaves
    // half of every CU's wave slots free).  (Measured again in round 2, 8K: geometry under the tile pass 483 + 211 us,
    // frame 0.715 ms; geometry under the lighting pass 462 + 239 us, frame 0.723 ms.)
    if (t->raster_begin_recorded && (t->start_hint_epoch == 0 || t->start_hint_epoch == t->ctx->ev_epoch))
        VR_HIP(hipStreamWaitEvent(t->geo_streams[t->geo_turn & 1u], t->start_hint, 0));   // (the stream launch_geometry takes next)
    if ((rc = launch_geometry(t, g, nullptr, view, rp, a, pt))) return rc;
    // Where the context's stream waits for this chain.  Never in front of the tile pass that consumes it (the lighting pass ->
    // tile pass boundary then holds no cross-stream wait) and never earlier than it has to:
    //   - the only prepared set (a loop that prepares one frame ahead): NOW - in a frame loop that is in front of the current
    //     frame's lighting pass, a whole tile pass after the chain started;
    //   - another frame is prepared already (two frames ahead: this chain is not the next tile pass's): behind the NEXT tile
    //     pass (vr_terrain_render queues it), a whole frame later.  Below 8K a chain under load (~0.2 ms) outlasts the tile
    //     pass it starts under (4K: 0.15 ms), and a wait in front of the current lighting pass stalled every frame by the
    //     difference (4K: 0.28 -> 0.22 ms per frame without it).
    bool other_prepared = false;
    for (const GeoSet& p : t->sets) other_prepared |= (&p != &g) && p.prepared;
    if (other_prepared) g.main_waited = false;
    else { VR_HIP(hipStreamWaitEvent(t->ctx->stream, g.ev_geo_done, 0)); g.main_waited = true; g.main_wait_stream = t->ctx->stream; }
    g.prepared = true; g.prep_view = *view; g.prep_rp = *rp; g.prep_w = gb->w; g.prep_h = gb->h; g.prep_rank = a.rank; g.prep_world = a.world; g.prep_tile_shift = a.tile_shift;
    g.prep_serial = ++t->prep_counter;
    return VR_OK;
}

// what vr_terrain_render_lit adds to a render: the lighting pass's inputs and its output image
struct LitRequest { const vr_light* lights; int32_t num_lights; const float* amb_top; const float* amb_bottom; vr_image* hdr; };
static int terrain_render_impl(vr_terrain* t, const vr_view* view, vr_gbuffer* gb, const vr_render_params* rp, const vr_partition* part,
                               const LitRequest* lit_req, bool* lit_done, int* earlier_out);

extern "C" VR_API int vr_terrain_render(vr_terrain* t, const vr_view* view, const vr_view* view_prev, vr_gbuffer* gb,
                                         const vr_render_params* rp, const vr_partition* part)
{
    (void)view_prev;   // MOTION_VECTORS = 0 (TerrainPass.cpp:361,368)
    int earlier = VR_OK;
    const int rc = terrain_render_impl(t, view, gb, rp, part, nullptr, nullptr, &earlier);
    return rc ? rc : earlier;          // this frame is queued either way; `earlier` = a completed frame's device-side condition (sticky, once)
}

// TerrainPass::Render + DeferredLightingPass::Render in one pass over the pixels (Renderer.cpp:401-428): the tile
// pass's resolve shades every pixel it has just produced and writes depth + HdrColor; the other four planes of the G-buffer are
// not written (they keep what they held).  Bit-identical HdrColor and depth to vr_terrain_render(assume_cleared = 1) +
// vr_deferred_light on the same inputs.  The fused kernel exists for the reference's case (the conditions of the tile pass's
// fast variant, up to 16 directional / punctual point lights, no shadow term); anything else runs the two passes one after the
// other inside this call - same result, no saving.
extern "C" VR_API int vr_terrain_render_lit(vr_terrain* t, const vr_view* view, vr_gbuffer* gb, const vr_render_params* rp,
                                             const vr_partition* part, const vr_light* lights, int32_t num_lights,
                                             const float ambient_top[3], const float ambient_bottom[3], vr_image* hdr_out)
{
    VR_REQUIRE(t && view && gb && rp && hdr_out && ambient_top && ambient_bottom, "NULL argument");
    VR_REQUIRE(rp->assume_cleared && !rp->depth_only && !rp->wireframe && !rp->depth_ranges,
               "vr_terrain_render_lit draws into a cleared target (assume_cleared = 1), shaded fill mode, no depth ranges");
    VR_REQUIRE(num_lights >= 0 && num_lights <= kMaxLights && (num_lights == 0 || lights), "at most 16 lights (terrain_cb.h:15)");
    const LitRequest req = { lights, num_lights, ambient_top, ambient_bottom, hdr_out };
    bool fused = false;
    int earlier = VR_OK;
    int rc = terrain_render_impl(t, view, gb, rp, part, &req, &fused, &earlier);
    if (rc) return rc;
    if (!fused && (rc = vr_deferred_light(t->ctx, view, gb, lights, num_lights, ambient_top, ambient_bottom, hdr_out, part))) return rc;     // the unfused pair
    return earlier;
}

static int terrain_render_impl(vr_terrain* t, const vr_view* view, vr_gbuffer* gb, const vr_render_params* rp, const vr_partition* part,
                               const LitRequest* lit_req, bool* lit_done, int* earlier_out)
{
    int rc = check_render_inputs(t, view, gb, rp);
    if (rc) return rc;
    vr_context* ctx = t->ctx;
    VR_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // what earlier frames' chains left in the host mirror: grows the scratch if due, and a device-side condition of a completed
    // frame (too many nodes, a full work list) is returned - once - behind this frame's launches
    const int earlier = vr_terrain_poll(t, true);
    if (earlier && earlier != VR_ERR_OVERFLOW && earlier != VR_ERR_TOO_MANY_INSTANCES) return earlier;     // (the scratch could not grow)
    RasterArgs a;
    if ((rc = make_raster_args(t, view, rp, gb->w, gb->h, part, a))) return rc;
    if ((rc = vr_terrain_reserve_bins(t, (size_t)a.rtx * a.rty / (size_t)(a.world > 1 ? a.world : 1)))) return rc;
    a.bin_capacity = (uint32_t)t->bin_capacity;
    const PartTables* pt = nullptr;       // this rank's raster tiles; unused (NULL) for the whole frame
    if (a.world > 1 && (rc = vr_partition_tables(ctx, gb->w, gb->h, part, &pt))) return rc;
    // RenderTargets::Clear is lazy under the plane-state tracking (vr_gbuffer::clear_pending): a shaded pass over the whole frame
    // writes every pixel of every plane anyway and runs as "over a cleared target" - Clear + Render is one pass over the
    // memory; any other pass (a rank's share, depth only, the fused variant) needs the clear values in memory first
    if (gb->clear_pending) {
        if (a.world <= 1 && !a.depth_only && lit_req == nullptr) { a.assume_cleared = 1; gb->clear_pending = false; }
        else if ((rc = vr_gbuffer_materialise(gb, s))) return rc;
    }

    // a set prepared for exactly this frame, else a free one (the oldest prepared set is given up if all are taken)
    int gi = -1;
    if (!rp->lock_view)
        for (int i = 0; i < kGeoSets; i++)
            if (i != t->cur && t->sets[i].prepared && prepared_matches(t->sets[i], view, rp, gb->w, gb->h, a)) { gi = i; break; }
    const bool use_prepared = gi >= 0;
    if (!use_prepared) gi = vr_terrain_pick_set(t);
    GeoSet& g = t->sets[gi];
    GeoSet& last = t->sets[t->cur];
    g.prepared = false;
    if (!use_prepared) {
        GeoSet* sel = (rp->lock_view && last.have_selection) ? &last : nullptr;
        if ((rc = launch_geometry(t, g, sel, view, rp, a, pt))) return rc;
    }
    t->cur = gi;
    // the tile pass consumes verts + bins (a wait queued at prepare time counts only if it sits on the stream this pass runs on)
    if (!(use_prepared && g.main_waited && g.main_wait_stream == s)) VR_HIP(hipStreamWaitEvent(s, g.ev_geo_done, 0));
    g.main_waited = false;
    const uint32_t spec_const = vr_specular_constant(ctx);               // terrain_ps.hlsl:76 -> SRGBA8
    const int grid = pt ? pt->num_raster_tiles : a.rtx * a.rty;
    // vr_terrain_prepare's start hint: "the context's stream has reached this tile pass".  With dispatch-stamped events that
    // is the stop event of whatever ran last on the stream (the previous frame's lighting pass); else an explicit record.
    if (ctx->dispatch_events && ctx->last_stop) { t->start_hint = ctx->last_stop; t->start_hint_epoch = ctx->ev_epoch; t->raster_begin_recorded = true; }
    else { VR_HIP(hipEventRecord(t->ev_raster_begin, s)); t->start_hint = t->ev_raster_begin; t->start_hint_epoch = 0; t->raster_begin_recorded = true; }
    hipEvent_t pass_stop = nullptr;
    if (grid > 0) {
        // a depth-only tile pass (the shadow map's) is timed under its own id: it is an order of magnitude shorter than the
        // G-buffer pass and must not be averaged with it
        VrKernelScope ks(ctx, rp->depth_only ? VR_K_RASTER_DEPTH : VR_K_RASTER, s, true);
        const int32_t* tiles = g.d_tile_order;            // this frame's tiles, longest bins first (k_scan)
        // the fast variant: heightmap and albedo of one size (the albedo footprint shares the height taps' coordinates), a
        // power-of-two world size, the five planes of the G-buffer within 4 GB (one buffer resource), filled and shaded
        const bool same = t->height.w0 == t->albedo.w0 && t->height.h0 == t->albedo.h0 && t->height.levels == t->albedo.levels;
        const uint64_t span = (uint64_t)((const char*)(gb->emissive + (size_t)gb->w * gb->h) - (const char*)gb->depth);
        const bool one_rsrc = (const char*)gb->depth < (const char*)gb->diffuse && (const char*)gb->depth < (const char*)gb->specular
                           && (const char*)gb->depth < (const char*)gb->normals && (const char*)gb->depth < (const char*)gb->emissive && span < (1ull << 32);
        const bool fast = same && a.ws_pow2 && one_rsrc && !a.wireframe && !a.depth_only;
        const bool depth = a.depth_only && !a.wireframe;
        // the light tiles' depth ranges, if asked for: only from the fast variant over a target it fills completely
        const bool ranges = rp->depth_ranges && fast && a.assume_cleared && !gb->escaped;
        if (ranges) {
            if ((rc = vr_gbuffer_ranges_prepare(gb, s))) return rc;
            gb->ranges_state = vr_gbuffer::RANGES_VALID; gb->ranges_rank = a.rank; gb->ranges_world = a.world;
        } else vr_gbuffer_touch(gb);
        // plane-state tracking: the emissive plane holds zeros already and the fast variant would only write zeros again
        const bool noemi = fast && ctx->plane_tracking && gb->emissive_zero && !gb->escaped;
        auto kern = a.tile_shift == 5 ? pick_raster<32>(a.wireframe != 0, fast, depth, ranges, noemi) : pick_raster<64>(a.wireframe != 0, fast, depth, ranges, noemi);
        // the fused variant: only where the fast variant applies and the light list is the streaming pass's plain case
        LitArgs la;
        bool fuse = false;
        if (lit_req && fast && !ranges) {
            bool extra = false;
            memset(&la, 0, sizeof(la));
            if ((rc = vr_deferred_make_args(view, gb->w, gb->h, lit_req->lights, lit_req->num_lights, lit_req->amb_top, lit_req->amb_bottom, &la.da, &extra))) return rc;
            VR_REQUIRE(view->viewport_w == gb->w && view->viewport_h == gb->h && view->viewport_x == 0 && view->viewport_y == 0,
                       "view viewport must cover the G-buffer");
            fuse = !extra && gb->w % 4 == 0;
            if (fuse) {
                la.lut_g = ctx->d_srgb_lut; la.hdr = (uint2*)lit_req->hdr->data;
                if (pt) {
                    VR_REQUIRE((size_t)pt->max_owned * VR_OWNER_TILE * VR_OWNER_TILE * 6 <= lit_req->hdr->capacity_bytes, "hdr_out is smaller than vr_partition_packed_bytes()");
                    la.tile_slot = pt->d_tile_slot; la.slot_base = pt->rank * pt->max_owned; la.owner_tiles_x = (gb->w + VR_OWNER_TILE - 1) / VR_OWNER_TILE;
                    la.da.tiles_x = la.owner_tiles_x;
                } else VR_REQUIRE((size_t)gb->w * gb->h * 8 <= lit_req->hdr->capacity_bytes, "hdr_out is smaller than the frame");
            }
        }
        if (lit_done) *lit_done = fuse;
        // region states: kept by the fast variant on 32-pixel tiles; any other variant writes the planes without keeping them
        uint8_t* region = nullptr;
        if (fast && !fuse && a.tile_shift == 5 && ctx->plane_tracking && !gb->escaped) { if ((rc = vr_gbuffer_region_prepare(gb, s, &region))) return rc; }
        else gb->region_fill = 0;
#define VR_RASTER_ARGS a, t->height, t->albedo, g.d_verts, g.d_hard_tris, g.d_hard_first, \
                           (const uint4*)g.d_recs, (uint32_t)t->cap_instances * (uint32_t)kTrisPerInst, g.d_tile_cursor, g.d_tile_offset, g.d_bin_entries, tiles, g.d_counters + C_CLASS0, \
                           gb->depth, gb->diffuse, gb->specular, gb->normals, gb->emissive, ctx->d_srgb_thr, ctx->d_enc_tab, spec_const, ranges ? gb->d_ranges : (uint2*)nullptr, region
        if (fuse) {
            ks.id = VR_K_RASTER_LIT;
            if (a.tile_shift == 5) VR_LAUNCH_TIMED(ks, (k_raster<false, 32, RM_FAST, false, true, true>), dim3(grid), dim3(kRT), s, VR_RASTER_ARGS, la);
            else VR_LAUNCH_TIMED(ks, (k_raster<false, 64, RM_FAST, false, true, true>), dim3(grid), dim3(kRT), s, VR_RASTER_ARGS, la);
        } else
        VR_LAUNCH_TIMED(ks, kern, dim3(grid), dim3(kRT), s, VR_RASTER_ARGS, LitNone());
#undef VR_RASTER_ARGS
        if (ctx->dispatch_events && ks.e0 && ks.e1) pass_stop = ks.e1;        // stamped by the dispatch: complete when the tile pass is
        // a shaded pass over a cleared target writes the emissive texel (0) of EVERY pixel of the frame, covered or not: from here
        // on the plane is known zero again, whatever it held (a partitioned or keep-what-is-there pass writes zeros to some pixels:
        // the state stays what it was)
        if (!fuse && !noemi && !a.depth_only && a.assume_cleared && a.world <= 1) gb->emissive_zero = true;     // (the fused variant writes depth only)
    }
    if (pass_stop) { g.raster_done = pass_stop; g.raster_done_epoch = ctx->ev_epoch; }
    else { VR_HIP(hipEventRecord(g.ev_raster_done, s)); g.raster_done = g.ev_raster_done; g.raster_done_epoch = 0; }
    g.raster_recorded = true;
    // chains prepared further ahead whose wait vr_terrain_prepare left for later: behind this tile pass
    for (GeoSet& p : t->sets)
        if (&p != &g && p.prepared && !(p.main_waited && p.main_wait_stream == s) && p.geo_recorded) { VR_HIP(hipStreamWaitEvent(s, p.ev_geo_done, 0)); p.main_waited = true; p.main_wait_stream = s; }
    VR_HIP(hipGetLastError());
    if (earlier_out) *earlier_out = earlier;
    return VR_OK;
}

extern "C" VR_API int vr_debug_tile_order(vr_terrain* t, int32_t* out_tiles, uint32_t* out_bin_lengths, int32_t capacity, int32_t* out_count)
{
    VR_REQUIRE(t && out_tiles && out_bin_lengths && out_count && capacity >= 0, "bad arguments");
    VR_HIP(hipSetDevice(t->ctx->device));
    const GeoSet& g = t->sets[t->cur];
    VR_HIP(hipStreamSynchronize(g.stream));
    VR_HIP(hipStreamSynchronize(t->ctx->stream));
    *out_count = 0;
    const int n_tiles = g.last_tiles;
    if (n_tiles <= 0 || !g.d_tile_order) return VR_OK;
    uint32_t cls[kScanClasses];
    VR_HIP(hipMemcpy(cls, g.d_counters + C_CLASS0, sizeof(cls), hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint32_t c : cls) total += c;
    VR_REQUIRE(total <= (uint64_t)n_tiles && total <= (uint64_t)capacity, "capacity too small for the launch order");
    std::vector<int32_t> order((size_t)n_tiles * kScanClasses);
    std::vector<uint32_t> cursor((size_t)n_tiles), offset((size_t)n_tiles);
    VR_HIP(hipMemcpy(order.data(), g.d_tile_order, order.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    VR_HIP(hipMemcpy(cursor.data(), g.d_tile_cursor, cursor.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    VR_HIP(hipMemcpy(offset.data(), g.d_tile_offset, offset.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int k = 0;
    for (int c = 0; c < kScanClasses; c++)
        for (uint32_t i = 0; i < cls[c]; i++) {
            const int32_t tile = order[(size_t)c * n_tiles + i];
            VR_REQUIRE(tile >= 0 && tile < n_tiles, "launch order names a tile outside the target");
            out_tiles[k] = tile; out_bin_lengths[k] = cursor[(size_t)tile] - offset[(size_t)tile]; k++;
        }
    *out_count = k;
    return VR_OK;
}
