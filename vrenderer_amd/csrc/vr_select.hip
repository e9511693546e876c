// Terrain object + quadtree LOD select on the device.
//
// The reference builds a pointer quadtree of (4^(L+1)-1)/3 heap nodes at start-up
// (QuadTree::Split, QuadTree.cpp:210-232; 5.6 M nodes / 313 MB at L=11) and walks it
// recursively every frame (QuadTree::NodeSelect, QuadTree.cpp:80-131).  Here the tree
// is implicit: a node is its path from the root (2 bits per level, child order
// TL,TR,BL,BR as QuadTree.h:51-54), its position/extents are recomputed by replaying
// the same float additions Split performs, and selection is a level-synchronous sweep
// in one workgroup: a frontier of paths in LDS, one lane per node, survivors expanded
// into the next frontier.  The reference's m_SelectedNodes order (depth-first push
// order) is the lexicographic order of left-aligned paths, recovered by a rank sort.
// Nothing is allocated per frame and nothing crosses PCIe: UpdateTransforms
// (TerrainPass.cpp:234-256) is fused into the write-out.
#include "vr_internal.h"
#include "vr_experiments.h"

#include <math.h>
#include <string.h>
#include <stdlib.h>

// One workgroup of 256 threads with 16 KB of LDS: small enough to start on a CU that is full of tile-pass workgroups
// (four of them leave 17 KB of LDS, sixteen wave slots and 128 registers per SIMD free).  Round 2's version - 1024
// threads, 48 KB - had to wait for a CU to drain: 27 us alone, 90-130 us under the tile pass it shares the device with.
constexpr int kSelThreads = 256;
constexpr int kFrontierCap = 4096;   // entries per frontier buffer: the first kFrontLds in LDS, the rest in the set's global scratch
constexpr int kFrontLds = 1536;
constexpr int kSelectedCap = 4096;   // == the largest max_instances (more is an error anyway); the keys live in the global scratch
constexpr int kRankChunk = 1024;     // the rank sort streams the keys through LDS this many at a time
constexpr int kSelScratchWords = 2 * (kFrontierCap - kFrontLds) + kSelectedCap;

struct SelectArgs {
    float cam[3];
    float planes[6][4];
    float range2[VR_MAX_LODS];   // m_LodRanges[i]^2
    float loc[3];
    float half_w, half_h;        // root extents (m_Width/2, m_Height/2)
    int num_lods;
    int max_instances;
    int cap_instances;           // what the scratch of this frame holds (<= max_instances)
    int height_loaded;           // m_HeightLoaded
    float max_height;
    // multi-surface worlds (TerrainPass.cpp:97-110): one quadtree per surface
    int surfaces_per_side, num_surfaces;
    float surface_size;
    uint32_t nodes_per_tree;
};

constexpr int kPathBits = 22;    // 2 bits x 11 levels; frontier entry = surface << 22 | path

// location of surface s (TerrainPass.cpp:102-108): ((-0.5(n-1) + column) * S, 0, (-0.5(n-1) + row) * S)
__device__ __forceinline__ void surface_location(const SelectArgs& a, int s, float loc[3])
{
    const int column = s % a.surfaces_per_side, row = s / a.surfaces_per_side;
    const float x = -0.5f * (float)(a.surfaces_per_side - 1) + (float)column, y = -0.5f * (float)(a.surfaces_per_side - 1) + (float)row;
    loc[0] = a.loc[0] + x * a.surface_size; loc[1] = a.loc[1] + 0.0f; loc[2] = a.loc[2] + y * a.surface_size;
}

struct NodeGeom { float px, pz, ex, ez; uint32_t ix, iz; };

// Replays QuadTree::Split's arithmetic along a path (QuadTree.cpp:212-216).
__device__ __forceinline__ NodeGeom node_from_path(const SelectArgs& a, uint32_t path, int depth, int surface = 0)
{
    float loc[3];
    surface_location(a, surface, loc);
    NodeGeom g; g.px = loc[0]; g.pz = loc[2]; g.ex = a.half_w; g.ez = a.half_h; g.ix = 0; g.iz = 0;
    for (int d = depth - 1; d >= 0; d--) {
        uint32_t c = (path >> (2 * d)) & 3u;
        g.ex = g.ex / 2.0f; g.ez = g.ez / 2.0f;
        bool right = (c & 1u) != 0u, top = c < 2u;      // TL=0 TR=1 BL=2 BR=3
        g.px = right ? g.px + g.ex : g.px - g.ex;
        g.pz = top ? g.pz + g.ez : g.pz - g.ez;
        g.ix = g.ix * 2u + (right ? 1u : 0u);
        g.iz = g.iz * 2u + (top ? 1u : 0u);
    }
    return g;
}

// Node::Intersects (QuadTree.h:31-45): xz distance to the box vs a SQUARED range.
__device__ __forceinline__ bool node_in_range(const NodeGeom& g, const float cam[3], float r2)
{
    float minx = g.px - g.ex, maxx = g.px + g.ex, minz = g.pz - g.ez, maxz = g.pz + g.ez;
    float dx = 0.0f, dz = 0.0f;
    if (cam[0] < minx) dx = cam[0] - minx; else if (cam[0] > maxx) dx = cam[0] - maxx;
    if (cam[2] < minz) dz = cam[2] - minz; else if (cam[2] > maxz) dz = cam[2] - maxz;
    return ((dx * dx + 0.0f) + dz * dz) <= r2;
}

// dm::frustum::intersectsWith(box3) (call site QuadTree.cpp:97-99).
__device__ __forceinline__ bool box_in_frustum(const SelectArgs& a, float mnx, float mny, float mnz, float mxx, float mxy, float mxz)
{
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float nx = a.planes[i][0], ny = a.planes[i][1], nz = a.planes[i][2];
        float x = nx > 0.0f ? mnx : mxx, y = ny > 0.0f ? mny : mxy, z = nz > 0.0f ? mnz : mxz;
        float dist = ((nx * x + ny * y) + nz * z) - a.planes[i][3];
        if (dist > 0.0f) return false;
    }
    return true;
}

__device__ __forceinline__ uint32_t node_id_of(const NodeGeom& g, int depth)
{
    return (uint32_t)((((uint64_t)1 << (2 * depth)) - 1) / 3) + g.iz * (1u << depth) + g.ix;
}

__global__ __launch_bounds__(kSelThreads) void k_select(SelectArgs a, uint32_t* __restrict__ node_ids,
                                                        vr_instance* __restrict__ inst, uint32_t* __restrict__ counters,
                                                        const float2* __restrict__ heights, uint32_t* __restrict__ scratch)
{
    VR_GEOMETRY_PRIORITY();
    __shared__ uint32_t frontier[2][kFrontLds];
    __shared__ __attribute__((aligned(16))) uint32_t chunk[kRankChunk];
    __shared__ uint32_t n_front[2], n_sel, overflow;
    uint32_t* __restrict__ spill[2] = { scratch, scratch + (kFrontierCap - kFrontLds) };
    uint32_t* __restrict__ selected = scratch + 2 * (kFrontierCap - kFrontLds);
    // (the scratch is written and read by this one workgroup only, with a barrier in between: one CU, one L1)
#define FR_GET(buf, i) ((i) < (uint32_t)kFrontLds ? frontier[buf][i] : spill[buf][(i) - (uint32_t)kFrontLds])
#define FR_PUT(buf, i, v) do { if ((i) < (uint32_t)kFrontLds) frontier[buf][i] = (v); else spill[buf][(i) - (uint32_t)kFrontLds] = (v); } while (0)
    const int tid = threadIdx.x;
    const int L = a.num_lods;
#ifdef VR_SELECT_PROFILE
    unsigned long long prof_t[6]; int prof_i = 0;
#define SEL_MARK() do { prof_t[prof_i++] = __builtin_readcyclecounter(); } while (0)
#else
#define SEL_MARK() do { } while (0)
#endif
    SEL_MARK();
    if (tid < a.num_surfaces) frontier[0][tid] = (uint32_t)tid << kPathBits;      // every quadtree's root (at most 64)
    if (tid == 0) { n_front[0] = (uint32_t)a.num_surfaces; n_front[1] = 0u; n_sel = 0u; overflow = 0u; }
    __syncthreads();

    int cur = 0;
    for (int lod = L; lod >= 0; lod--) {
        const int depth = L - lod;
        const uint32_t n = n_front[cur];
        for (uint32_t i = tid; i < n; i += kSelThreads) {
            const uint32_t entry = FR_GET(cur, i);
            const uint32_t path = entry & ((1u << kPathBits) - 1u);
            const int surf = (int)(entry >> kPathBits);
            NodeGeom g = node_from_path(a, path, depth, surf);
            bool select_self = false, expand = false;
            if (!node_in_range(g, a.cam, a.range2[lod])) {
                // NodeSelect returned false: the parent pushes this child (QuadTree.cpp:120-126);
                // the root's return value is ignored (TerrainPass.cpp:181).
                select_self = depth > 0;    // (a root's return value is ignored for every tree)
            } else {
                float mny = 0.0f, mxy = a.cam[1];                 // m_HeightLoaded == false (QuadTree.cpp:92-96)
                if (a.height_loaded) {                            // QuadTree.cpp:87-91
                    const float2 hy = heights[(uint32_t)surf * a.nodes_per_tree + node_id_of(g, depth)];
                    mny = (hy.x - hy.y) * a.max_height; mxy = (hy.x + hy.y) * a.max_height;
                }
                if (box_in_frustum(a, g.px - g.ex, mny, g.pz - g.ez, g.px + g.ex, mxy, g.pz + g.ez)) {
                    if (lod == 0) select_self = true;                                   // :106-111
                    else if (!node_in_range(g, a.cam, a.range2[lod - 1])) select_self = true;   // :114-118
                    else expand = true;                                                 // :119-128
                }
            }
            if (select_self) {
                uint32_t slot = atomicAdd(&n_sel, 1u);
                const uint32_t skey = ((uint32_t)surf << 26) | ((path << (2 * (L - depth))) << 4) | (uint32_t)depth;
                if (slot < (uint32_t)kRankChunk) chunk[slot] = skey;          // the first chunk stays in LDS for the rank pass
                if (slot < (uint32_t)kSelectedCap) selected[slot] = skey;
                else overflow = 1u;
            }
            if (expand) {
                uint32_t slot = atomicAdd(&n_front[cur ^ 1], 4u);
                if (slot + 4u <= (uint32_t)kFrontierCap) {
#pragma unroll
                    for (uint32_t c = 0; c < 4u; c++) FR_PUT(cur ^ 1, slot + c, ((uint32_t)surf << kPathBits) | (path << 2) | c);
                } else overflow = 1u;
            }
        }
        __syncthreads();
        if (tid == 0) { n_front[cur] = 0u; if (n_front[cur ^ 1] > (uint32_t)kFrontierCap) n_front[cur ^ 1] = 0u; }
        cur ^= 1;
        __syncthreads();
    }
#undef FR_GET
#undef FR_PUT
    SEL_MARK();

    // Depth-first order = ascending left-aligned path.  Keys are unique, so a node's rank is the number of smaller
    // keys: the keys stream through LDS a chunk at a time and every thread counts for the (up to 16) keys it owns
    // (LDS broadcast reads).
    uint32_t total = n_sel;
    if (total > (uint32_t)kSelectedCap) total = kSelectedCap;
    const uint32_t limit = min(total, (uint32_t)a.max_instances);             // ids and instances: the selection itself (their arrays hold max_instances)
    constexpr int kOwn = kSelectedCap / kSelThreads;
    const int own_n = (int)((total + kSelThreads - 1) / kSelThreads);        // keys per thread that exist at all (uniform; 2 for ~300 nodes)
    uint32_t key[kOwn], rank[kOwn];
#pragma unroll
    for (int o = 0; o < kOwn; o++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)o * kSelThreads;
        key[o] = i < total ? (i < (uint32_t)kRankChunk ? chunk[i] : selected[i]) : 0xffffffffu; rank[o] = 0u;
    }
    for (uint32_t c0 = 0; c0 < total; c0 += kRankChunk) {
        const uint32_t m = min(total - c0, (uint32_t)kRankChunk);
        __syncthreads();
        // eight keys per step (two 16-byte LDS reads in flight) against every key this thread owns: the pass is bounded by
        // LDS latency, not by the compares (one key per step: 15 us of the kernel's 37 for 290 nodes)
        const uint32_t m8 = (m + 7u) & ~7u;                   // (kRankChunk is a multiple of 8)
        if (c0 == 0u) { for (uint32_t j = tid + m; j < m8; j += kSelThreads) chunk[j] = 0xffffffffu; }          // (already in LDS) pad: counts for no key
        else for (uint32_t j = tid; j < m8; j += kSelThreads) chunk[j] = j < m ? selected[c0 + j] : 0xffffffffu;
        __syncthreads();
        for (uint32_t j = 0; j < m8; j += 8) {
            const uint4 q0 = *reinterpret_cast<const uint4*>(&chunk[j]), q1 = *reinterpret_cast<const uint4*>(&chunk[j + 4]);
#pragma unroll
            for (int o = 0; o < kOwn; o++) {
                if (o >= own_n) break;                                        // (uniform)
                const uint32_t ko = key[o];
                rank[o] += (q0.x < ko ? 1u : 0u) + (q0.y < ko ? 1u : 0u) + (q0.z < ko ? 1u : 0u) + (q0.w < ko ? 1u : 0u)
                         + (q1.x < ko ? 1u : 0u) + (q1.y < ko ? 1u : 0u) + (q1.z < ko ? 1u : 0u) + (q1.w < ko ? 1u : 0u);
            }
        }
    }
    SEL_MARK();
#pragma unroll
    for (int o = 0; o < kOwn; o++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)o * kSelThreads;
        if (i >= total || rank[o] >= limit) continue;
        const uint32_t k = key[o];
        int depth = (int)(k & 15u);
        const int surf = (int)(k >> 26);
        uint32_t path = ((k >> 4) & ((1u << kPathBits) - 1u)) >> (2 * (L - depth));
        NodeGeom g = node_from_path(a, path, depth, surf);
        const uint32_t id = (uint32_t)surf * a.nodes_per_tree + node_id_of(g, depth);
        node_ids[rank[o]] = id;
        float py = a.loc[1], ey = 0.0f;
        if (a.height_loaded) { const float2 hy = heights[id]; py = hy.x; ey = hy.y; }
        // scaling(extents) * translation(position) -> float3x4 rows (TerrainPass.cpp:245-253)
        vr_instance ins;
        ins.padding = 0u; ins.first_geometry_instance_index = 0u; ins.first_geometry_index = 0u; ins.num_geometries = 1u;
#pragma unroll
        for (int q = 0; q < 12; q++) { ins.transform[q] = 0.0f; ins.prev_transform[q] = 0.0f; }
        ins.transform[0] = g.ex; ins.transform[3] = g.px;
        ins.transform[5] = ey; ins.transform[7] = py;
        ins.transform[10] = g.ez; ins.transform[11] = g.pz;
        inst[rank[o]] = ins;
    }
    SEL_MARK();
#ifdef VR_SELECT_PROFILE
    if (tid == 0) for (int q = 0; q < 4; q++) { selected[kSelectedCap - 8 + 2 * q] = (uint32_t)prof_t[q]; selected[kSelectedCap - 7 + 2 * q] = (uint32_t)(prof_t[q] >> 32); }
#endif
    if (tid == 0) {
        counters[0] = min(limit, (uint32_t)a.cap_instances);                   // what the rest of the chain draws: as many as its scratch holds
        counters[7] = limit;                                                   // what NodeSelect returns
        uint32_t flags = 0;
        if (n_sel > (uint32_t)a.max_instances) flags |= 1u;      // VR_ERR_TOO_MANY_INSTANCES
        else if (n_sel > (uint32_t)a.cap_instances) flags |= 4u; // more than the scratch holds (it grows: vr_terrain_poll)
        counters[6] = n_sel;                                     // the count before any truncation (the scratch's high-water mark)
        if (overflow) flags |= 2u;                               // VR_ERR_OVERFLOW
        counters[1] = flags;
    }
}

int vr_select_launch(vr_terrain* t, GeoSet& g, const vr_view* view, float max_height, hipStream_t stream)
{
    SelectArgs a;
    for (int i = 0; i < 3; i++) { a.cam[i] = view->camera_pos[i]; a.loc[i] = t->p.location[i]; }
    for (int i = 0; i < 6; i++) for (int j = 0; j < 4; j++) a.planes[i][j] = view->planes[i][j];
    for (int i = 0; i < VR_MAX_LODS; i++) a.range2[i] = t->lod_ranges[i] * t->lod_ranges[i];
    a.half_w = t->p.surface_size / 2.0f; a.half_h = t->p.surface_size / 2.0f;
    a.num_lods = t->num_lods; a.max_instances = t->p.max_instances; a.cap_instances = t->cap_instances; a.max_height = max_height;
    a.height_loaded = (t->height_loaded && t->d_node_heights) ? 1 : 0;
    a.surfaces_per_side = t->surfaces_per_side; a.num_surfaces = t->surfaces_per_side * t->surfaces_per_side;
    a.surface_size = t->p.surface_size;
    a.nodes_per_tree = (uint32_t)((((uint64_t)1 << (2 * (t->num_lods + 1))) - 1) / 3);
    VrKernelScope ks(t->ctx, VR_K_SELECT, stream);
    hipLaunchKernelGGL(k_select, dim3(1), dim3(kSelThreads), 0, stream, a, g.d_node_ids, g.d_instances, g.d_counters,
                       (const float2*)t->d_node_heights, g.d_sel_scratch);
    VR_HIP(hipGetLastError());
    g.have_selection = true;
    return VR_OK;
}

// ---- QuadTree::SetHeight on the device (QuadTree.cpp:164-208) ----------------------------------
struct HeightArgs {
    float loc[3]; float half_w, half_h; float world_size; float texel_x, texel_y;
    int tex_w, tex_h; int depth;
};

// texel rectangle of a node, exactly as GetMinMaxHeightValue computes it
__device__ __forceinline__ void node_texel_rect(const HeightArgs& a, const NodeGeom& g, int& x0, int& x1, int& y0, int& y1)
{
    const float width = g.ex * 2.0f, height = g.ez * 2.0f;
    float minx = g.px - width / 2, miny = g.pz - height / 2;
    minx += a.world_size / 2; miny += a.world_size / 2;
    minx *= a.texel_x; miny *= a.texel_y;
    const float maxx = minx + width * a.texel_x, maxy = miny + height * a.texel_y;
    x0 = (int)floorf(minx); x1 = (int)ceilf(maxx); y0 = (int)floorf(miny); y1 = (int)ceilf(maxy);
}
__device__ __forceinline__ int height_byte(const HeightArgs& a, const uint8_t* __restrict__ tex, int i, int j)
{
    // GetHeightValue: index = int(x + y * width) in float; clamped where the reference would read out of bounds
    int index = (int)((float)i + (float)j * (float)a.tex_w);
    const int n = a.tex_w * a.tex_h;
    index = index < 0 ? 0 : (index >= n ? n - 1 : index);
    return tex[index];
}
__device__ __forceinline__ float2 finish_minmax(int mnb, int mxb)
{
    float mn = mnb <= 255 ? (float)mnb / 255.0f : INFINITY, mx = mxb >= 0 ? (float)mxb / 255.0f : -INFINITY;
    mn = (mx - mn) == 0.0f ? 0.0f : mn;
    const float extent = (mx - mn) / 2.0f;
    return make_float2(mn + extent, extent);          // m_Position.y, m_Extents.y (QuadTree.cpp:197-198)
}
__device__ __forceinline__ NodeGeom node_from_index(const HeightArgs& a, uint32_t ix, uint32_t iz, int depth)
{
    SelectArgs s;   // only the fields node_from_path reads; a.loc is already the surface's location
    s.loc[0] = a.loc[0]; s.loc[1] = a.loc[1]; s.loc[2] = a.loc[2]; s.half_w = a.half_w; s.half_h = a.half_h;
    s.surfaces_per_side = 1; s.surface_size = 0.0f;
    uint32_t path = 0;
    for (int l = depth - 1; l >= 0; l--) {
        const uint32_t bx = (ix >> l) & 1u, bz = (iz >> l) & 1u;
        path = (path << 2) | (bz ? (bx ? 1u : 0u) : (bx ? 3u : 2u));
    }
    return node_from_path(s, path, depth);
}

// one workgroup per node (large texel rectangles)
__global__ __launch_bounds__(256) void k_node_heights_block(HeightArgs a, const uint8_t* __restrict__ tex, float2* __restrict__ out)
{
    __shared__ int s_mn[4], s_mx[4];
    const uint32_t n = 1u << a.depth, node = blockIdx.x, ix = node % n, iz = node / n;
    const NodeGeom g = node_from_index(a, ix, iz, a.depth);
    int x0, x1, y0, y1; node_texel_rect(a, g, x0, x1, y0, y1);
    const int w = max(x1 - x0, 0), h = max(y1 - y0, 0);
    int mn = 256, mx = -1;
    for (long long k = threadIdx.x; k < (long long)w * h; k += 256) {
        const int b = height_byte(a, tex, x0 + (int)(k % w), y0 + (int)(k / w));
        mn = min(mn, b); mx = max(mx, b);
    }
    for (int off = 32; off >= 1; off >>= 1) { mn = min(mn, __shfl_xor(mn, off)); mx = max(mx, __shfl_xor(mx, off)); }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3])); mx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
        const uint32_t base = (uint32_t)((((uint64_t)1 << (2 * a.depth)) - 1) / 3);
        out[base + node] = finish_minmax(mn, mx);
    }
}
// one lane per node (small rectangles)
__global__ __launch_bounds__(256) void k_node_heights_thread(HeightArgs a, const uint8_t* __restrict__ tex, float2* __restrict__ out)
{
    const uint32_t n = 1u << a.depth;
    const uint64_t total = (uint64_t)n * n;
    for (uint64_t node = (uint64_t)blockIdx.x * 256 + threadIdx.x; node < total; node += (uint64_t)gridDim.x * 256) {
        const uint32_t ix = (uint32_t)(node % n), iz = (uint32_t)(node / n);
        const NodeGeom g = node_from_index(a, ix, iz, a.depth);
        int x0, x1, y0, y1; node_texel_rect(a, g, x0, x1, y0, y1);
        int mn = 256, mx = -1;
        for (int i = x0; i < x1; i++) for (int j = y0; j < y1; j++) { const int b = height_byte(a, tex, i, j); mn = min(mn, b); mx = max(mx, b); }
        const uint32_t base = (uint32_t)((((uint64_t)1 << (2 * a.depth)) - 1) / 3);
        out[base + node] = finish_minmax(mn, mx);
    }
}

// Mip-style path (the usual case: a power-of-two surface sampled one texel per world unit): every node's texel
// rectangle is then an exact dyadic square and the four children tile their parent, so the raw (min, max) bytes of a
// level follow from the level below; the reference's "max == min => min = 0" quirk and the float conversion are
// applied per node on the way out (finish_minmax), which keeps the result identical to the per-node scan.
__global__ __launch_bounds__(256) void k_minmax_leaf(const uint8_t* __restrict__ tex, int tex_w, int rx0, int ry0, uint32_t n, int s,
                                                      uchar2* __restrict__ mm, float2* __restrict__ out)
{
    const uint64_t total = (uint64_t)n * n;
    for (uint64_t node = (uint64_t)blockIdx.x * 256 + threadIdx.x; node < total; node += (uint64_t)gridDim.x * 256) {
        const uint32_t ix = (uint32_t)(node % n), iz = (uint32_t)(node / n);
        const uint8_t* p = tex + (size_t)(ry0 + (int)iz * s) * tex_w + (rx0 + (int)ix * s);
        int mn = 256, mx = -1;
        for (int j = 0; j < s; j++) for (int i = 0; i < s; i++) { const int b = p[(size_t)j * tex_w + i]; mn = min(mn, b); mx = max(mx, b); }
        mm[node] = make_uchar2((unsigned char)mn, (unsigned char)mx);
        out[node] = finish_minmax(mn, mx);
    }
}
__global__ __launch_bounds__(256) void k_minmax_up(const uchar2* __restrict__ child, uint32_t n, uchar2* __restrict__ mm, float2* __restrict__ out)
{
    const uint64_t total = (uint64_t)n * n;
    for (uint64_t node = (uint64_t)blockIdx.x * 256 + threadIdx.x; node < total; node += (uint64_t)gridDim.x * 256) {
        const uint32_t ix = (uint32_t)(node % n), iz = (uint32_t)(node / n);
        const uchar2* c = child + (size_t)(2u * iz) * (2u * n) + 2u * ix;
        const uchar2 a0 = c[0], a1 = c[1], a2 = c[2u * n], a3 = c[2u * n + 1];
        const int mn = min(min((int)a0.x, (int)a1.x), min((int)a2.x, (int)a3.x)), mx = max(max((int)a0.y, (int)a1.y), max((int)a2.y, (int)a3.y));
        mm[node] = make_uchar2((unsigned char)mn, (unsigned char)mx);
        out[node] = finish_minmax(mn, mx);
    }
}

// texel origin of a surface's root if the mip-style path applies to it, else false
static bool dyadic_root(const vr_terrain* t, const float loc[3], int* rx0, int* ry0, int* leaf_texels)
{
    const float S = t->p.surface_size;
    if (t->texel_size[0] != 1.0f || t->texel_size[1] != 1.0f) return false;
    if (!(S >= 1.0f) || S != floorf(S) || ((int)S & ((int)S - 1)) != 0) return false;
    if ((int64_t)t->height.w0 * t->height.h0 > (1 << 24)) return false;       // GetHeightValue indexes in float: exact only up to 2^24
    const float minx = (loc[0] - S / 2) + t->p.world_size / 2, miny = (loc[2] - S / 2) + t->p.world_size / 2;
    if (minx != floorf(minx) || miny != floorf(miny)) return false;
    if (minx < 0.0f || miny < 0.0f || minx + S > (float)t->height.w0 || miny + S > (float)t->height.h0) return false;
    const int s = (int)S >> t->num_lods;
    if (s < 1 || (s << t->num_lods) != (int)S) return false;
    *rx0 = (int)minx; *ry0 = (int)miny; *leaf_texels = s;
    return true;
}

extern "C" VR_API int vr_terrain_update_heights(vr_terrain* t, int enable)
{
    VR_REQUIRE(t != nullptr, "terrain is NULL");
    VR_HIP(hipSetDevice(t->ctx->device));
    for (GeoSet& g : t->sets) g.prepared = false;          // geometry prepared with the old bounds is stale
    if (!enable) { t->height_loaded = false; return VR_OK; }
    const uint64_t nodes = ((((uint64_t)1 << (2 * (t->num_lods + 1))) - 1) / 3);
    const int num_surfaces = t->surfaces_per_side * t->surfaces_per_side;
    if (!t->d_node_heights) VR_HIP(hipMalloc(&t->d_node_heights, nodes * num_surfaces * sizeof(float2)));
    HeightArgs a;
    a.half_w = t->p.surface_size / 2.0f; a.half_h = t->p.surface_size / 2.0f; a.world_size = t->p.world_size;
    a.texel_x = t->texel_size[0]; a.texel_y = t->texel_size[1]; a.tex_w = t->height.w0; a.tex_h = t->height.h0;
    VrKernelScope ks(t->ctx, VR_K_NODE_HEIGHTS);
    for (int surf = 0; surf < num_surfaces; surf++) {
    {   // TerrainPass.cpp:102-108
        const int column = surf % t->surfaces_per_side, row = surf / t->surfaces_per_side;
        const float x = -0.5f * (float)(t->surfaces_per_side - 1) + (float)column, y = -0.5f * (float)(t->surfaces_per_side - 1) + (float)row;
        a.loc[0] = t->p.location[0] + x * t->p.surface_size; a.loc[1] = t->p.location[1] + 0.0f; a.loc[2] = t->p.location[2] + y * t->p.surface_size;
    }
    float2* out = t->d_node_heights + (size_t)surf * nodes;
    int rx0 = 0, ry0 = 0, leaf = 0;
    if (dyadic_root(t, a.loc, &rx0, &ry0, &leaf)) {
        if (!t->d_minmax) VR_HIP(hipMalloc(&t->d_minmax, nodes * sizeof(uchar2)));
        for (int d = t->num_lods; d >= 0; d--) {
            const uint32_t n = 1u << d;
            const uint64_t base = ((((uint64_t)1 << (2 * d)) - 1) / 3), cells = (uint64_t)n * n, blocks = (cells + 255) / 256;
            const unsigned grid = (unsigned)(blocks > 16384 ? 16384 : blocks);
            if (d == t->num_lods)
                hipLaunchKernelGGL(k_minmax_leaf, dim3(grid), dim3(256), 0, t->ctx->stream, (const uint8_t*)t->d_height, t->height.w0, rx0, ry0, n, leaf,
                                   t->d_minmax + base, out + base);
            else
                hipLaunchKernelGGL(k_minmax_up, dim3(grid), dim3(256), 0, t->ctx->stream,
                                   (const uchar2*)(t->d_minmax + ((((uint64_t)1 << (2 * (d + 1))) - 1) / 3)), n, t->d_minmax + base, out + base);
        }
        continue;
    }
    for (int d = 0; d <= t->num_lods; d++) {
        a.depth = d;
        const uint64_t n = (uint64_t)1 << (2 * d);
        // texels per node at this depth (approximate; only chooses the kernel)
        const double per_node = ((double)t->p.surface_size * t->texel_size[0]) * ((double)t->p.surface_size * t->texel_size[1]) / (double)n;
        if (per_node >= 512.0 && n <= (1u << 20))
            hipLaunchKernelGGL(k_node_heights_block, dim3((unsigned)n), dim3(256), 0, t->ctx->stream, a, t->d_height, out);
        else {
            const uint64_t blocks = (n + 255) / 256;
            hipLaunchKernelGGL(k_node_heights_thread, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, t->ctx->stream, a,
                               t->d_height, out);
        }
    }
    }
    VR_HIP(hipGetLastError());
    // NodeSelect runs on the geometry stream: make it see these results
    VR_HIP(hipEventRecord(t->ev_main_dep, t->ctx->stream));
    for (GeoSet& g : t->sets) g.main_dep_pending = true;
    t->height_loaded = true;
    return VR_OK;
}

extern "C" VR_API int vr_terrain_download_node_heights(vr_terrain* t, uint32_t first, uint32_t count, float* out)
{
    VR_REQUIRE(t && out && t->d_node_heights, "node heights have not been computed");
    const uint64_t nodes = ((((uint64_t)1 << (2 * (t->num_lods + 1))) - 1) / 3) * (uint64_t)(t->surfaces_per_side * t->surfaces_per_side);
    VR_REQUIRE((uint64_t)first + count <= nodes, "node id range out of bounds");
    VR_HIP(hipSetDevice(t->ctx->device));
    VR_HIP(hipStreamSynchronize(t->ctx->stream));
    VR_HIP(hipMemcpy(out, t->d_node_heights + first, (size_t)count * sizeof(float2), hipMemcpyDeviceToHost));
    return VR_OK;
}

// ---- terrain object -------------------------------------------------------------------
// The part of a terrain's per-frame scratch that scales with the number of nodes a frame selects (vr_terrain::cap_instances).
static void free_scratch(vr_terrain* t)
{
    for (GeoSet& g : t->sets) {
        (void)hipFree(g.d_verts); (void)hipFree(g.d_rect); (void)hipFree(g.d_recs); (void)hipFree(g.d_hard_first); (void)hipFree(g.d_bin_entries);
        g.d_verts = nullptr; g.d_rect = nullptr; g.d_recs = nullptr; g.d_hard_first = nullptr; g.d_bin_entries = nullptr;
    }
}
static int alloc_scratch(vr_terrain* t, int cap)
{
    free_scratch(t);
    const size_t mi = (size_t)cap, fixed = (size_t)t->p.max_instances * (sizeof(uint32_t) + sizeof(vr_instance)) + 64 * sizeof(uint32_t)
                    + kSelScratchWords * sizeof(uint32_t) + (size_t)t->hard_cap * (sizeof(uint32_t) + 4 * sizeof(HardTriRec));
    t->bytes_scratch = (uint64_t)fixed * kGeoSets;
    // (triangle, tile) pairs: an 8K frame of ~300 nodes has ~0.3 M, a 1080p frame ~0.6 M; 1 M per 1024 nodes and never fewer
    // - or what the frames seen so far asked for (a large target on 32-pixel tiles: every triangle lands in more bins)
    t->bin_capacity = ((size_t)1 << 20) * ((mi + 1023) / 1024);
    if (const char* e = getenv("VR_SCRATCH_INITIAL_BINS")) { const long v = atol(e); if (v >= 1024) t->bin_capacity = (size_t)v; }   // (tests: force the growth path)
    if (t->bin_want > t->bin_capacity) t->bin_capacity = t->bin_want;
#define VR_ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc(&(ptr), (bytes)); if (e_ != hipSuccess) { \
        vr_set_error("hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(e_)); return VR_ERR_OUT_OF_MEMORY; } \
        t->bytes_scratch += (uint64_t)(bytes); } while (0)
    for (GeoSet& g : t->sets) {
        VR_ALLOC(g.d_verts, (mi * kVertsPerInst + t->extra_vert_cap) * sizeof(DevVert));
        VR_ALLOC(g.d_rect, mi * kTrisPerInst * sizeof(uint64_t));
        VR_ALLOC(g.d_recs, (mi * kTrisPerInst + (size_t)t->hard_cap * 4) * kRecGroups * sizeof(uint4));
        VR_ALLOC(g.d_hard_first, mi * kTrisPerInst * sizeof(uint32_t));
        VR_ALLOC(g.d_bin_entries, t->bin_capacity * sizeof(TileEntry));
        g.prepared = false;                   // geometry built for the old buffers is gone (a selection lives in the fixed part: kept)
    }
#undef VR_ALLOC
    t->cap_instances = cap;
    return VR_OK;
}

// A target of `tiles` raster tiles is about to be drawn: make room for ~8 bin entries per tile up front (measured: 9.3 per tile at
// 8K, 5.9 at 15360x8640, 4.7 at 16384^2 - the terrain's triangles grow with the frame, the tiles do not), so that the first frame
// on a very large target does not have to overflow before the bins grow.
int vr_terrain_reserve_bins(vr_terrain* t, size_t tiles)
{
    const size_t est = tiles * 8;
    if (est <= t->bin_capacity) return VR_OK;
    size_t b = t->bin_capacity ? t->bin_capacity : ((size_t)1 << 20);
    while (b < est) b *= 2;
    t->bin_want = b;
    VR_HIP(hipSetDevice(t->ctx->device));
    for (hipStream_t gs : t->geo_streams) VR_HIP(hipStreamSynchronize(gs));
    VR_HIP(hipStreamSynchronize(t->ctx->stream));
    return alloc_scratch(t, t->cap_instances);
}

int vr_terrain_poll(vr_terrain* t, bool report)
{
    uint32_t seen = 0;
    for (int i = 0; i < kGeoSets; i++) {
        GeoSet& g = t->sets[i];
        if (!g.status_pending || !g.geo_recorded || hipEventQuery(g.ev_geo_done) != hipSuccess) continue;
        g.status_pending = false;
        const volatile uint32_t* st = t->h_status + i * 8;
        const uint32_t flags = st[1], wanted = st[6];
        if (wanted > seen) seen = wanted;
        if ((size_t)st[5] > t->bin_high_water) t->bin_high_water = (size_t)st[5];     // (triangle, tile) pairs the frame wanted
        if (flags & 1u) { t->sticky_error = VR_ERR_TOO_MANY_INSTANCES; t->sticky_count = wanted; }
        else if (flags & 6u) { if (!t->sticky_error) t->sticky_error = VR_ERR_OVERFLOW; t->sticky_count = wanted; }
    }
    if (seen > t->high_water) t->high_water = seen;
    // grow before a frame can outgrow the scratch: twice the largest count seen, once that passes half the capacity
    const int max_i = t->p.max_instances;
    const bool grow_nodes = t->cap_instances < max_i && (size_t)t->high_water * 2 > (size_t)t->cap_instances;
    const bool grow_bins = t->bin_high_water * 2 > t->bin_capacity;            // (the same rule for the bins: twice what was seen)
    if (grow_nodes || grow_bins) {
        int want = t->cap_instances;
        while (want < max_i && (size_t)t->high_water * 2 > (size_t)want) want *= 2;
        if (want > max_i) want = max_i;
        if (grow_bins) { size_t b = t->bin_capacity; while (b < t->bin_high_water * 2) b *= 2; t->bin_want = b; }
        VR_HIP(hipSetDevice(t->ctx->device));
        for (hipStream_t gs : t->geo_streams) VR_HIP(hipStreamSynchronize(gs));
        VR_HIP(hipStreamSynchronize(t->ctx->stream));
        const int had = t->cap_instances;
        int rc = alloc_scratch(t, want);
        if (rc) {
            // the larger scratch does not fit: back to the old size (frames that need more stay truncated and reported)
            t->bin_want = 0;
            const int rc2 = alloc_scratch(t, had);
            t->high_water = 0; t->bin_high_water = 0;
            return rc2 ? rc2 : rc;
        }
    }
    if (report && t->sticky_error) {
        const int e = t->sticky_error;
        t->sticky_error = VR_OK;
        if (e == VR_ERR_TOO_MANY_INSTANCES) vr_set_error("an earlier frame selected %u nodes, more than max_instances (TerrainPass.cpp:238 assert); it was drawn without the excess", t->sticky_count);
        else vr_set_error("an earlier frame (%u nodes) overflowed a work list or the scratch: triangles or nodes were dropped (the scratch now holds %d nodes)", t->sticky_count, t->cap_instances);
        return e;
    }
    return VR_OK;
}

static int ilog2_floor(float x)
{
    uint32_t b; memcpy(&b, &x, 4);
    return (int)((b >> 23) & 255u) - 127;
}

extern "C" VR_API int vr_terrain_create(vr_context* ctx, const vr_terrain_params* params, const uint8_t* height_r8,
                                         int32_t hm_w, int32_t hm_h, const uint8_t* albedo, int32_t al_w, int32_t al_h,
                                         vr_terrain** out)
{
    VR_REQUIRE(ctx && params && out, "NULL argument");
    VR_REQUIRE(height_r8 && albedo, "heightmap/albedo missing (reference logs 'Heightmap texture data missing', QuadTree.cpp:39)");
    VR_REQUIRE(params->grid_size == kGrid, "grid_size must be 32");
    VR_REQUIRE(params->max_instances > 0 && params->max_instances <= 4096, "max_instances must be in 1..4096");
    VR_REQUIRE(params->surface_size >= 1.0f && params->world_size >= params->surface_size, "bad surface/world size");
    // static_assert(WORLD_SIZE >= SURFACE_SIZE && WORLD_SIZE % SURFACE_SIZE == 0) (TerrainPass.h:30)
    const int per_side = (int)params->world_size / (int)params->surface_size;
    VR_REQUIRE((float)((int)params->world_size) == params->world_size && (float)((int)params->surface_size) == params->surface_size
               && (int)params->world_size % (int)params->surface_size == 0, "world_size must be an integer multiple of surface_size");
    VR_REQUIRE(per_side * per_side <= 64, "at most 64 surfaces");
    VR_REQUIRE(params->min_lod_distance > 0.0f && params->morph_start > 0.0f && params->morph_start < 1.0f, "bad LOD parameters");
    VR_HIP(hipSetDevice(ctx->device));
    vr_terrain* t = new vr_terrain();
    struct Guard { vr_terrain*& t; ~Guard() { if (t) vr_terrain_destroy(t); } } guard{ t };   // an early return frees what exists so far
    t->ctx = ctx; t->p = *params; t->surfaces_per_side = per_side;
    // QuadTree::InitLodRanges (QuadTree.cpp:234-241), QuadTree::Init numLods (QuadTree.cpp:22)
    for (int i = 0; i < VR_MAX_LODS; i++) t->lod_ranges[i] = params->min_lod_distance * powf(2.0f, (float)i);
    int l2 = ilog2_floor(params->surface_size);
    t->num_lods = (VR_MAX_LODS - 1) < l2 ? (VR_MAX_LODS - 1) : l2;
    t->texel_size[0] = (float)hm_w / params->world_size; t->texel_size[1] = (float)hm_h / params->world_size;   // QuadTree.cpp:29
    int rc;
    if ((rc = vr_tex_upload_and_mip(ctx, height_r8, hm_w, hm_h, 1, &t->height, &t->d_height, &t->bytes_textures))) return rc;
    if ((rc = vr_tex_upload_and_mip(ctx, albedo, al_w, al_h, 4, &t->albedo, &t->d_albedo, &t->bytes_textures))) return rc;
    const size_t mi = (size_t)params->max_instances;
    t->extra_vert_cap = 1u << 16; t->hard_cap = 1u << 15;
    for (GeoSet& g : t->sets) {         // what does not depend on the scratch's capacity
        hipError_t e_ = hipMalloc(&g.d_node_ids, mi * sizeof(uint32_t));
        if (e_ == hipSuccess) e_ = hipMalloc(&g.d_instances, mi * sizeof(vr_instance));
        if (e_ == hipSuccess) e_ = hipMalloc(&g.d_counters, 64 * sizeof(uint32_t));
        if (e_ == hipSuccess) e_ = hipMalloc(&g.d_sel_scratch, kSelScratchWords * sizeof(uint32_t));
        if (e_ == hipSuccess) e_ = hipMalloc(&g.d_hard_list, (size_t)t->hard_cap * sizeof(uint32_t));
        if (e_ == hipSuccess) e_ = hipMalloc(&g.d_hard_tris, (size_t)t->hard_cap * 4 * sizeof(HardTriRec));
        if (e_ != hipSuccess) { vr_set_error("hipMalloc failed: %s", hipGetErrorString(e_)); return VR_ERR_OUT_OF_MEMORY; }
    }
    VR_HIP(hipHostMalloc((void**)&t->h_status, kGeoSets * 8 * sizeof(uint32_t), hipHostMallocMapped));
    memset(t->h_status, 0, kGeoSets * 8 * sizeof(uint32_t));
    VR_HIP(hipHostGetDevicePointer((void**)&t->d_status, t->h_status, 0));
    {   // what the scratch holds to begin with: 1024 nodes (a 2048^2 world selects 300-600), or everything (VR_OPT_SCRATCH_WORST_CASE);
        // VR_SCRATCH_INITIAL_NODES (environment) lets a test start smaller and watch it grow
        int initial = 1024;
        if (const char* e = getenv("VR_SCRATCH_INITIAL_NODES")) { const int v = atoi(e); if (v >= 1) initial = v; }
        if (ctx->scratch_worst_case || initial > (int)mi) initial = (int)mi;
        if ((rc = alloc_scratch(t, initial))) return rc;
    }
    for (GeoSet& g : t->sets) {
        VR_HIP(hipMemsetAsync(g.d_counters, 0, 64 * sizeof(uint32_t), ctx->stream));
        VR_HIP(hipEventCreateWithFlags(&g.ev_geo_done, hipEventDisableTiming));
        VR_HIP(hipEventCreateWithFlags(&g.ev_raster_done, hipEventDisableTiming));
        VR_HIP(hipEventCreateWithFlags(&g.ev_sel_read, hipEventDisableTiming));
    }
    VR_HIP(hipStreamSynchronize(ctx->stream));
    {   // lowest priority: geometry fills in around the tile / lighting passes (highest priority measured the same:
        // what delays these small kernels is LDS space on the CUs, not queue arbitration)
        int least = 0, greatest = 0;
        VR_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        for (hipStream_t& gs : t->geo_streams) VR_HIP(hipStreamCreateWithPriority(&gs, hipStreamNonBlocking, least));
        for (GeoSet& g : t->sets) g.stream = t->geo_streams[0];
    }
    VR_HIP(hipEventCreateWithFlags(&t->ev_main_dep, hipEventDisableTiming));
    VR_HIP(hipEventCreateWithFlags(&t->ev_sel_copy, hipEventDisableTiming));
    VR_HIP(hipEventCreateWithFlags(&t->ev_raster_begin, hipEventDisableTiming));
    *out = t;
    t = nullptr;                 // released to the caller
    return VR_OK;
}

extern "C" VR_API void vr_terrain_destroy(vr_terrain* t)
{
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    for (hipStream_t gs : t->geo_streams) if (gs) { (void)hipStreamSynchronize(gs); (void)hipStreamDestroy(gs); }
    if (t->ev_main_dep) (void)hipEventDestroy(t->ev_main_dep);
    if (t->ev_sel_copy) (void)hipEventDestroy(t->ev_sel_copy);
    if (t->ev_raster_begin) (void)hipEventDestroy(t->ev_raster_begin);
    for (GeoSet& g : t->sets) {
        if (g.ev_geo_done) (void)hipEventDestroy(g.ev_geo_done);
        if (g.ev_raster_done) (void)hipEventDestroy(g.ev_raster_done);
        if (g.ev_sel_read) (void)hipEventDestroy(g.ev_sel_read);
        (void)hipFree(g.d_node_ids); (void)hipFree(g.d_instances); (void)hipFree(g.d_counters); (void)hipFree(g.d_sel_scratch);
        (void)hipFree(g.d_hard_list); (void)hipFree(g.d_hard_tris);
        (void)hipFree(g.d_tile_count); (void)hipFree(g.d_tile_offset); (void)hipFree(g.d_tile_cursor); (void)hipFree(g.d_tile_order);
    }
    free_scratch(t);
    if (t->h_status) (void)hipHostFree(t->h_status);
    (void)hipFree(t->d_height); (void)hipFree(t->d_albedo); (void)hipFree(t->d_node_heights); (void)hipFree(t->d_minmax);
    delete t;
}

extern "C" VR_API int vr_terrain_num_lods(const vr_terrain* t) { return t ? t->num_lods : -1; }
extern "C" VR_API int vr_terrain_lod_ranges(const vr_terrain* t, float out[VR_MAX_LODS])
{
    VR_REQUIRE(t && out, "NULL argument");
    for (int i = 0; i < VR_MAX_LODS; i++) out[i] = t->lod_ranges[i];
    return VR_OK;
}

extern "C" VR_API int vr_terrain_download_mip(vr_terrain* t, int which, int level, void* host, size_t bytes,
                                               int32_t* w, int32_t* h, int32_t* levels)
{
    VR_REQUIRE(t && (which == 0 || which == 1), "bad arguments");
    const DevTex& tx = which == 0 ? t->height : t->albedo;
    if (levels) *levels = tx.levels;
    VR_REQUIRE(level >= 0 && level < tx.levels, "mip level out of range");
    const int lw = (tx.w0 >> level) > 1 ? (tx.w0 >> level) : 1, lh = (tx.h0 >> level) > 1 ? (tx.h0 >> level) : 1;
    if (w) *w = lw;
    if (h) *h = lh;
    if (!host) return VR_OK;
    const size_t nb = (size_t)lw * lh * (which == 0 ? 1 : 4);
    VR_REQUIRE(bytes == nb, "byte count does not match the level size");
    VR_HIP(hipSetDevice(t->ctx->device));
    uint32_t off[kMaxLevels];
    VR_HIP(hipMemcpy(off, tx.off, sizeof(off), hipMemcpyDeviceToHost));
    VR_HIP(hipMemcpy(host, tx.base + off[level], nb, hipMemcpyDeviceToHost));
    return VR_OK;
}

static int read_counters(vr_terrain* t, uint32_t* count, bool selection_only = false)
{
    uint32_t c[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    VR_HIP(hipStreamSynchronize(t->sets[t->cur].stream)); // selection and bins are produced on the set's geometry stream
    VR_HIP(hipMemcpyAsync(c, t->sets[t->cur].d_counters, sizeof(c), hipMemcpyDeviceToHost, t->ctx->stream));
    VR_HIP(hipStreamSynchronize(t->ctx->stream));
    if (count) *count = selection_only ? c[7] : c[0];           // NodeSelect's list is complete up to max_instances whatever the scratch holds
    t->sets[t->cur].status_pending = false;                     // read here; not reported a second time
    if (c[1] & 1u) { vr_set_error("more than max_instances nodes selected (TerrainPass.cpp:238 assert)"); return VR_ERR_TOO_MANY_INSTANCES; }
    if ((size_t)c[5] > t->bin_high_water) t->bin_high_water = (size_t)c[5];
    if (c[1] & 2u) {
        // a full work list: if it was the bins, they grow now (twice what this frame wanted) and rendering the frame again is complete
        (void)vr_terrain_poll(t, false);
        vr_set_error("internal work list overflowed (the frame wanted %u bin entries; the bins now hold %zu)", c[5], t->bin_capacity);
        return VR_ERR_OVERFLOW;
    }
    if (c[1] & 4u) {
        // the frame wanted more nodes than the scratch held: grow now, so that rendering the frame again is complete
        if (c[6] > t->high_water) t->high_water = c[6];
        (void)vr_terrain_poll(t, false);
        if (selection_only) return VR_OK;                       // (a selection alone needs no scratch)
        t->sticky_error = VR_OK;
        vr_set_error("the frame selected more nodes than the scratch held (drawn without the excess); the scratch has been grown to %d nodes - render again",
                     t->cap_instances);
        return VR_ERR_OVERFLOW;
    }
    return VR_OK;
}

// The set the next select / geometry build goes to: never the current one (a tile pass may be reading it and lock_view
// copies its selection); among the others one that holds no prepared frame, else the one prepared longest ago.
int vr_terrain_pick_set(vr_terrain* t)
{
    int best = -1;
    for (int i = 0; i < kGeoSets; i++) {
        if (i == t->cur) continue;
        if (!t->sets[i].prepared) return i;
        if (best < 0 || t->sets[i].prep_serial < t->sets[best].prep_serial) best = i;
    }
    return best;
}

extern "C" VR_API int vr_terrain_select(vr_terrain* t, const vr_view* view, float max_height, uint32_t* node_ids,
                                         vr_instance* instances, uint32_t* count)
{
    VR_REQUIRE(t && view, "NULL argument");
    VR_HIP(hipSetDevice(t->ctx->device));
    // (grows the scratch if due; a sticky condition of an earlier frame is vr_terrain_render's to report)
    { const int prc = vr_terrain_poll(t, false); if (prc) return prc; }
    // the selection buffers are consumed by geometry-stream kernels: order this launch behind them and
    // behind whatever the context's stream did to the terrain (heights)
    const int gi = vr_terrain_pick_set(t);                         // a set no tile pass in flight is reading
    GeoSet& g = t->sets[gi];
    g.stream = t->geo_streams[t->geo_turn++ & 1u];
    if (g.main_dep_pending) { VR_HIP(hipStreamWaitEvent(g.stream, t->ev_main_dep, 0)); g.main_dep_pending = false; }
    // the set's previous chain may have run on the other geometry stream and may never have been rastered
    if (g.geo_recorded) VR_HIP(hipStreamWaitEvent(g.stream, g.ev_geo_done, 0));
    if (g.sel_read_pending) { VR_HIP(hipStreamWaitEvent(g.stream, g.ev_sel_read, 0)); g.sel_read_pending = false; }
    if (g.raster_recorded && (g.raster_done_epoch == 0 || g.raster_done_epoch == t->ctx->ev_epoch)) VR_HIP(hipStreamWaitEvent(g.stream, g.raster_done, 0));
    g.prepared = false;
    int rc = vr_select_launch(t, g, view, max_height, g.stream);
    if (rc) return rc;
    VR_HIP(hipEventRecord(g.ev_geo_done, g.stream));           // this select is the set's latest writer
    g.geo_recorded = true;
    t->cur = gi;
    if (!node_ids && !instances && !count) return VR_OK;       // stays asynchronous
    uint32_t n = 0;
    rc = read_counters(t, &n, true);
    if (count) *count = n;
    if (node_ids && n) VR_HIP(hipMemcpy(node_ids, g.d_node_ids, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (instances && n) VR_HIP(hipMemcpy(instances, g.d_instances, n * sizeof(vr_instance), hipMemcpyDeviceToHost));
    return rc;
}

extern "C" VR_API int vr_terrain_num_chunks(vr_terrain* t, uint32_t* count)
{
    VR_REQUIRE(t && count, "NULL argument");
    VR_HIP(hipSetDevice(t->ctx->device));
    if (!t->sets[t->cur].have_selection) { *count = 0; return VR_OK; }
    return read_counters(t, count);
}

extern "C" VR_API int vr_debug_render_stats(vr_terrain* t, uint32_t out[8])
{
    VR_REQUIRE(t && out, "NULL argument");
    VR_HIP(hipSetDevice(t->ctx->device));
    VR_HIP(hipStreamSynchronize(t->sets[t->cur].stream));
    VR_HIP(hipStreamSynchronize(t->ctx->stream));
    const GeoSet& g = t->sets[t->cur];
    VR_HIP(hipMemcpy(out, g.d_counters, 6 * sizeof(uint32_t), hipMemcpyDeviceToHost));
#ifdef VR_SELECT_PROFILE
    { unsigned long long ts[4]; VR_HIP(hipMemcpy(ts, g.d_sel_scratch + 2 * (kFrontierCap - kFrontLds) + kSelectedCap - 8, sizeof(ts), hipMemcpyDeviceToHost));
      fprintf(stderr, "k_select cycles: levels %llu  rank %llu  write-out %llu\n", ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2]); }
#endif
    out[6] = 0; out[7] = 0;
    if (g.scratch_tiles > 0 && g.d_tile_cursor) {
        std::vector<uint32_t> c((size_t)g.scratch_tiles), o((size_t)g.scratch_tiles);
        VR_HIP(hipMemcpy(c.data(), g.d_tile_cursor, c.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        VR_HIP(hipMemcpy(o.data(), g.d_tile_offset, o.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < c.size(); i++) { const uint32_t v = c[i] - o[i]; if (v > out[6]) out[6] = v; if (v) out[7]++; }
    }
    return VR_OK;
}

extern "C" VR_API int vr_debug_download_vertices(vr_terrain* t, uint32_t first, uint32_t count, float* out)
{
    VR_REQUIRE(t && out, "NULL argument");
    VR_HIP(hipSetDevice(t->ctx->device));
    VR_HIP(hipStreamSynchronize(t->sets[t->cur].stream));
    VR_HIP(hipStreamSynchronize(t->ctx->stream));
    const GeoSet& g = t->sets[t->cur];
    uint32_t n = 0;
    VR_HIP(hipMemcpy(&n, g.d_counters, sizeof(uint32_t), hipMemcpyDeviceToHost));
    VR_REQUIRE((uint64_t)first + count <= (uint64_t)n * kVertsPerInst, "vertex range exceeds the last draw's instances");
    std::vector<DevVert> v(count);
    if (count) VR_HIP(hipMemcpy(v.data(), g.d_verts + first, (size_t)count * sizeof(DevVert), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < count; i++) {
        out[i * 6 + 0] = v[i].cx; out[i * 6 + 1] = v[i].cy; out[i * 6 + 2] = v[i].cz; out[i * 6 + 3] = v[i].cw;
        out[i * 6 + 4] = v[i].wx; out[i * 6 + 5] = v[i].wz;
    }
    return VR_OK;
}

extern "C" VR_API int vr_terrain_memory_bytes(const vr_terrain* t, uint64_t out[4])
{
    VR_REQUIRE(t && out, "NULL argument");
    uint64_t tiles = 0;
    for (const GeoSet& g : t->sets) tiles += (uint64_t)g.scratch_tiles * 4u * sizeof(uint32_t);
    uint64_t heights = 0;
    if (t->d_node_heights) {
        const uint64_t nodes = ((((uint64_t)1 << (2 * (t->num_lods + 1))) - 1) / 3);
        heights = nodes * (uint64_t)(t->surfaces_per_side * t->surfaces_per_side) * sizeof(float2) + (t->d_minmax ? nodes * sizeof(uchar2) : 0);
    }
    out[0] = t->bytes_textures; out[1] = t->bytes_scratch + tiles; out[2] = heights; out[3] = out[0] + out[1] + out[2];
    return VR_OK;
}
