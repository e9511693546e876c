// Internal declarations shared by the translation units of libvrterrain.so.
// gfx950 (MI355X) only; wave = 64 lanes everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "../../include/vrterrain.h"

// ---- error plumbing -------------------------------------------------------------
void vr_set_error(const char* fmt, ...);
#define VR_HIP(expr)                                                                  \
    do {                                                                              \
        hipError_t e__ = (expr);                                                      \
        if (e__ != hipSuccess) {                                                      \
            vr_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return e__ == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP;    \
        }                                                                             \
    } while (0)
#define VR_REQUIRE(cond, msg)                                                         \
    do { if (!(cond)) { vr_set_error("%s:%d: %s", __FILE__, __LINE__, msg); return VR_ERR_INVALID_ARGUMENT; } } while (0)

// ---- geometry constants -----------------------------------------------------------
constexpr int kGrid = 32;                    // GRID_SIZE (TerrainPass.h:28)
constexpr int kSide = kGrid + 1;             // 33 vertices per side
constexpr int kVertsPerInst = kSide * kSide; // 1089
constexpr int kTrisPerInst = kGrid * kGrid * 2; // 2048
constexpr int kRecGroups = 8;                // 16-byte groups per triangle record (vr_raster.hip: write_tri_rec)
constexpr int kRasterTile = 64;              // raster/bin tile (pixels) of large frames; also the upper bound of either size
// Tile edge of the raster / bin grid for a w x h target: 64, or 32 when this rank draws fewer than kTile64Min 64-pixel
// tiles.  A 64-pixel tile amortises the per-workgroup work (tables, bin header, barrier) over four times the pixels and
// sweeps a triangle of the 8K frame (30-60 pixels) in one or two tiles instead of four; 32-pixel tiles are four times as
// many workgroups (launch waves, tail) and five instead of four of them fit a CU.  Measured again in the second session of
// round 3 with that round's tile pass, frame time of the bench, 64- vs 32-pixel tiles: 4K 0.248 vs 0.231 ms, 5120x2880
// 0.312 vs 0.287, 6400x3600 0.414 vs 0.404, 7040x3960 0.475 vs 0.475, 8K 0.541 vs 0.552; a rank of an N-way split of the
// 8K frame (tile pass alone in brackets): N = 2 0.400 vs 0.378 ms, N = 4 0.232 vs 0.210 (149 vs 124 us), N = 8 0.154 vs 0.130
// (profiles/r03_tile_size_thresholds.txt).  Round 2's thresholds - 2560 tiles for the whole frame, 1536 per rank -
// came from a tile pass that was 40 % slower per pixel and cost N = 2 and N = 4 5-10 %.  With the 32-pixel variants at six waves
// per SIMD (vr_raster.hip) they win up to ~9.6K x 5.4K: 8K 0.544 vs 0.527 ms, 7040x3960 0.472 vs 0.455, 9600x5400 0.796 vs 0.792,
// 12288x6912 1.250 vs 1.265 - hence 13000.
// Round 4: with the plane-state tracking (region states exist on 32-pixel tiles only: vr_gbuffer) and the pre-set-up bin entries
// 32-pixel tiles win at every size that was measured - 11520x6480 0.902 vs 0.983 ms, 15360x8640 1.589 vs 1.694 (the lighting pass
// reads what the states let it skip: 324 vs 384 us and 575 vs 683 us; the tile pass itself 560 vs 577 and 996 vs 989 us) - so the
// rule now picks 32 for any target the library accepts; 64-pixel tiles remain behind VR_OPT_RASTER_TILE.
constexpr long kTile64Min = 1L << 40;
// `force`: 0 = the rule above, 5 / 6 = 32- / 64-pixel tiles whatever the size (vr_context_set_option(VR_OPT_RASTER_TILE): tests
// compare both variants with the oracle at sizes the oracle finishes in seconds; a host may pin the choice)
inline int vr_raster_tile_shift(int w, int h, int world = 1, int force = 0)
{
    if (force == 5 || force == 6) return force;
    const long tiles64 = (long)((w + 63) / 64) * (long)((h + 63) / 64);
    return tiles64 / (world > 1 ? world : 1) < kTile64Min ? 5 : 6;
}
constexpr int kMaxLights = 16;               // terrain_cb.h:15 / Donut DEFERRED_MAX_LIGHTS
constexpr int kMaxLevels = 16;
constexpr float kGuardBand = 100.0f;
// linear -> sRGB8 start table: one bucket per (exponent, top 7 mantissa bits) from 2^-13 to 1.0
constexpr int kEncTabBase = (127 - 13) << 7;
constexpr int kEncTabSize = (13 << 7) + 1;

// Mip chain in device memory, passed to kernels by value.
struct DevTex {
    const uint8_t* base;        // level 0 first, then the coarser levels
    const uint32_t* off;        // device table: byte offset of each level (kMaxLevels entries)
    int levels;
    int w0, h0;                 // level l is max(1, w0 >> l) x max(1, h0 >> l)
    uint32_t chain_bytes;       // bytes of the mip chain at `base` (bound of the buffer resource the tile pass reads texels through)
    // R8 textures only: per level a (w+2) x (h+2) table of bilinear footprints.  Entry (ix, iy)
    // packs the four clamp-addressed texels (x0,y0) (x1,y0) (x0,y1) (x1,y1) for floor(x) = ix-1,
    // floor(y) = iy-1 into one dword, so a bilinear tap is ONE load instead of four byte loads.
    const uint32_t* quad;
    const uint32_t* qoff;       // device table: dword offset of each level's quad table
    uint32_t quad_bytes, pad;   // bytes of all quad tables at `quad`
    // The same footprints decoded: per entry (t00, t10 - t00, t01, t11 - t01) as floats (t = byte / 255), same indexing.
    // A height tap is then one 16-byte load and 7 float operations - no byte extraction, no conversion table.
    const float4* quadf;
    // SRGBA8 textures only: the chain decoded to linear floats, one float4 per texel (r, g, b, 0); level l starts at byte
    // 4 * off[l].  An albedo texel is then one aligned 16-byte load with nothing to extract or look up, at (index << 4).
    const float* rgbf;
    // Both kinds, for the tile pass's fast variant (heightmap and albedo of one size): per level a (w+3) x (h+3) table of
    // 16-byte entries - R8: the footprint (t00, t10 - t00, t01, t11 - t01) whose floor is (ix-1, iy-1); SRGBA8: the decoded
    // texel (r, g, b, 0) at (ix-1, iy-1) - clamp-addressed, so that a bilinear footprint needs no clamp: texel (x, y),
    // x in [-1, w], y in [-1, h], is entry (x+1, y+1) and its right / lower neighbours are +16 / +row bytes.  Textures of
    // one size get identical layouts, so ONE set of byte offsets addresses the height taps and the albedo footprint.
    // fast_lv[l] = { byte offset of entry (1, 1), row bytes, (float)w, (float)h } (what a pixel needs of level l).
    const float4* fast;
    const uint4* fast_lv;
    uint32_t fast_bytes, pad2;
};

// Per-light constants the deferred kernel reads (host precomputes the half-angle terms).
struct DevLight {
    float dir[3];  int type;
    float pos[3];  float inv_range;
    float color[3]; float intensity;
    float cosH, sinH, tanH, radius;   // radius > 0: spherical source (half angle per pixel)
    float inner_angle, outer_angle, pad0, pad1;   // spot cone (radians)
};

// Transformed vertex: clip position, world xz and the snapped screen vertex.
struct DevVert {
    float cx, cy, cz, cw;
    float wx, wz;
    int32_t X, Y;      // 24.8 fixed point, valid when cw > 0
    float z, iw;
    uint32_t pad0, pad1;
};
static_assert(sizeof(DevVert) == 48, "DevVert layout");

// Explicit triangle (output of the clipper for triangles that cross the near plane
// or leave the guard band): indices into the vertex array's extra region, the
// draw-order key and the raster-tile rectangle it touches.
struct HardTriRec { uint32_t v0, v1, v2, order_key; uint32_t rect_lo, rect_hi, pad0, pad1; };
static_assert(sizeof(HardTriRec) == 32, "HardTriRec layout");

// Bin entry (round 4): a triangle as ONE raster tile sees it, set up once by k_fill - per (triangle, tile) pair, one lane each,
// full waves - instead of by every tile-pass workgroup from the triangle's record: edge functions at the centre of the tile's
// pixel (0, 0) and their steps per pixel, the triangle's pixel box inside the tile, the depth plane with the tile's origin
// relative to the plane's anchor, and the draw-order word of the visibility buffer.  64 bytes = four 16-byte groups, so that
// a sparse bin's entries can be read with scalar loads (the entry index is wave-uniform) straight into SGPRs: no per-lane
// record fetch, no per-lane set-up, no v_readlane broadcast in the tile pass.  (With the resolve's planes in the
// entry as well - 128 bytes - the tile pass alone gained 2.5 %, and k_fill, which runs beside it, cost the frame 2 %: not kept.)
struct TileEntry {
    int32_t e0, e1, e2;                    // E_i at the tile's first pixel centre (bias NOT subtracted); exact when kTeFits32 is set
    int32_t sx0, sy0, sx1, sy1, sx2, sy2;  // steps per pixel: A_i * 256, B_i * 256
    uint32_t box;                          // x0 | y0 << 8 | x1 << 16 | y1 << 24, tile-local, inclusive (clamped to tile and viewport)
    float z0, zx, zy;                      // depth plane about the triangle's anchor pixel
    uint32_t offs;                         // (tile origin - anchor): x & 0xffff | y << 16
    uint32_t order;                        // ~(key + 1): low word of the visibility buffer
    uint32_t flags;                        // bias0..2 (bits 0-2), kTeFits32, kTeValid, kTeSmall
};
static_assert(sizeof(TileEntry) == 64, "TileEntry layout");
constexpr int kTeGroups = 4;                // 16-byte groups per entry
constexpr int kTeSlotBits = 5;              // a bin of up to 32 entries keeps its triangles' planes in LDS: slot = low bits of the visibility word
constexpr uint32_t kTeFits32 = 8u, kTeValid = 16u, kTeSmall = 32u;     // kTeSmall: every |A_i|, |B_i| <= 2^14 (the row hand-out's 24-bit products)

// Screen-tile partition of a w x h frame for one rank of `world`: built once, immutable afterwards.
struct PartTables {
    int w = 0, h = 0, rank = 0, world = 1;
    int tile_shift = 0;                 // raster tile edge the table of raster tiles was built for (part of the cache key)
    int32_t* d_owned_tiles = nullptr;   // owner-tile ids (128x128) owned by this rank
    int32_t* d_tile_slot = nullptr;     // per owner tile: rank * max_owned + local index
    int32_t* d_raster_tiles = nullptr;  // raster-tile ids (64x64 or 32x32) inside owned owner tiles
    int num_owned = 0, max_owned = 0, num_raster_tiles = 0;
};

struct vr_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_sysfence = nullptr;   // the one event WITH a system-scope release (vr_comm.hip: in front of a collective)
    float* d_srgb_lut = nullptr;   // 256 floats: sRGB8 -> linear
    float* d_srgb_thr = nullptr;   // 256 floats: encode thresholds
    float h_srgb_lut[256];
    float h_srgb_thr[256];
    uint8_t* d_enc_tab = nullptr;  // kEncTabSize bytes: sRGB8 code of the smallest float of each bucket
    // partition tables (device), one set per (w,h,rank,world) seen; they live as long as the context, so a
    // kernel queued on any stream can never read a freed or foreign table (a frame alternates between the
    // shadow map's and the main view's geometry every frame)
    std::vector<PartTables*> part_tables;
    bool async_geometry = true;    // VR_OPT_ASYNC_GEOMETRY
    bool scratch_worst_case = false;   // VR_OPT_SCRATCH_WORST_CASE: terrains created from now on size their scratch for max_instances up front
    bool plane_tracking = true;    // VR_OPT_PLANE_TRACKING: the tile pass does not rewrite a G-buffer plane the library knows to be all zero (vr_gbuffer)
    int raster_tile_force = 0;     // VR_OPT_RASTER_TILE: 0 = by size (vr_raster_tile_shift), 5 / 6 = 32- / 64-pixel raster tiles
    // VR_OPT_DISPATCH_EVENTS: the tile pass and the lighting pass are launched with hipExtLaunchKernelGGL, whose start/stop
    // events are stamped by the dispatch itself; the stop events double as the cross-stream dependencies (tile pass done ->
    // its geometry set is free; lighting pass done -> the next frame's geometry may start), so no event-record packets
    // sit between the two big kernels of a frame.  Outside timing runs the events come from this ring.
    bool dispatch_events = true;
    std::vector<hipEvent_t> ev_ring; size_t ev_ring_pos = 0;
    hipEvent_t last_stop = nullptr;     // stop event of the most recent dispatch-stamped launch on `stream`
    // Handles to pooled timing events are only good until the pool is recycled (vr_timing_enable / vr_timing_collect, both
    // of which synchronise the stream first): holders remember the epoch and treat a handle of an older epoch as 'already
    // complete' instead of waiting on an event that may since have been re-recorded for an unrelated kernel.
    uint64_t ev_epoch = 1;
    // light list of vr_deferred_light_tiled
    DevLight* d_lights = nullptr; size_t light_capacity = 0; std::vector<DevLight> h_lights, h_lights_on_device;   // (the list d_lights holds)
    uint32_t* d_macro_scratch = nullptr; size_t macro_scratch_words = 0;   // per macro tile: lights touching its box (k_light_cull's first stage)
    uint32_t* d_light_lists = nullptr; size_t light_list_words = 0;    // per 32x32 light tile: count + light indices (k_light_cull)
    uint32_t* d_flags = nullptr;
    // per-kernel timing (vr_timing_*): event pairs recorded on `stream`
    int timing = 0;                      // 0 off, 1 every kernel (two event records each), 2 only the launches whose events the dispatch stamps
    std::vector<hipEvent_t> ev_pool;     // reusable events
    std::vector<hipEvent_t> ev_begin, ev_end;
    std::vector<int> ev_id;
};

// Records a begin/end event pair around one kernel launch when timing is enabled.
struct VrKernelScope {
    vr_context* c; hipEvent_t e0 = nullptr, e1 = nullptr; hipStream_t st = nullptr; bool attach = false; int id = 0; bool committed = false;
    bool pooled = false;            // the pair came from the timing pool (else from the context's ring: dependencies only)
    VrKernelScope(vr_context* ctx, int id);                       // on the context's stream
    VrKernelScope(vr_context* ctx, int id, hipStream_t stream);   // on another stream of the same device
    // attach = true: the events are not recorded as separate stream operations; the launch passes them to
    // hipExtLaunchKernelGGL (VR_LAUNCH_TIMED), which stamps them from the dispatch itself - no extra packets between
    // two dependent kernels on the stream
    VrKernelScope(vr_context* ctx, int id, hipStream_t stream, bool attach);
    // the pair enters the context's list only once something will stamp it (attach = true: after the launch was issued;
    // a scope that returns early without launching hands its events back instead of leaving a never-recorded pair)
    void commit();
    void launched();                // the dispatch that stamps the pair has been issued
    ~VrKernelScope();
};
// launch under a scope created with attach = true
#define VR_LAUNCH_TIMED(scope, kernel, grid, block, stream, ...) do { \
        if ((scope).e0 && (scope).e1) { hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, (scope).e0, (scope).e1, 0, __VA_ARGS__); (scope).launched(); } \
        else hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__); } while (0)

struct vr_gbuffer {
    vr_context* ctx;
    int w, h;
    float* depth; uint32_t* diffuse; uint32_t* specular; uint2* normals; uint2* emissive;
    // Depth range (bits of the smallest / largest depth below 1.0) of every 32x32 light tile, left behind by a tile pass that
    // was asked for it (vr_render_params::depth_ranges) and consumed - and reset to "none" = (0x7f800000, 0) - by the tiled
    // lighting pass's culling stage, which then need not read the depth plane a second time.
    uint2* d_ranges = nullptr;
    int ranges_tiles = 0;
    enum { RANGES_NONE = 0, RANGES_CLEAN, RANGES_VALID, RANGES_DIRTY };
    int ranges_state = RANGES_NONE;          // CLEAN: every entry "none"; VALID: the last writer of the G-buffer left them; DIRTY: stale
    int ranges_rank = 0, ranges_world = 1;   // the screen-tile split they were rendered for
    // Plane-state tracking (VR_OPT_PLANE_TRACKING): main_ps writes 0 to the emissive target for every pixel it shades
    // (terrain_ps.hlsl:80) and RenderTargets::Clear writes 0 everywhere, so on this path the plane only ever holds zeros -
    // 8 of the G-buffer's 28 bytes per pixel.  While the library KNOWS the plane is all zero (it cleared it, or a tile pass
    // that writes every pixel of the target has run since the last foreign write) the tile pass does not rewrite it.
    // Foreign writes: vr_gbuffer_upload of the plane -> not known zero; vr_gbuffer_describe -> the pointers have left the
    // library for good (`escaped`): nothing about the planes' contents is assumed ever again (the same goes for the light
    // tiles' depth ranges above).
    bool emissive_zero = false;
    bool escaped = false;
    // The same idea per REGION (the 8 rows x 32 pixels one wave of a 32-pixel raster tile resolves), one byte each, kept on the
    // device by the fast variant of the tile pass: kRegionClear = every pixel holds the clear values in all planes (RenderTargets::
    // Clear, or a pass over a cleared target that drew nothing there); kRegionSpec = every pixel holds the pass's one specular
    // constant (terrain_ps.hlsl:76).  A sky region that is known clear is not written again and a terrain region keeps its
    // specular plane - the planes' contents are what they would be anyway.  region_fill: -1 = the device array is current,
    // else the byte it has to be filled with before its next use (0 after anything foreign wrote a plane, kRegionClear after a clear).
    uint8_t* d_region = nullptr;
    int region_tiles = 0;
    int region_fill = 0;
    // RenderTargets::Clear (vr_gbuffer_clear) under the tracking is LAZY: the next tile pass that writes every pixel of every plane
    // anyway (whole frame, shaded) runs as "over a cleared target" and the 929 MB of clear values are never written twice;
    // anything else that looks at the planes first (a lighting pass, a partitioned / depth-only / fused pass, download, upload,
    // describe) materialises the clear (vr_gbuffer_materialise).  Clear + Render then costs what Render(assume_cleared) costs.
    bool clear_pending = false;
    bool cleared_once = false;       // (the clear at creation is a real one: the allocation holds anything)
};
int vr_gbuffer_materialise(vr_gbuffer* g, hipStream_t s);      // a pending clear is written now, on `s` (vr_host.hip)
constexpr uint32_t kRegionSpec = 1u, kRegionClear = 2u;
int vr_gbuffer_region_prepare(vr_gbuffer* g, hipStream_t s, uint8_t** out);     // allocated and current (vr_host.hip)
// What a pass that READS the G-buffer may take from the tracking instead of from memory (the lighting passes).
struct PlaneHints {
    const uint8_t* region;       // region states (NULL: nothing known per region)
    uint32_t spec_const;         // the value kRegionSpec stands for
    int emissive_zero;           // the emissive plane holds only zeros
    int tiles32_x;               // regions are indexed [32-pixel tile][wave of the tile = 8 rows]
};
int vr_gbuffer_plane_hints(vr_gbuffer* g, hipStream_t s, PlaneHints* out);
uint32_t vr_specular_constant(const vr_context* c);     // main_ps's specular output as the G-buffer holds it (terrain_ps.hlsl:76 -> SRGBA8)
// (anything else that writes the G-buffer: its depth ranges are stale)
inline void vr_gbuffer_touch(vr_gbuffer* g) { if (g->ranges_state == vr_gbuffer::RANGES_VALID) g->ranges_state = vr_gbuffer::RANGES_DIRTY; }
int vr_gbuffer_ranges_prepare(vr_gbuffer* g, hipStream_t s);      // allocated and every entry "none" (vr_host.hip)

struct vr_image {
    vr_context* ctx;
    int w, h;
    void* data;
    bool owned;
    size_t capacity_bytes;
    // vr_frame_submit's cross-stream bookkeeping: the lighting pass that wrote the image (when no dispatch-stamped event exists)
    // and the stage on another stream that still reads it
    hipEvent_t ev_written = nullptr, ev_read_done = nullptr;
    bool read_pending = false;
};

struct vr_ldr_image {
    vr_context* ctx;
    int w, h;
    void* data;
    bool owned;
    size_t capacity_bytes;
};

// Everything one frame's geometry stages produce and its tile pass consumes.  Three sets rotate: the tile pass of
// frame N reads one while the geometry of frames N+1 and N+2 is built in the other two (vr_terrain_prepare).
constexpr int kGeoSets = 3;
struct GeoSet {
    uint32_t* d_node_ids = nullptr;      // select outputs
    vr_instance* d_instances = nullptr;
    uint32_t* d_counters = nullptr;      // [0] selected count, [1] status flags, [2..7] frame work counters
    uint32_t* d_sel_scratch = nullptr;   // k_select: the frontiers' overflow beyond their LDS part and the selected keys (one workgroup's scratch)
    DevVert* d_verts = nullptr;          // max_instances*1089 regular + extra (clipper) region
    uint64_t* d_rect = nullptr;          // per triangle: tile rect or ~0 when culled
    uint4* d_recs = nullptr;             // per surviving triangle: its set-up record (kRecGroups x 16 B), then the clipper's (hard_cap * 4)
    uint32_t* d_hard_list = nullptr;     // triangle ids that need the clipper
    HardTriRec* d_hard_tris = nullptr;   // capacity hard_cap * 4
    uint32_t* d_hard_first = nullptr;    // per regular triangle id: first HardTriRec index
    uint32_t* d_tile_count = nullptr;    // per raster tile
    uint32_t* d_tile_offset = nullptr;
    uint32_t* d_tile_cursor = nullptr;
    int32_t* d_tile_order = nullptr;     // launch order of the tile pass: this frame's tiles by falling bin length
    TileEntry* d_bin_entries = nullptr;   // bin_capacity entries of 64 B (k_fill)
    int scratch_tiles = 0;
    int last_tiles = 0;                  // raster tiles of the target the last chain was built for (the stride of d_tile_order's class regions)
    hipEvent_t ev_geo_done = nullptr, ev_raster_done = nullptr;
    hipEvent_t raster_done = nullptr;    // what the geometry stream waits on before reusing this set: ev_raster_done, or the tile pass's own stop event
    uint64_t raster_done_epoch = 0;      // 0: raster_done is this set's own ev_raster_done (always valid); else the context's ev_epoch when the handle was taken
    bool raster_recorded = false, have_selection = false;
    // Successive chains on one set may run on different geometry streams (they take turns) and a chain is not always
    // consumed by a tile pass (an evicted prepared set, vr_terrain_select alone): every writer of the set first waits for
    // the set's previous chain (ev_geo_done) and for a lock_view copy that may still be reading its selection.
    bool geo_recorded = false;           // ev_geo_done has been recorded at least once
    hipEvent_t ev_sel_read = nullptr;    // recorded behind a lock_view copy OUT of this set (on the copying set's stream)
    bool sel_read_pending = false;
    bool main_waited = false;            // the context's stream already waits for this set's chain (queued by vr_terrain_prepare)
    hipStream_t main_wait_stream = nullptr;   // ... the stream that wait was queued on: a host may change the context's stream (vr_context_set_stream)
                                         // between vr_terrain_prepare and vr_terrain_render; the wait only counts for the stream that holds it
    // The geometry stream this set's last chain ran on (the terrain's two streams take turns, so that two prepared
    // frames have their latency-bound chains in flight at once).
    hipStream_t stream = nullptr;
    bool main_dep_pending = false;       // the context's stream changed the terrain (node heights): wait for ev_main_dep first
    bool status_pending = false;         // the last chain's counters have not been read from the host mirror yet
    // vr_terrain_prepare: geometry already built for exactly these inputs
    bool prepared = false;
    uint64_t prep_serial = 0;            // order of the vr_terrain_prepare calls (the oldest prepared set is evicted first)
    vr_view prep_view; vr_render_params prep_rp; int prep_w = 0, prep_h = 0, prep_rank = 0, prep_world = 0, prep_tile_shift = 0;
};

struct vr_terrain {
    vr_context* ctx;
    vr_terrain_params p;
    int num_lods;
    float lod_ranges[VR_MAX_LODS];
    DevTex height, albedo;
    uint8_t* d_height = nullptr; uint8_t* d_albedo = nullptr;
    uint64_t bytes_textures = 0, bytes_scratch = 0;     // vr_terrain_memory_bytes
    uint32_t extra_vert_cap = 0, hard_cap = 0;
    size_t bin_capacity = 0;
    // Per-frame scratch by HIGH-WATER MARK (round 4): vertices, triangle records and bins are sized for cap_instances nodes - not
    // for params.max_instances (4096: 1.7 GB per geometry set, of which an 8K frame's ~300 nodes use a few percent).  Every
    // chain leaves its counters in a pinned host mirror (k_fill's first act); the next API call reads the mirrors of the chains that
    // have completed (hipEventQuery, no wait), keeps the largest node count seen and doubles the scratch BEFORE a frame can
    // exceed it (count > half the capacity).  A frame that does exceed it - the count more than doubled within three frames -
    // is drawn without the excess nodes and reported like the other device-side conditions: sticky, by the next vr_terrain_render.
    int cap_instances = 0;
    uint32_t* h_status = nullptr;          // kGeoSets x 8 words, hipHostMalloc (mapped)
    uint32_t* d_status = nullptr;          // the device's view of it
    uint32_t high_water = 0;               // most nodes a completed frame selected
    size_t bin_high_water = 0, bin_want = 0;   // most bin entries a completed frame wanted; the capacity asked for because of it
    int sticky_error = 0;                  // VR_ERR_* of a completed frame, not yet reported
    uint32_t sticky_count = 0;
    GeoSet sets[kGeoSets];
    int cur = 0;                            // set of the most recent select / render
    uint64_t prep_counter = 0;
    // Two geometry streams, taken in turns by successive chains.  Not one per set: HIP multiplexes streams onto 4 hardware
    // queues by default (GPU_MAX_HW_QUEUES), and a frame loop with an exchange stage already uses 2-3 of its own; streams
    // that share a queue serialise (measured: a fifth stream cost the N = 8 rank emulation 0.23 -> 0.39 ms per frame).
    hipStream_t geo_streams[2] = { nullptr, nullptr };
    unsigned geo_turn = 0;
    // QuadTree::SetHeight results: (position.y, extents.y) per node id; m_HeightLoaded
    float2* d_node_heights = nullptr;
    uchar2* d_minmax = nullptr;          // raw (min, max) bytes per node of one surface: scratch of the mip-style SetHeight
    bool height_loaded = false;
    float texel_size[2] = { 0.0f, 0.0f };   // m_TexelSize (QuadTree.cpp:29)
    int surfaces_per_side = 1;              // WORLD_SIZE / SURFACE_SIZE (TerrainPass.cpp:97)
    // Geometry stream: select / vertex / setup / bins depend only on the view, so they run on their own
    // stream and overlap whatever the context's stream is doing (the lighting pass of the previous frame,
    // or - after vr_terrain_prepare - its tile pass); the tile pass on the context's stream waits for them.
    hipEvent_t ev_sel_copy = nullptr;      // lock_view: the source set's selection is complete
    hipEvent_t ev_main_dep = nullptr, ev_raster_begin = nullptr;   // ev_raster_begin: the context's stream reached the last tile pass
    bool raster_begin_recorded = false;
    hipEvent_t start_hint = nullptr;        // vr_terrain_prepare starts its geometry behind this: ev_raster_begin, or the previous lighting pass's stop event
    uint64_t start_hint_epoch = 0;          // as GeoSet::raster_done_epoch
};

struct vr_tonemap;
vr_context* vr_tonemap_context(vr_tonemap* tm);
// ---- cross-TU entry points ----------------------------------------------------------
int vr_tex_upload_and_mip(vr_context* ctx, const uint8_t* host, int w, int h, int texel_bytes,
                          DevTex* out, uint8_t** out_mem, uint64_t* out_bytes = nullptr);
int vr_select_launch(vr_terrain* t, GeoSet& g, const vr_view* view, float max_height, hipStream_t stream);
int vr_terrain_pick_set(vr_terrain* t);
// reads the counters of completed chains (no wait), grows the scratch by the high-water mark; returns a completed frame's sticky
// device-side error once (VR_OK otherwise).  Called at the head of vr_terrain_render / vr_terrain_prepare / vr_terrain_select.
int vr_terrain_poll(vr_terrain* t, bool report);
int vr_terrain_reserve_bins(vr_terrain* t, size_t tiles);     // room for a target of that many raster tiles (vr_select.hip)
// tables of (w, h, part); part == NULL is the whole frame as rank 0 of 1
int vr_partition_tables(vr_context* ctx, int w, int h, const vr_partition* part, const PartTables** out);
// any cached table set of (w, h, world): the slot table does not depend on the rank
int vr_partition_slot_tables(vr_context* ctx, int w, int h, int world, const PartTables** out);

// The geometry kernels (select, vertex, setup, clip, scan, fill) are short and latency-bound and run on CUs full of
// tile-pass waves: they raise their waves' issue priority so that their few instructions are not queued behind the tile
// pass's many (the sequencer arbitrates by priority, then age).
#ifndef VR_GEOMETRY_PRIO
#define VR_GEOMETRY_PRIO 3
#endif
#define VR_GEOMETRY_PRIORITY() __builtin_amdgcn_s_setprio(VR_GEOMETRY_PRIO)

// ---- device helpers shared by kernels ---------------------------------------------------
__device__ __forceinline__ float vr_max(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float vr_min(float a, float b) { return a < b ? a : b; }
// x clamped to [lo, hi] in one instruction (v_med3_f32); x must not be NaN
__device__ __forceinline__ float vr_clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// 1 / d and sqrt(x), correctly rounded, for operands well inside the normal range (2^-60 < |d|, x < 2^60): short sequences
// found by exhaustive search (tools/micro/exact_math.hip, profiles/r03_exact_math_search.txt: every float of the range
// compared with 1.0f / d and sqrtf(x) on the device; the product's copies are swept again by vr_debug_fastmath_check).
//   reciprocal : v_rcp_f32 (1 ulp) + ONE Newton step with an exact fma residual      (the compiler's IEEE division: 10 instructions)
//   square root: v_rsq_f32 + the compiler's own refinement without the range scaling (no compare / select)
__device__ __forceinline__ float vr_rcp_exact(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float vr_sqrt_exact(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    float s = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    h = __builtin_fmaf(h, e, h);
    s = __builtin_fmaf(s, e, s);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ float vr_saturate(float x) { return vr_min(vr_max(x, 0.0f), 1.0f); }
__device__ __forceinline__ float vr_dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;
}
