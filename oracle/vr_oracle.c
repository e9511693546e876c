/*
 * vr_oracle.c — CPU restatement of the vrenderer terrain + deferred-shading hot path.
 *
 * TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (see vr_oracle.h for both statements).
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference tree).  Pieces the reference takes from the absent Donut submodule are
 * marked [DONUT-RECOLLECTION]: they restate Donut's published behaviour from memory
 * and *define* the model the HIP path is checked against.
 *
 * Floating-point discipline (so a GPU can reproduce results bit for bit): every
 * expression is evaluated in the written order with no implicit contraction (build
 * with -ffp-contract=off); the only operations are + - * / sqrt and the fused
 * multiply-add WHERE IT IS WRITTEN as fmaf()/fma() (all IEEE correctly rounded), and
 * no libm transcendental runs on any per-vertex / per-pixel path.  dot3(a,b) is
 * (a.x*b.x + a.y*b.y) + a.z*b.z everywhere.
 *
 * Raster model, revision 3 (round 3).  The parts of the pipeline that D3D leaves to
 * the implementation - attribute interpolation, the sampler's address / filter
 * arithmetic and the implicit level of detail - are stated the way GPUs execute them:
 *   - interpolation by per-triangle PLANE EQUATIONS set up in double precision and
 *     evaluated per pixel with two fused multiply-adds (depth, 1/w and attribute/w
 *     are affine in screen space; fixed-function interpolators work this way);
 *   - the sampler's `u * size - 0.5`, its lerps and the LOD's sums of squares as
 *     fused multiply-adds (what `mad` is on every current GPU).
 * Everything the HLSL source itself pins (uv = (pos + half) / size, uv + offset,
 * hDx = a - b, normalize) keeps its written order.  Revisions 1-2 evaluated the same
 * quantities from barycentrics with unfused arithmetic; the results differ in the last
 * bits only, and the HIP path is checked against THIS file bit for bit either way.
 */
#define _POSIX_C_SOURCE 200809L
#include "vr_oracle.h"

/* The raster / sampler model's revision (header comment above).  FROZEN from round 4 on: tests/test_oracle_cpu.py bounds this
 * revision against an exact (float64 / rational) evaluation of the HLSL semantics and pins its output by digest
 * (tests/golden/MODEL_REVISIONS.txt); a change of any per-pixel result needs a new revision number, a new digest line and
 * the diff statistics against that float64 model in the commit.  A kernel optimisation is no reason for one. */
#define ORC_MODEL_REVISION 3

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* small helpers                                                               */
/* ------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float fmaxx(float a, float b) { return a > b ? a : b; }
static inline float fminx(float a, float b) { return a < b ? a : b; }
static inline float saturatef(float x) { return fminx(fmaxx(x, 0.0f), 1.0f); }
static inline float dot3(const float a[3], const float b[3]) { return (a[0]*b[0] + a[1]*b[1]) + a[2]*b[2]; }

/* IEEE binary16 <-> binary32, round to nearest even (RGBA16_FLOAT targets). */
uint16_t orc_float_to_half(float f)
{
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);           /* rounds to inf */
    if (ax < 0x33000001u) return (uint16_t)sign;                         /* rounds to 0   */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t hexp;
    if (e < -14) { shift = 13 + (-14 - e); hexp = 0; }                    /* subnormal     */
    else         { shift = 13; hexp = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    uint32_t r = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (r & 1u))) r++;
    return (uint16_t)(sign | (hexp + r));
}

float orc_half_to_float(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 31u, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float)m * (1.0f / 16777216.0f);                        /* m * 2^-24     */
        return (sign ? -v : v);
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

/* sRGB tables.  Decode: exact sRGB EOTF evaluated in double, rounded to float
 * (what a SRGBA8 texture fetch returns before filtering).  Encode: code(x) =
 * #{k in 1..255 : x >= thr[k]}, thr[k] = EOTF((k-0.5)/255) — i.e. round-to-nearest
 * of the OETF without evaluating a pow per pixel. */
static float g_srgb_lut[256];
static float g_srgb_thr[256];
static int   g_tables_ready = 0;
static double srgb_eotf(double c) { return c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4); }
static void init_tables(void)
{
    if (g_tables_ready) return;
    for (int i = 0; i < 256; i++) g_srgb_lut[i] = (float)srgb_eotf((double)i / 255.0);
    g_srgb_thr[0] = 0.0f;
    for (int k = 1; k < 256; k++) g_srgb_thr[k] = (float)srgb_eotf(((double)k - 0.5) / 255.0);
    g_tables_ready = 1;
}
float orc_srgb8_to_linear(uint8_t c) { init_tables(); return g_srgb_lut[c]; }
uint8_t orc_linear_to_srgb8(float x)
{
    init_tables();
    int lo = 0, hi = 255;                 /* largest k with x >= thr[k] (thr[0] = 0) */
    if (!(x >= 0.0f)) return 0;           /* negatives and NaN -> 0                   */
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (x >= g_srgb_thr[mid]) lo = mid; else hi = mid - 1; }
    return (uint8_t)lo;
}
void orc_linear_to_srgb8_batch(const float* in, size_t n, uint8_t* out)
{
    for (size_t i = 0; i < n; i++) out[i] = orc_linear_to_srgb8(in[i]);
}
static inline uint8_t unorm8(float a)
{
    if (!(a > 0.0f)) return 0;
    if (a >= 1.0f) return 255;
    return (uint8_t)(int)floorf(a * 255.0f + 0.5f);
}
/* D3D float -> SNORM16: clamp, scale by 32767, round half away from zero. */
static inline uint16_t snorm16(float v)
{
    if (!(v == v)) return 0;
    v = fminx(fmaxx(v, -1.0f), 1.0f);
    float s = v * 32767.0f;
    int i = (int)(s >= 0.0f ? s + 0.5f : s - 0.5f);
    return (uint16_t)(int16_t)i;
}
static inline float snorm16_decode(uint16_t u) { return fmaxx((float)(int16_t)u / 32767.0f, -1.0f); }

/* ------------------------------------------------------------------------- */
/* textures: mip chains + sampling [DONUT-RECOLLECTION: TextureCache uploads the  */
/* PNG with a full mip chain generated by 2x2 box blits; heightmap R8_UNORM,    */
/* albedo SRGBA8_UNORM (Renderer.cpp:51-55)]                                    */
/* ------------------------------------------------------------------------- */
#define ORC_MAX_LEVELS 16
typedef struct {
    int levels;
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    uint8_t* data[ORC_MAX_LEVELS];
    int texel_bytes; /* 1 (R8) or 4 (sRGBA8) */
} orc_tex;

static int mip_levels(int w, int h) { int m = w > h ? w : h, l = 1; while (m > 1) { m >>= 1; l++; } return l; }

static void tex_build(orc_tex* t, const uint8_t* src, int w, int h, int texel_bytes)
{
    init_tables();
    t->texel_bytes = texel_bytes;
    t->levels = mip_levels(w, h);
    t->w[0] = w; t->h[0] = h;
    t->data[0] = (uint8_t*)malloc((size_t)w * h * texel_bytes);
    memcpy(t->data[0], src, (size_t)w * h * texel_bytes);
    for (int l = 1; l < t->levels; l++) {
        int sw = t->w[l-1], sh = t->h[l-1];
        int dw = sw > 1 ? sw >> 1 : 1, dh = sh > 1 ? sh >> 1 : 1;
        t->w[l] = dw; t->h[l] = dh;
        t->data[l] = (uint8_t*)malloc((size_t)dw * dh * texel_bytes);
        const uint8_t* s = t->data[l-1];
        uint8_t* d = t->data[l];
        for (int y = 0; y < dh; y++) for (int x = 0; x < dw; x++) {
            int x0 = 2*x, x1 = 2*x+1 < sw ? 2*x+1 : sw-1;
            int y0 = 2*y, y1 = 2*y+1 < sh ? 2*y+1 : sh-1;
            if (texel_bytes == 1) {
                int sum = s[y0*sw+x0] + s[y0*sw+x1] + s[y1*sw+x0] + s[y1*sw+x1];
                d[y*dw+x] = (uint8_t)((sum + 2) >> 2);
            } else {
                const uint8_t* p00 = s + 4*((size_t)y0*sw+x0); const uint8_t* p10 = s + 4*((size_t)y0*sw+x1);
                const uint8_t* p01 = s + 4*((size_t)y1*sw+x0); const uint8_t* p11 = s + 4*((size_t)y1*sw+x1);
                uint8_t* o = d + 4*((size_t)y*dw+x);
                for (int c = 0; c < 3; c++) {
                    float a = (g_srgb_lut[p00[c]] + g_srgb_lut[p10[c]]) + (g_srgb_lut[p01[c]] + g_srgb_lut[p11[c]]);
                    o[c] = orc_linear_to_srgb8(a * 0.25f);
                }
                o[3] = (uint8_t)((p00[3] + p10[3] + p01[3] + p11[3] + 2) >> 2);
            }
        }
    }
}
static void tex_free(orc_tex* t) { for (int l = 0; l < t->levels; l++) free(t->data[l]); t->levels = 0; }

/* one bilinear tap, clamp addressing, fp32 weights, lerps as fused multiply-adds; out = 1 (R8) or 3 (rgb) floats */
static void tex_bilinear(const orc_tex* t, int level, float u, float v, float out[3])
{
    int w = t->w[level], h = t->h[level];
    const uint8_t* d = t->data[level];
    float x = fmaf(u, (float)w, -0.5f), y = fmaf(v, (float)h, -0.5f);
    float xf = floorf(x), yf = floorf(y);
    float fx = x - xf, fy = y - yf;
    /* clamp in float first so huge / non-finite coordinates cannot overflow the int */
    xf = fminx(fmaxx(xf, -1.0f), (float)w); yf = fminx(fmaxx(yf, -1.0f), (float)h);
    int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > w-1 ? w-1 : x0); x1 = x1 < 0 ? 0 : (x1 > w-1 ? w-1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > h-1 ? h-1 : y0); y1 = y1 < 0 ? 0 : (y1 > h-1 ? h-1 : y1);
    if (t->texel_bytes == 1) {
        float t00 = (float)d[y0*w+x0] / 255.0f, t10 = (float)d[y0*w+x1] / 255.0f;
        float t01 = (float)d[y1*w+x0] / 255.0f, t11 = (float)d[y1*w+x1] / 255.0f;
        float top = fmaf(t10 - t00, fx, t00), bot = fmaf(t11 - t01, fx, t01);
        out[0] = fmaf(bot - top, fy, top);
    } else {
        const uint8_t* p00 = d + 4*((size_t)y0*w+x0); const uint8_t* p10 = d + 4*((size_t)y0*w+x1);
        const uint8_t* p01 = d + 4*((size_t)y1*w+x0); const uint8_t* p11 = d + 4*((size_t)y1*w+x1);
        for (int c = 0; c < 3; c++) {
            float t00 = g_srgb_lut[p00[c]], t10 = g_srgb_lut[p10[c]], t01 = g_srgb_lut[p01[c]], t11 = g_srgb_lut[p11[c]];
            float top = fmaf(t10 - t00, fx, t00), bot = fmaf(t11 - t01, fx, t01);
            out[c] = fmaf(bot - top, fy, top);
        }
    }
}
/* trilinear at an explicit level of detail */
static void tex_trilinear(const orc_tex* t, float lod, float u, float v, float out[3])
{
    int n = t->texel_bytes == 1 ? 1 : 3;
    float maxl = (float)(t->levels - 1);
    if (!(lod > 0.0f)) lod = 0.0f;
    if (lod > maxl) lod = maxl;
    float lf = floorf(lod);
    int l0 = (int)lf;
    float f = lod - lf;
    float a[3], b[3];
    tex_bilinear(t, l0, u, v, a);
    if (f > 0.0f) {
        tex_bilinear(t, l0 + 1, u, v, b);
        for (int c = 0; c < n; c++) out[c] = fmaf(b[c] - a[c], f, a[c]);
    } else {
        for (int c = 0; c < n; c++) out[c] = a[c];
    }
}
/* implicit LOD of Texture2D::Sample from screen-space uv differences
 * (D3D11 functional spec 7.18.11, isotropic): rho = max(|d/dx|, |d/dy|) in texels,
 * lod = log2(rho).  log2 is the pinned cubic below (max error 1.1e-3 LOD; the spec
 * permits an approximate LOD) so that CPU and GPU agree bit for bit. */
static float lod_from_derivs(float dudx, float dvdx, float dudy, float dvdy, int w, int h)
{
    float ax = dudx * (float)w, ay = dvdx * (float)h, bx = dudy * (float)w, by = dvdy * (float)h;
    float r2x = fmaf(ax, ax, ay*ay), r2y = fmaf(bx, bx, by*by);
    float r2 = fmaxf(r2x, r2y);                                   /* rho^2; IEEE maxNum: a NaN operand loses */
    /* 0.5 * log2(r2) from the bits, for every input: magnified footprints (r2 <= 1) give a value <= 0 and an
     * infinite one gives 64; the sampler clamps the result to [0, last level] (tex_trilinear) either way. */
    uint32_t bits = f2u(r2);
    int e = (int)((bits >> 23) & 255u) - 127;
    float t = u2f((bits & 0x7fffffu) | 0x3f800000u) - 1.0f;
    float p = t * fmaf(t, fmaf(t, 0.1563861f, -0.57725066f), 1.4208646f);
    return 0.5f * ((float)e + p);
}

/* ------------------------------------------------------------------------- */
/* quadtree (QuadTree.h:22-55, QuadTree.cpp)                                    */
/* ------------------------------------------------------------------------- */
typedef struct orc_node {
    float pos[3];                 /* m_Position */
    float ext[3];                 /* m_Extents  */
    struct orc_node* child[4];    /* TL, TR, BL, BR (QuadTree.h:51-54) */
    uint32_t id;                  /* not in the reference: (4^d-1)/3 + iz*2^d + ix */
} orc_node;

struct orc_terrain {
    vr_terrain_params p;
    orc_tex height, albedo;
    orc_node* root;               /* first quadtree (== roots[0]) */
    orc_node* roots[64];          /* one QuadTree per surface (TerrainPass.cpp:97-110) */
    int num_surfaces;
    long nodes_per_tree;
    uint32_t id_offset;           /* while building: surface index * nodes_per_tree */
    long num_nodes;
    int num_lods;
    float lod_ranges[VR_MAX_LODS];
    float texel_size[2];
    int height_loaded;            /* m_HeightLoaded (QuadTree.h:70) */
    /* select scratch */
    const orc_node** selected; int num_selected, cap_selected;
    int stub_frustum;
    int have_selection;
};

/* QuadTree::InitLodRanges (QuadTree.cpp:234-241) */
static void init_lod_ranges(float* r, float min_lod_distance)
{
    for (int i = 0; i < VR_MAX_LODS; i++) r[i] = min_lod_distance * powf(2.0f, (float)i);
}

static orc_node* node_new(const float pos[3], const float ext[3], uint32_t id)
{
    orc_node* n = (orc_node*)malloc(sizeof(orc_node));
    memcpy(n->pos, pos, 12); memcpy(n->ext, ext, 12);
    n->child[0] = n->child[1] = n->child[2] = n->child[3] = NULL;
    n->id = id;
    return n;
}
static uint32_t level_base(int d) { return (uint32_t)((((uint64_t)1 << (2*d)) - 1) / 3); }

/* QuadTree::Split (QuadTree.cpp:210-232) */
static void split(orc_terrain* t, orc_node* node, int num_splits, int d, uint32_t ix, uint32_t iz)
{
    float e[3] = { node->ext[0] / 2.0f, node->ext[1] / 2.0f, node->ext[2] / 2.0f };
    float p0[3] = { node->pos[0] + (-e[0]), node->pos[1] + 0.0f, node->pos[2] + e[2] };        /* TL */
    float p1[3] = { node->pos[0] + e[0],    node->pos[1] + e[1], node->pos[2] + e[2] };        /* TR */
    float p2[3] = { node->pos[0] - e[0],    node->pos[1] - e[1], node->pos[2] - e[2] };        /* BL */
    float p3[3] = { node->pos[0] - (-e[0]), node->pos[1] - 0.0f, node->pos[2] - e[2] };        /* BR */
    int cd = d + 1;
    uint32_t base = level_base(cd), n = 1u << cd;
    uint32_t cix[4] = { 2*ix, 2*ix+1, 2*ix, 2*ix+1 }, ciz[4] = { 2*iz+1, 2*iz+1, 2*iz, 2*iz };
    const float* ps[4] = { p0, p1, p2, p3 };
    for (int i = 0; i < 4; i++) node->child[i] = node_new(ps[i], e, t->id_offset + base + ciz[i] * n + cix[i]);
    t->num_nodes += 4;
    num_splits++;
    if (num_splits <= t->num_lods)
        for (int i = 0; i < 4; i++) split(t, node->child[i], num_splits, cd, cix[i], ciz[i]);
}
static void free_tree(orc_node* n) { if (!n) return; for (int i = 0; i < 4; i++) free_tree(n->child[i]); free(n); }

/* Node::Intersects (QuadTree.h:31-45): xz-only distance to the box, compared with
 * the SQUARED range the caller passes. */
static int node_intersects(const orc_node* n, const float position[3], float radius)
{
    float mn[3] = { n->pos[0] - n->ext[0], n->pos[1] - n->ext[1], n->pos[2] - n->ext[2] };
    float mx[3] = { n->pos[0] + n->ext[0], n->pos[1] + n->ext[1], n->pos[2] + n->ext[2] };
    float d[3] = { 0.0f, 0.0f, 0.0f };
    if (position[0] < mn[0]) d[0] = position[0] - mn[0];
    else if (position[0] > mx[0]) d[0] = position[0] - mx[0];
    if (position[2] < mn[2]) d[2] = position[2] - mn[2];
    else if (position[2] > mx[2]) d[2] = position[2] - mx[2];
    return dot3(d, d) <= radius;
}

/* dm::frustum::intersectsWith(box3) [DONUT-RECOLLECTION]: per plane, take the box
 * corner nearest to the inside (min where the outward normal is positive, else
 * max) and reject when it lies outside. */
static int frustum_intersects_box(const vr_view* v, const float mn[3], const float mx[3])
{
    for (int i = 0; i < 6; i++) {
        const float* pl = v->planes[i];
        float pt[3] = { pl[0] > 0.0f ? mn[0] : mx[0], pl[1] > 0.0f ? mn[1] : mx[1], pl[2] > 0.0f ? mn[2] : mx[2] };
        float dist = dot3(pl, pt) - pl[3];
        if (dist > 0.0f) return 0;
    }
    return 1;
}

static void push_selected(orc_terrain* t, const orc_node* n)
{
    if (t->num_selected == t->cap_selected) {
        t->cap_selected = t->cap_selected ? t->cap_selected * 2 : 4096;
        t->selected = (const orc_node**)realloc((void*)t->selected, sizeof(void*) * t->cap_selected);
    }
    t->selected[t->num_selected++] = n;
}

/* QuadTree::NodeSelect (QuadTree.cpp:80-131) */
static int node_select(orc_terrain* t, const float position[3], const orc_node* node, int lod,
                       const vr_view* v, float max_height)
{
    if (!node_intersects(node, position, t->lod_ranges[lod] * t->lod_ranges[lod])) return 0;
    float mn[3] = { node->pos[0] - node->ext[0], node->pos[1] - node->ext[1], node->pos[2] - node->ext[2] };
    float mx[3] = { node->pos[0] + node->ext[0], node->pos[1] + node->ext[1], node->pos[2] + node->ext[2] };
    if (t->height_loaded) { mn[1] *= max_height; mx[1] *= max_height; }
    else { mn[1] = 0.0f; mx[1] = position[1]; }
    if (!t->stub_frustum && !frustum_intersects_box(v, mn, mx)) return 1;   /* culled: parent must not select it */
    if (lod == 0) { push_selected(t, node); return 1; }
    if (!node_intersects(node, position, t->lod_ranges[lod-1] * t->lod_ranges[lod-1])) {
        push_selected(t, node);
    } else {
        for (int i = 0; i < 4; i++)
            if (!node_select(t, position, node->child[i], lod - 1, v, max_height))
                push_selected(t, node->child[i]);
    }
    return 1;
}

/* QuadTree::GetHeightValue / GetMinMaxHeightValue / SetHeight (QuadTree.cpp:153-208) */
static float get_height_value(const orc_terrain* t, float px, float py)
{
    int index = (int)(px + py * (float)t->height.w[0]);
    long n = (long)t->height.w[0] * t->height.h[0];
    if (index < 0) index = 0;                 /* the reference reads out of bounds at  */
    if (index >= n) index = (int)(n - 1);     /* the max edge (ceil(maxV)); clamp here */
    return (float)t->height.data[0][index] / 255.0f;
}
static void get_min_max_height(const orc_terrain* t, float px, float pz, float width, float height, float out[2])
{
    float minx = px - width / 2, miny = pz - height / 2;
    minx += t->p.world_size / 2; miny += t->p.world_size / 2;
    minx *= t->texel_size[0]; miny *= t->texel_size[1];
    float maxx = minx + width * t->texel_size[0], maxy = miny + height * t->texel_size[1];
    int lx0 = (int)floorf(minx), lx1 = (int)ceilf(maxx), ly0 = (int)floorf(miny), ly1 = (int)ceilf(maxy);
    float mn = INFINITY, mx = -INFINITY;
    for (int i = lx0; i < lx1; i++) for (int j = ly0; j < ly1; j++) {
        float s = get_height_value(t, (float)i, (float)j);
        mn = fminx(mn, s); mx = fmaxx(mx, s);
    }
    mn = (mx - mn) == 0.0f ? 0.0f : mn;
    out[0] = mn; out[1] = mx;
}
static void set_height(orc_terrain* t, orc_node* node, int num_splits)
{
    float mm[2];
    get_min_max_height(t, node->pos[0], node->pos[2], node->ext[0] * 2.0f, node->ext[2] * 2.0f, mm);
    float extent = (mm[1] - mm[0]) / 2.0f;
    node->pos[1] = mm[0] + extent;
    node->ext[1] = extent;
    num_splits++;
    if (num_splits <= t->num_lods)
        for (int i = 0; i < 4; i++) set_height(t, node->child[i], num_splits);
}

static int ilog2_floor_f(float x)   /* static_cast<int>(log2(width)) for width >= 1 */
{
    uint32_t b = f2u(x);
    return (int)((b >> 23) & 255u) - 127;
}

orc_terrain* orc_terrain_create(const vr_terrain_params* p, const uint8_t* height_r8, int hm_w, int hm_h,
                                const uint8_t* albedo, int al_w, int al_h)
{
    init_tables();
    orc_terrain* t = (orc_terrain*)calloc(1, sizeof(*t));
    t->p = *p;
    tex_build(&t->height, height_r8, hm_w, hm_h, 1);
    tex_build(&t->albedo, albedo, al_w, al_h, 4);
    init_lod_ranges(t->lod_ranges, p->min_lod_distance);
    /* QuadTree::Init (QuadTree.cpp:19-52) */
    int l2 = p->surface_size >= 1.0f ? ilog2_floor_f(p->surface_size) : 0;
    t->num_lods = (VR_MAX_LODS - 1) < l2 ? (VR_MAX_LODS - 1) : l2;
    t->texel_size[0] = (float)hm_w / p->world_size;
    t->texel_size[1] = (float)hm_h / p->world_size;
    float ext[3] = { p->surface_size / 2.0f, 0.0f, p->surface_size / 2.0f };
    /* TerrainPass::Init (TerrainPass.cpp:97-110): WORLD_SIZE / SURFACE_SIZE surfaces per side, one QuadTree each */
    const int per_side = (int)p->world_size / (int)p->surface_size;
    t->num_surfaces = per_side * per_side;
    t->nodes_per_tree = (long)level_base(t->num_lods + 1);
    t->num_nodes = 0;
    for (int i = 0; i < t->num_surfaces && i < 64; i++) {
        int column = i % per_side, row = i / per_side;
        float x = -0.5f * (float)(per_side - 1) + (float)column, y = -0.5f * (float)(per_side - 1) + (float)row;
        float loc[3] = { p->location[0] + x * p->surface_size, p->location[1] + 0.0f, p->location[2] + y * p->surface_size };
        t->id_offset = (uint32_t)((long)i * t->nodes_per_tree);
        t->roots[i] = node_new(loc, ext, t->id_offset);
        t->num_nodes += 1;
        split(t, t->roots[i], 1, 0, 0, 0);
    }
    t->root = t->roots[0];
    return t;
}
void orc_terrain_destroy(orc_terrain* t)
{
    if (!t) return;
    for (int i = 0; i < t->num_surfaces; i++) free_tree(t->roots[i]);
    tex_free(&t->height); tex_free(&t->albedo);
    free((void*)t->selected); free(t);
}
int  orc_terrain_num_lods(const orc_terrain* t) { return t->num_lods; }
void orc_terrain_lod_ranges(const orc_terrain* t, float out[VR_MAX_LODS]) { memcpy(out, t->lod_ranges, sizeof(float) * VR_MAX_LODS); }
long orc_terrain_num_nodes(const orc_terrain* t) { return t->num_nodes; }
int  orc_terrain_height_levels(const orc_terrain* t) { return t->height.levels; }
int  orc_terrain_albedo_levels(const orc_terrain* t) { return t->albedo.levels; }
const uint8_t* orc_terrain_height_mip(const orc_terrain* t, int l, int* w, int* h) { *w = t->height.w[l]; *h = t->height.h[l]; return t->height.data[l]; }
const uint8_t* orc_terrain_albedo_mip(const orc_terrain* t, int l, int* w, int* h) { *w = t->albedo.w[l]; *h = t->albedo.h[l]; return t->albedo.data[l]; }

/* TerrainPass::UpdateTransforms (TerrainPass.cpp:234-256): scaling(extents) *
 * translation(position) in Donut's row-vector affine3, written by
 * affineToColumnMajor as float3x4 rows (ex,0,0,px) (0,ey,0,py) (0,0,ez,pz). */
static void update_transform(const orc_node* n, vr_instance* o)
{
    memset(o, 0, sizeof(*o));
    o->transform[0] = n->ext[0]; o->transform[3]  = n->pos[0];
    o->transform[5] = n->ext[1]; o->transform[7]  = n->pos[1];
    o->transform[10] = n->ext[2]; o->transform[11] = n->pos[2];
    o->first_geometry_instance_index = 0; o->first_geometry_index = 0; o->num_geometries = 1; o->padding = 0;
}

int orc_select(orc_terrain* t, const vr_view* v, float max_height, int stub_frustum,
               uint32_t* node_ids, vr_instance* inst, int capacity)
{
    t->num_selected = 0;                       /* ClearSelectedNodes (TerrainPass.cpp:178) */
    t->stub_frustum = stub_frustum;
    float pos[3] = { v->camera_pos[0], v->camera_pos[1], v->camera_pos[2] };
    for (int s = 0; s < t->num_surfaces; s++)                    /* TerrainPass.cpp:176-186: trees in order, shared instance buffer */
        node_select(t, pos, t->roots[s], t->num_lods, v, max_height);
    t->have_selection = 1;
    for (int i = 0; i < t->num_selected && i < capacity; i++) {
        if (node_ids) node_ids[i] = t->selected[i]->id;
        if (inst) update_transform(t->selected[i], &inst[i]);
    }
    return t->num_selected;
}

void orc_set_height(orc_terrain* t) { for (int s = 0; s < t->num_surfaces; s++) set_height(t, t->roots[s], 0); }
/* m_HeightLoaded = true is what the reference's (commented-out) async task sets after SetHeight
 * (QuadTree.cpp:46-51); NodeSelect then culls with the node's real y-bounds (:87-91). */
void orc_set_height_loaded(orc_terrain* t, int loaded) { t->height_loaded = loaded; }
/* Back to the tree as Split built it (y = location.y, extents.y = 0) with m_HeightLoaded = false: the state
 * vr_terrain_update_heights(t, 0) stands for.  (The reference never leaves the loaded state once entered.) */
static void reset_height(orc_node* n, float y)
{
    if (!n) return;
    n->pos[1] = y; n->ext[1] = 0.0f;
    for (int i = 0; i < 4; i++) reset_height(n->child[i], y);
}
void orc_reset_height(orc_terrain* t)
{
    for (int s = 0; s < t->num_surfaces; s++) reset_height(t->roots[s], t->p.location[1]);
    t->height_loaded = 0;
}
static void dump_heights(const orc_node* n, float* out, long max_ids)
{
    if (!n) return;
    if ((long)n->id < max_ids) { out[2 * n->id] = n->pos[1]; out[2 * n->id + 1] = n->ext[1]; }
    for (int i = 0; i < 4; i++) dump_heights(n->child[i], out, max_ids);
}
void orc_node_heights(const orc_terrain* t, float* out, long max_ids) { for (int s = 0; s < t->num_surfaces; s++) dump_heights(t->roots[s], out, max_ids); }
int orc_node_height(const orc_terrain* t, uint32_t id, float* py, float* ey)
{
    /* walk down from the root following the id's (depth, ix, iz) */
    int d = 0; while (level_base(d + 1) <= id) d++;
    uint32_t rel = id - level_base(d), n = 1u << d, ix = rel % n, iz = rel / n;
    const orc_node* node = t->root;
    for (int l = d - 1; l >= 0 && node; l--) {
        uint32_t bx = (ix >> l) & 1u, bz = (iz >> l) & 1u;
        int c = bz ? (bx ? 1 : 0) : (bx ? 3 : 2);
        node = node->child[c];
    }
    if (!node || node->id != id) return -1;
    *py = node->pos[1]; *ey = node->ext[1];
    return 0;
}

/* ------------------------------------------------------------------------- */
/* view  [DONUT-RECOLLECTION]: FirstPersonCamera::LookAt, perspProjD3DStyle,   */
/* PlanarView::UpdateCache, dm::frustum(viewProj)  (Renderer.cpp:97,312-319)   */
/* ------------------------------------------------------------------------- */
static void mat4_mul(const float a[16], const float b[16], float o[16])
{
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        float s = ((a[i*4+0]*b[0*4+j] + a[i*4+1]*b[1*4+j]) + a[i*4+2]*b[2*4+j]) + a[i*4+3]*b[3*4+j];
        o[i*4+j] = s;
    }
}
static int mat4_inverse_d(const float m[16], float out[16])
{
    double a[4][8];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = m[i*4+j]; a[i][j+4] = (i == j); }
    for (int c = 0; c < 4; c++) {
        int piv = c; double best = fabs(a[c][c]);
        for (int r = c + 1; r < 4; r++) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
        if (best == 0.0) return -1;
        if (piv != c) for (int j = 0; j < 8; j++) { double tmp = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = tmp; }
        double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; j++) a[c][j] *= inv;
        for (int r = 0; r < 4; r++) if (r != c) { double f = a[r][c]; if (f != 0.0) for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[i*4+j] = (float)a[i][j+4];
    return 0;
}
static void normalize3(float v[3])
{
    float l = sqrtf(dot3(v, v));
    if (l > 0.0f) { v[0] /= l; v[1] /= l; v[2] /= l; }
}
static void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1]*b[2] - a[2]*b[1]; o[1] = a[2]*b[0] - a[0]*b[2]; o[2] = a[0]*b[1] - a[1]*b[0];
}
static void set_plane(float pl[4], float x, float y, float z, float d)
{
    float n[3] = { x, y, z };
    float l2 = dot3(n, n);
    float s = l2 > 0.0f ? 1.0f / sqrtf(l2) : 0.0f;
    pl[0] = x * s; pl[1] = y * s; pl[2] = z * s; pl[3] = d * s;
}
void orc_view_from_camera(const float eye[3], const float target[3], const float up_in[3],
                          float vfov, float z_near, float z_far, int w, int h, vr_view* o)
{
    memset(o, 0, sizeof(*o));
    float dir[3] = { target[0]-eye[0], target[1]-eye[1], target[2]-eye[2] };
    normalize3(dir);
    float up[3] = { up_in[0], up_in[1], up_in[2] };
    normalize3(up);
    float right[3]; cross3(dir, up, right); normalize3(right);
    cross3(right, dir, up); normalize3(up);
    /* worldToView = translation(-eye) * from_cols(right, up, dir) (row-vector convention) */
    float* m = o->world_to_view;
    for (int i = 0; i < 3; i++) { m[i*4+0] = right[i]; m[i*4+1] = up[i]; m[i*4+2] = dir[i]; m[i*4+3] = 0.0f; }
    float ne[3] = { -eye[0], -eye[1], -eye[2] };
    m[12] = dot3(ne, right); m[13] = dot3(ne, up); m[14] = dot3(ne, dir); m[15] = 1.0f;
    /* perspProjD3DStyle */
    float ys = 1.0f / tanf(0.5f * vfov), xs = ys / ((float)w / (float)h), zs = 1.0f / (z_far - z_near);
    float* p = o->view_to_clip;
    p[0] = xs; p[5] = ys; p[10] = z_far * zs; p[11] = 1.0f; p[14] = -z_near * z_far * zs;
    mat4_mul(o->world_to_view, o->view_to_clip, o->world_to_clip);
    mat4_inverse_d(o->world_to_clip, o->clip_to_world);
    o->camera_pos[0] = eye[0]; o->camera_pos[1] = eye[1]; o->camera_pos[2] = eye[2]; o->camera_pos[3] = 1.0f;
    const float* c = o->world_to_clip;   /* c[i*4+j] */
    set_plane(o->planes[0], -c[2],          -c[6],          -c[10],           c[14]);           /* NEAR   */
    set_plane(o->planes[1], -c[3] + c[2],   -c[7] + c[6],   -c[11] + c[10],   c[15] - c[14]);   /* FAR    */
    set_plane(o->planes[2], -c[3] - c[0],   -c[7] - c[4],   -c[11] - c[8],    c[15] + c[12]);   /* LEFT   */
    set_plane(o->planes[3], -c[3] + c[0],   -c[7] + c[4],   -c[11] + c[8],    c[15] - c[12]);   /* RIGHT  */
    set_plane(o->planes[4], -c[3] + c[1],   -c[7] + c[5],   -c[11] + c[9],    c[15] - c[13]);   /* TOP    */
    set_plane(o->planes[5], -c[3] - c[1],   -c[7] - c[5],   -c[11] - c[9],    c[15] + c[13]);   /* BOTTOM */
    o->viewport_x = 0; o->viewport_y = 0; o->viewport_w = w; o->viewport_h = h;
    /* IView::IsMirrored: det of the view's linear part < 0 (right = dir x up makes the
     * view space left-handed for a right-handed world, so this is 1 for LookAt views) */
    float det = right[0] * (up[1]*dir[2] - up[2]*dir[1]) - up[0] * (right[1]*dir[2] - right[2]*dir[1])
              + dir[0] * (right[1]*up[2] - right[2]*up[1]);
    o->mirrored = det < 0.0f;
    o->reverse_depth = 0;
}

/* CascadedShadowMap::SetupForPlanarViewStable for the one cascade the reference asks for
 * (Renderer.cpp:345-352) [DONUT-RECOLLECTION]: bounding sphere of the camera frustum slice
 * [0, maxShadowDistance]; its centre is snapped to whole shadow-map texels along the light's x / y axes so
 * that the map does not shimmer when the camera moves ("stable"); orthographic D3D-style projection over
 * [-radius, radius]^2 x [centre - zUp, centre + zDown] along the light direction. */
void orc_shadow_view(const vr_light* light, const vr_view* cam, const vr_shadow_params* p, vr_view* o)
{
    memset(o, 0, sizeof(*o));
    const float* W = cam->world_to_view;
    float fwd[3] = { W[2], W[6], W[10] };                       /* view +z in world space */
    float d = p->max_shadow_distance;
    float hw = d / cam->view_to_clip[0], hh = d / cam->view_to_clip[5];
    float r2 = hw * hw + hh * hh;                               /* far-plane half diagonal, squared */
    float c = (d * d + r2) / (2.0f * d);                        /* sphere through the apex and the far corners */
    if (c > d) c = d;
    float radius = sqrtf((d - c) * (d - c) + r2);
    float centre[3] = { cam->camera_pos[0] + fwd[0] * c, cam->camera_pos[1] + fwd[1] * c, cam->camera_pos[2] + fwd[2] * c };
    float zl[3] = { light->direction[0], light->direction[1], light->direction[2] };
    normalize3(zl);
    float up[3] = { 0.0f, 1.0f, 0.0f };
    if (fabsf(zl[1]) > 0.99f) { up[1] = 0.0f; up[2] = 1.0f; }
    float xl[3], yl[3];
    cross3(up, zl, xl); normalize3(xl);
    cross3(zl, xl, yl);
    float texel = (2.0f * radius) / (float)p->resolution;
    float cx = floorf(dot3(centre, xl) / texel) * texel, cy = floorf(dot3(centre, yl) / texel) * texel;
    float cz = dot3(centre, zl) - p->light_space_z_up;          /* near plane */
    float origin[3];
    for (int k = 0; k < 3; k++) origin[k] = (xl[k] * cx + yl[k] * cy) + zl[k] * cz;
    float* m = o->world_to_view;
    for (int i = 0; i < 3; i++) { m[i*4+0] = xl[i]; m[i*4+1] = yl[i]; m[i*4+2] = zl[i]; m[i*4+3] = 0.0f; }
    float no[3] = { -origin[0], -origin[1], -origin[2] };
    m[12] = dot3(no, xl); m[13] = dot3(no, yl); m[14] = dot3(no, zl); m[15] = 1.0f;
    float* q = o->view_to_clip;                                  /* orthoProjD3DStyle */
    q[0] = 1.0f / radius; q[5] = 1.0f / radius; q[10] = 1.0f / (p->light_space_z_up + p->light_space_z_down); q[15] = 1.0f;
    mat4_mul(o->world_to_view, o->view_to_clip, o->world_to_clip);
    mat4_inverse_d(o->world_to_clip, o->clip_to_world);
    o->camera_pos[0] = origin[0]; o->camera_pos[1] = origin[1]; o->camera_pos[2] = origin[2]; o->camera_pos[3] = 1.0f;
    const float* cc = o->world_to_clip;
    set_plane(o->planes[0], -cc[2],          -cc[6],          -cc[10],           cc[14]);
    set_plane(o->planes[1], -cc[3] + cc[2],  -cc[7] + cc[6],  -cc[11] + cc[10],  cc[15] - cc[14]);
    set_plane(o->planes[2], -cc[3] - cc[0],  -cc[7] - cc[4],  -cc[11] - cc[8],   cc[15] + cc[12]);
    set_plane(o->planes[3], -cc[3] + cc[0],  -cc[7] + cc[4],  -cc[11] + cc[8],   cc[15] - cc[12]);
    set_plane(o->planes[4], -cc[3] + cc[1],  -cc[7] + cc[5],  -cc[11] + cc[9],   cc[15] - cc[13]);
    set_plane(o->planes[5], -cc[3] - cc[1],  -cc[7] - cc[5],  -cc[11] - cc[9],   cc[15] + cc[13]);
    o->viewport_x = 0; o->viewport_y = 0; o->viewport_w = p->resolution; o->viewport_h = p->resolution;
    float det = xl[0] * (yl[1]*zl[2] - yl[2]*zl[1]) - yl[0] * (xl[1]*zl[2] - xl[2]*zl[1]) + zl[0] * (xl[1]*yl[2] - xl[2]*yl[1]);
    o->mirrored = det < 0.0f;
    o->reverse_depth = 0;
}

/* ------------------------------------------------------------------------- */
/* vertex stage: main_vs (terrain_vs.hlsl:10-62)                               */
/* ------------------------------------------------------------------------- */
typedef struct { float c[4]; float wpos[3]; } orc_vtx;   /* clip position, world position */

static void mul_row4(const float v[4], const float m[16], float o[4])
{
    for (int j = 0; j < 4; j++) o[j] = ((v[0]*m[0*4+j] + v[1]*m[1*4+j]) + v[2]*m[2*4+j]) + v[3]*m[3*4+j];
}
static void vertex_shader(const orc_terrain* t, const vr_view* v, float max_height,
                          const vr_instance* inst, int vx, int vz, orc_vtx* out)
{
    const float gs = (float)t->p.grid_size;
    const float half = (float)(t->p.grid_size / 2);
    /* grid mesh (TerrainPass.cpp:52-66): (w/halfSize, 0, h/halfSize) */
    float p[3] = { (float)(vx - t->p.grid_size / 2) / half, 0.0f, (float)(vz - t->p.grid_size / 2) / half };
    const float* M = inst->transform;
    float world[4];
    for (int r = 0; r < 3; r++) world[r] = ((M[r*4+0]*p[0] + M[r*4+1]*p[1]) + M[r*4+2]*p[2]) + M[r*4+3]*1.0f;   /* :44 */
    world[3] = 1.0f;
    float dx = world[0] - v->camera_pos[0], dz = world[2] - v->camera_pos[2];
    float distance = sqrtf(dx*dx + dz*dz);                                                      /* :46 */
    float gridExtents = 2.0f * sqrtf((M[0]*M[0] + M[4]*M[4]) + M[8]*M[8]);                      /* :47 */
    /* computeMorphK (:16-25); int(log2(x)) via the exponent (exact floor(log2)) */
    int lod = gridExtents > 0.0f ? ilog2_floor_f(gridExtents) : 0;
    lod = lod < 0 ? 0 : (lod > 11 ? 11 : lod);
    float start = t->lod_ranges[lod] * t->p.morph_start, end = t->lod_ranges[lod];
    float delta = end - start;
    float morphK = saturatef((distance - start) / delta);
    /* morphVertex (:10-14) */
    float gp[2] = { (p[0] + 1.0f) * 0.5f, (p[2] + 1.0f) * 0.5f };                               /* :49 */
    for (int k = 0; k < 2; k++) {
        float a = gp[k] * gs * 0.5f;
        float fr = (a - floorf(a)) * 2.0f / gs;
        int wi = k == 0 ? 0 : 2;
        world[wi] = world[wi] - fr * gridExtents * morphK;
    }
    /* sampleHeight (:27-33): SampleLevel(linearClamp, uv, 0.1).r * maxHeight */
    float halfSize = t->p.world_size * 0.5f;
    float u = (world[0] + halfSize) / t->p.world_size, w_ = (world[2] + halfSize) / t->p.world_size;
    float hv[3];
    tex_trilinear(&t->height, 0.1f, u, w_, hv);
    world[1] = hv[0] * max_height;                                                              /* :51 */
    float viewPos[4];
    mul_row4(world, v->world_to_view, viewPos);                                                 /* :60 */
    mul_row4(viewPos, v->view_to_clip, out->c);                                                 /* :61 */
    out->wpos[0] = world[0]; out->wpos[1] = world[1]; out->wpos[2] = world[2];
}
void orc_vertex(const orc_terrain* t, const vr_view* v, float max_height, const vr_instance* inst,
                int vx, int vz, float clip[4], float world[3])
{
    orc_vtx o; vertex_shader(t, v, max_height, inst, vx, vz, &o);
    memcpy(clip, o.c, 16); memcpy(world, o.wpos, 12);
}

/* ------------------------------------------------------------------------- */
/* rasteriser: D3D11 rules for the state of TerrainPass::CreateGraphicsPipeline */
/* (TerrainPass.cpp:460-485): back-face cull, front = CW unless mirrored, depth */
/* LessOrEqual + write, fill mode solid; top-left rule, pixel centres at +0.5,  */
/* 8 sub-pixel bits, clip to 0 <= z <= w, guard band, in-order depth test.      */
/* ------------------------------------------------------------------------- */
#define GUARD_BAND 100.0f
typedef struct {
    int w, h;
    float* depth; uint32_t* diffuse; uint32_t* specular; uint16_t* normals; uint16_t* emissive;
    const vr_partition* part;
    int depth_only;
    int wireframe;
} orc_target;

static int clip_poly(orc_vtx* poly, int n, int plane)
{
    /* Sutherland-Hodgman against one plane; intersection always computed from the
     * inside vertex towards the outside vertex so that both triangles sharing an
     * edge generate the identical new vertex. */
    orc_vtx out[12]; int m = 0;
    float d[12];
    for (int i = 0; i < n; i++) {
        const float* c = poly[i].c;
        switch (plane) {
        case 0: d[i] = c[2]; break;                                  /* z >= 0            */
        case 1: d[i] = GUARD_BAND * c[3] + c[0]; break;
        case 2: d[i] = GUARD_BAND * c[3] - c[0]; break;
        case 3: d[i] = GUARD_BAND * c[3] + c[1]; break;
        default: d[i] = GUARD_BAND * c[3] - c[1]; break;
        }
    }
    for (int i = 0; i < n; i++) {
        int j = (i + 1) % n;
        int ini = d[i] >= 0.0f, inj = d[j] >= 0.0f;
        if (ini) out[m++] = poly[i];
        if (ini != inj) {
            const orc_vtx* a = ini ? &poly[i] : &poly[j];
            const orc_vtx* b = ini ? &poly[j] : &poly[i];
            float da = ini ? d[i] : d[j], db = ini ? d[j] : d[i];
            float tt = da / (da - db);
            orc_vtx nv;
            for (int k = 0; k < 4; k++) nv.c[k] = a->c[k] + (b->c[k] - a->c[k]) * tt;
            for (int k = 0; k < 3; k++) nv.wpos[k] = a->wpos[k] + (b->wpos[k] - a->wpos[k]) * tt;
            out[m++] = nv;
        }
    }
    memcpy(poly, out, sizeof(orc_vtx) * m);
    return m;
}

typedef struct { int32_t X, Y; float z, iw; float wx, wz; } orc_sv;   /* snapped screen vertex */

static void to_screen(const orc_vtx* v, const vr_view* view, orc_sv* o)
{
    float iw = 1.0f / v->c[3];
    float nx = v->c[0] * iw, ny = v->c[1] * iw;
    float sx = (nx * 0.5f + 0.5f) * (float)view->viewport_w + (float)view->viewport_x;
    float sy = (ny * -0.5f + 0.5f) * (float)view->viewport_h + (float)view->viewport_y;
    o->X = (int32_t)floorf(sx * 256.0f + 0.5f);
    o->Y = (int32_t)floorf(sy * 256.0f + 0.5f);
    o->z = v->c[2] * iw; o->iw = iw;
    o->wx = v->wpos[0]; o->wz = v->wpos[2];
}

static inline int64_t edge_fn(const orc_sv* a, const orc_sv* b, int64_t px, int64_t py)
{
    return (int64_t)(b->X - a->X) * (py - a->Y) - (int64_t)(b->Y - a->Y) * (px - a->X);
}
static inline int top_left(const orc_sv* a, const orc_sv* b)
{
    int32_t dx = b->X - a->X, dy = b->Y - a->Y;
    return (dy < 0) || (dy == 0 && dx > 0);
}

/* Fixed-function interpolation, raster model revision 3: per-triangle PLANE EQUATIONS.
 * Depth z, q = 1/w and the attributes over w (wx/w, wz/w) are affine functions of the pixel
 * position, so each is P(x, y) = p0 + px * (x - ax) + py * (y - ay) about an anchor pixel (ax, ay)
 * of the triangle (the first pixel of its viewport-clamped bounding box).  The coefficients are
 * set up ONCE per triangle in double precision from the exact integer edge functions and rounded to
 * float; a pixel evaluates fmaf(py, dy, fmaf(px, dx, p0)) with dx, dy small exact integers.
 * Perspective correction: attribute = (attribute/w)(x, y) * (1 / q(x, y)), IEEE division.
 * The screen-space derivatives of the attributes - what Texture2D::Sample's implicit LOD is
 * computed from here (hardware takes quad finite differences; D3D leaves the method to the
 * implementation) - are the analytic derivatives of that quotient:
 *   d(N/q)/dx = (N_x - (N/q) q_x) / q. */
typedef struct { float p0, px, py; } orc_plane;
typedef struct { int ax, ay; orc_plane z, q, nx, nz; } orc_planes;

static orc_plane plane_of(const double l[3], const double lx[3], const double ly[3], double a0, double a1, double a2)
{
    orc_plane p;
    p.p0 = (float)((l[0] * a0 + l[1] * a1) + l[2] * a2);
    p.px = (float)((lx[0] * a0 + lx[1] * a1) + lx[2] * a2);
    p.py = (float)((ly[0] * a0 + ly[1] * a1) + ly[2] * a2);
    return p;
}
static inline float plane_at(const orc_plane* p, float dx, float dy) { return fmaf(p->py, dy, fmaf(p->px, dx, p->p0)); }

/* s0, s1, s2 in clockwise order, area2 = twice the (positive) area in 24.8 x 24.8 units */
static void planes_setup(orc_planes* P, const orc_sv* s0, const orc_sv* s1, const orc_sv* s2, int64_t area2, int ax, int ay)
{
    P->ax = ax; P->ay = ay;
    const double inv = 1.0 / (double)area2;
    const int64_t PX = (int64_t)ax * 256 + 128, PY = (int64_t)ay * 256 + 128;
    /* barycentrics of the anchor's centre and their steps per pixel: l1 = E(s2,s0) / area2, l2 = E(s0,s1) / area2 */
    double l[3], lx[3], ly[3];
    l[1] = (double)edge_fn(s2, s0, PX, PY) * inv; l[2] = (double)edge_fn(s0, s1, PX, PY) * inv; l[0] = (1.0 - l[1]) - l[2];
    lx[1] = (double)(-(int64_t)(s0->Y - s2->Y) * 256) * inv; lx[2] = (double)(-(int64_t)(s1->Y - s0->Y) * 256) * inv; lx[0] = (0.0 - lx[1]) - lx[2];
    ly[1] = (double)((int64_t)(s0->X - s2->X) * 256) * inv;  ly[2] = (double)((int64_t)(s1->X - s0->X) * 256) * inv;  ly[0] = (0.0 - ly[1]) - ly[2];
    P->z = plane_of(l, lx, ly, (double)s0->z, (double)s1->z, (double)s2->z);
    P->q = plane_of(l, lx, ly, (double)s0->iw, (double)s1->iw, (double)s2->iw);
    P->nx = plane_of(l, lx, ly, (double)s0->wx * (double)s0->iw, (double)s1->wx * (double)s1->iw, (double)s2->wx * (double)s2->iw);
    P->nz = plane_of(l, lx, ly, (double)s0->wz * (double)s0->iw, (double)s1->wz * (double)s1->iw, (double)s2->wz * (double)s2->iw);
}

typedef struct { float wx, wz, dwxdx, dwzdx, dwxdy, dwzdy; } orc_attr;
static orc_attr interp(const orc_planes* P, float dx, float dy)
{
    float q = plane_at(&P->q, dx, dy);
    float r = 1.0f / q;
    orc_attr a;
    a.wx = plane_at(&P->nx, dx, dy) * r;
    a.wz = plane_at(&P->nz, dx, dy) * r;
    a.dwxdx = fmaf(-a.wx, P->q.px, P->nx.px) * r; a.dwzdx = fmaf(-a.wz, P->q.px, P->nz.px) * r;
    a.dwxdy = fmaf(-a.wx, P->q.py, P->nx.py) * r; a.dwzdy = fmaf(-a.wz, P->q.py, P->nz.py) * r;
    return a;
}

/* Debug tap for tests/test_oracle_cpu.py (no part of the model): when set, every shaded pixel also leaves the implicit
 * level of detail main_ps's Sample calls used (before the sampler's clamp) and its interpolated world xz. */
static float* g_dbg_plane = NULL;       /* w * h * 3 floats: lod, world x, world z */
void orc_debug_set_pixel_plane(float* plane) { g_dbg_plane = plane; }
int  orc_model_revision(void) { return ORC_MODEL_REVISION; }

/* main_ps (terrain_ps.hlsl:45-82) for one pixel; p = world xz at the pixel centre with its
 * screen-space derivatives (for the implicit LOD of Sample). */
static void pixel_shader(const orc_terrain* t, orc_attr p,
                         uint32_t* diffuse, uint32_t* specular, uint16_t normals[4], uint16_t emissive[4], float* dbg)
{
    float halfSize = t->p.world_size * 0.5f, ws = t->p.world_size;
    float u = (p.wx + halfSize) / ws, v = (p.wz + halfSize) / ws;                      /* :12-13,20-21 */
    float dudx = p.dwxdx / ws, dvdx = p.dwzdx / ws, dudy = p.dwxdy / ws, dvdy = p.dwzdy / ws;
    float lod_h = lod_from_derivs(dudx, dvdx, dudy, dvdy, t->height.w[0], t->height.h[0]);
    float lod_c = lod_from_derivs(dudx, dvdx, dudy, dvdy, t->albedo.w[0], t->albedo.h[0]);
    if (dbg) { dbg[0] = lod_h; dbg[1] = p.wx; dbg[2] = p.wz; }
    const float offset = 0.1f;                                                         /* :59 */
    float a[3], b[3];
    tex_trilinear(&t->height, lod_h, u + offset, v + 0.0f, a);
    tex_trilinear(&t->height, lod_h, u + (-offset), v + 0.0f, b);
    float hDx = a[0] - b[0];                                                           /* :60 */
    tex_trilinear(&t->height, lod_h, u + 0.0f, v + offset, a);
    tex_trilinear(&t->height, lod_h, u + 0.0f, v + (-offset), b);
    float hDy = a[0] - b[0];                                                           /* :61 */
    float n[3] = { -hDx, 2.0f * offset, -hDy };                                        /* :63 */
    float inv = 1.0f / sqrtf(fmaf(n[2], n[2], fmaf(n[0], n[0], n[1] * n[1])));         /* normalize(): x / length(x) */
    n[0] *= inv; n[1] *= inv; n[2] *= inv;
    float col[3];
    tex_trilinear(&t->albedo, lod_c, u, v, col);                                       /* :68 */
    /* :73-81 -> SRGBA8 / SRGBA8 / RGBA16_SNORM / RGBA16_FLOAT render targets */
    *diffuse = (uint32_t)orc_linear_to_srgb8(col[0]) | ((uint32_t)orc_linear_to_srgb8(col[1]) << 8)
             | ((uint32_t)orc_linear_to_srgb8(col[2]) << 16) | ((uint32_t)unorm8(1.0f) << 24);
    uint32_t s = orc_linear_to_srgb8(1.0f * 0.01f);
    *specular = s | (s << 8) | (s << 16) | ((uint32_t)unorm8(1.0f) << 24);
    normals[0] = snorm16(n[0]); normals[1] = snorm16(n[1]); normals[2] = snorm16(n[2]); normals[3] = snorm16(1.0f);
    emissive[0] = emissive[1] = emissive[2] = emissive[3] = orc_float_to_half(0.0f);
}

static inline int owns_pixel(const vr_partition* part, int x, int y)
{
    if (!part || part->world_size <= 1) return 1;
    return ((x / VR_OWNER_TILE + y / VR_OWNER_TILE) % part->world_size) == part->rank;
}

/* One covered pixel of a triangle: depth from the triangle's plane at the pixel centre, depth clip,
 * LessOrEqual test (in draw order), then main_ps. */
typedef struct {
    const orc_terrain* t; orc_target* tg;
    const orc_sv *s0, *s1, *s2;
    orc_planes P;
} orc_frag_ctx;

static void fragment(const orc_frag_ctx* f, int x, int y)
{
    orc_target* tg = f->tg;
    if (!owns_pixel(tg->part, x, y)) return;
    const float dx = (float)(x - f->P.ax), dy = (float)(y - f->P.ay);
    float z = plane_at(&f->P.z, dx, dy);
    if (!(z >= 0.0f && z <= 1.0f)) return;                    /* depth clip */
    size_t idx = (size_t)y * tg->w + x;
    if (!(z <= tg->depth[idx])) return;                       /* ComparisonFunc::LessOrEqual (TerrainPass.cpp:482) */
    tg->depth[idx] = z + 0.0f;
    if (tg->depth_only) return;
    orc_attr p = interp(&f->P, dx, dy);
    pixel_shader(f->t, p, &tg->diffuse[idx], &tg->specular[idx], &tg->normals[idx*4], &tg->emissive[idx*4],
                 g_dbg_plane ? g_dbg_plane + idx * 3 : NULL);
}

static inline int64_t floor_div(int64_t num, int64_t den)
{
    if (den < 0) { num = -num; den = -den; }
    int64_t q = num / den;
    if (num % den < 0) q--;
    return q;
}

/* Aliased line a-b in 24.8 fixed point [DONUT/D3D-RECOLLECTION: stands in for the diamond-exit
 * rule, whose end-point cases this image has no way to pin].  X-major lines (|dX| >= |dY|) take
 * one pixel in every column whose centre lies in [min X, max X): the pixel containing the exact
 * line point at that column centre; Y-major lines likewise per row.  The rule is symmetric in
 * a <-> b, so two triangles sharing an edge draw the same pixels. */
static void wire_edge(const orc_frag_ctx* f, const orc_sv* a, const orc_sv* b, int vx0, int vy0, int vx1, int vy1)
{
    int64_t dX = (int64_t)b->X - a->X, dY = (int64_t)b->Y - a->Y;
    if (dX == 0 && dY == 0) return;
    int64_t adX = dX < 0 ? -dX : dX, adY = dY < 0 ? -dY : dY;
    if (adX >= adY) {
        int32_t lo = a->X < b->X ? a->X : b->X, hi = a->X < b->X ? b->X : a->X;
        int p0 = (lo - 128 + 255) >> 8, p1 = ((hi - 128 + 255) >> 8) - 1;
        if (p0 < vx0) p0 = vx0;
        if (p1 > vx1) p1 = vx1;
        for (int px = p0; px <= p1; px++) {
            int64_t PX = (int64_t)px * 256 + 128;
            int64_t py = floor_div((int64_t)a->Y * dX + (PX - a->X) * dY, dX * 256);
            if (py < vy0 || py > vy1) continue;
            fragment(f, px, (int)py);
        }
    } else {
        int32_t lo = a->Y < b->Y ? a->Y : b->Y, hi = a->Y < b->Y ? b->Y : a->Y;
        int p0 = (lo - 128 + 255) >> 8, p1 = ((hi - 128 + 255) >> 8) - 1;
        if (p0 < vy0) p0 = vy0;
        if (p1 > vy1) p1 = vy1;
        for (int py = p0; py <= p1; py++) {
            int64_t PY = (int64_t)py * 256 + 128;
            int64_t px = floor_div((int64_t)a->X * dY + (PY - a->Y) * dX, dY * 256);
            if (px < vx0 || px > vx1) continue;
            fragment(f, (int)px, py);
        }
    }
}

static void raster_triangle(const orc_terrain* t, const vr_view* view, orc_target* tg,
                            const orc_vtx* a, const orc_vtx* b, const orc_vtx* c)
{
    orc_sv s0, s1, s2;
    to_screen(a, view, &s0); to_screen(b, view, &s1); to_screen(c, view, &s2);
    int64_t area2 = (int64_t)(s1.X - s0.X) * (s2.Y - s0.Y) - (int64_t)(s2.X - s0.X) * (s1.Y - s0.Y);
    if (area2 == 0) return;
    int cw = area2 > 0;                               /* clockwise on the (y-down) render target */
    int front = view->mirrored ? !cw : cw;            /* frontCounterClockwise = IsMirrored (TerrainPass.cpp:301,474) */
    if (!front) return;                               /* RasterCullMode::Back (TerrainPass.cpp:211,475) */
    if (!cw) { orc_sv tmp = s1; s1 = s2; s2 = tmp; area2 = -area2; }
    int32_t minX = s0.X < s1.X ? s0.X : s1.X; if (s2.X < minX) minX = s2.X;
    int32_t maxX = s0.X > s1.X ? s0.X : s1.X; if (s2.X > maxX) maxX = s2.X;
    int32_t minY = s0.Y < s1.Y ? s0.Y : s1.Y; if (s2.Y < minY) minY = s2.Y;
    int32_t maxY = s0.Y > s1.Y ? s0.Y : s1.Y; if (s2.Y > maxY) maxY = s2.Y;
    int vx0 = view->viewport_x, vy0 = view->viewport_y;
    int vx1 = vx0 + view->viewport_w - 1, vy1 = vy0 + view->viewport_h - 1;
    if (vx1 > tg->w - 1) vx1 = tg->w - 1;
    if (vy1 > tg->h - 1) vy1 = tg->h - 1;
    if (vx0 < 0) vx0 = 0;
    if (vy0 < 0) vy0 = 0;
    orc_frag_ctx fc = { t, tg, &s0, &s1, &s2, { 0 } };
    if (tg->wireframe) {
        /* RasterFillMode::Wireframe (TerrainPass.cpp:476): the three edges of every triangle that
         * survives culling, as aliased lines; each line pixel is shaded as a sample of the
         * triangle's plane at that pixel centre.  A line pixel is the one that contains the line
         * point: the box of candidate pixels (and the planes' anchor) is floor(min) .. floor(max). */
        int wx0 = minX >> 8, wy0 = minY >> 8, wx1 = maxX >> 8, wy1 = maxY >> 8;
        if (wx0 < vx0) wx0 = vx0;
        if (wy0 < vy0) wy0 = vy0;
        if (wx1 > vx1) wx1 = vx1;
        if (wy1 > vy1) wy1 = vy1;
        if (wx0 > wx1 || wy0 > wy1) return;
        planes_setup(&fc.P, &s0, &s1, &s2, area2, wx0, wy0);
        wire_edge(&fc, &s0, &s1, vx0, vy0, vx1, vy1);
        wire_edge(&fc, &s1, &s2, vx0, vy0, vx1, vy1);
        wire_edge(&fc, &s2, &s0, vx0, vy0, vx1, vy1);
        return;
    }
    /* pixels whose centre (px*256+128) lies inside the bounding box */
    int x0 = (minX - 128 + 255) >> 8, x1 = (maxX - 128) >> 8;
    int y0 = (minY - 128 + 255) >> 8, y1 = (maxY - 128) >> 8;
    if (x0 < vx0) x0 = vx0;
    if (y0 < vy0) y0 = vy0;
    if (x1 > vx1) x1 = vx1;
    if (y1 > vy1) y1 = vy1;
    if (x0 > x1 || y0 > y1) return;
    planes_setup(&fc.P, &s0, &s1, &s2, area2, x0, y0);
    int b0 = top_left(&s1, &s2) ? 0 : 1, b1 = top_left(&s2, &s0) ? 0 : 1, b2 = top_left(&s0, &s1) ? 0 : 1;
    for (int y = y0; y <= y1; y++) for (int x = x0; x <= x1; x++) {
        int64_t PX = (int64_t)x * 256 + 128, PY = (int64_t)y * 256 + 128;
        int64_t E0 = edge_fn(&s1, &s2, PX, PY), E1 = edge_fn(&s2, &s0, PX, PY), E2 = edge_fn(&s0, &s1, PX, PY);
        if (E0 - b0 < 0 || E1 - b1 < 0 || E2 - b2 < 0) continue;
        fragment(&fc, x, y);
    }
}

static void draw_triangle(const orc_terrain* t, const vr_view* view, orc_target* tg,
                          const orc_vtx* v0, const orc_vtx* v1, const orc_vtx* v2)
{
    const orc_vtx* tv[3] = { v0, v1, v2 };
    /* trivial reject against the six clip planes */
    int out_l = 1, out_r = 1, out_b = 1, out_t = 1, out_n = 1, out_f = 1, need_near = 0, need_guard = 0;
    for (int i = 0; i < 3; i++) {
        const float* c = tv[i]->c;
        out_l &= c[0] < -c[3]; out_r &= c[0] > c[3];
        out_b &= c[1] < -c[3]; out_t &= c[1] > c[3];
        out_n &= c[2] < 0.0f;  out_f &= c[2] > c[3];
        need_near |= c[2] < 0.0f;
    }
    if (out_l || out_r || out_b || out_t || out_n || out_f) return;
    orc_vtx poly[12]; int n = 3;
    poly[0] = *v0; poly[1] = *v1; poly[2] = *v2;
    if (need_near) { n = clip_poly(poly, n, 0); if (n < 3) return; }
    for (int i = 0; i < n; i++) {
        const float* c = poly[i].c;
        float g = GUARD_BAND * c[3];
        need_guard |= (c[0] < -g) || (c[0] > g) || (c[1] < -g) || (c[1] > g);
    }
    if (need_guard) for (int pl = 1; pl <= 4; pl++) { n = clip_poly(poly, n, pl); if (n < 3) return; }
    for (int i = 1; i + 1 < n; i++) raster_triangle(t, view, tg, &poly[0], &poly[i], &poly[i+1]);
}

void orc_gbuffer_clear(int w, int h, float* depth, uint32_t* diffuse, uint32_t* specular, uint16_t* normals, uint16_t* emissive)
{
    size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; i++) depth[i] = 1.0f;
    if (diffuse) memset(diffuse, 0, n * 4);
    if (specular) memset(specular, 0, n * 4);
    if (normals) memset(normals, 0, n * 8);
    if (emissive) memset(emissive, 0, n * 8);
}

int orc_render(orc_terrain* t, const vr_view* v, const vr_render_params* rp, const vr_partition* part,
               int w, int h, float* depth, uint32_t* diffuse, uint32_t* specular, uint16_t* normals, uint16_t* emissive)
{
    orc_target tg = { w, h, depth, diffuse, specular, normals, emissive, part, rp->depth_only, rp->wireframe };
    int cap = t->p.max_instances;
    vr_instance* inst = (vr_instance*)malloc(sizeof(vr_instance) * cap);
    int n;
    if (rp->lock_view && t->have_selection) {
        /* lockView: keep m_SelectedNodes and the instance buffer of the last unlocked frame (TerrainPass.cpp:173,191-197) */
        n = t->num_selected;
        for (int i = 0; i < n && i < cap; i++) update_transform(t->selected[i], &inst[i]);
    } else {
        n = orc_select(t, v, rp->max_height, 0, NULL, inst, cap);
    }
    if (n > cap) n = cap;
    const int G = t->p.grid_size, S = G + 1;
    orc_vtx* verts = (orc_vtx*)malloc(sizeof(orc_vtx) * S * S);
    for (int i = 0; i < n; i++) {
        for (int vz = 0; vz < S; vz++) for (int vx = 0; vx < S; vx++)
            vertex_shader(t, v, rp->max_height, &inst[i], vx, vz, &verts[vz * S + vx]);
        /* index buffer (TerrainPass.cpp:68-87): per cell (BL,TL,TR) then (BL,TR,BR); row = z, column = x */
        for (int ci = 0; ci < G; ci++) for (int cj = 0; cj < G; cj++) {
            const orc_vtx* bl = &verts[ci * S + cj];
            const orc_vtx* tl = &verts[(ci + 1) * S + cj];
            const orc_vtx* tr = &verts[(ci + 1) * S + cj + 1];
            const orc_vtx* br = &verts[ci * S + cj + 1];
            draw_triangle(t, v, &tg, bl, tl, tr);
            draw_triangle(t, v, &tg, bl, tr, br);
        }
    }
    free(verts); free(inst);
    return n;
}

/* ------------------------------------------------------------------------- */
/* deferred lighting [DONUT-RECOLLECTION of render::DeferredLightingPass,     */
/* deferred_lighting_cs.hlsl, lighting.hlsli ShadeSurface, brdf.hlsli          */
/* GGX_AnalyticalLights_times_NdotL]; inputs per Renderer.cpp:417-428.         */
/* Shadow term (row f1): 4x4 tent PCF on the terrain shadow map, see shadow_factor. */
/* The area-light correction slerp(L, R, saturate(halfAngle/angle(L,R))) is     */
/* restated in closed form (no acos/sin per pixel):                            */
/*   angle <= half  -> R;  else  L*(cosH - cosT*sinH/sinT) + R*(sinH/sinT).     */
/* ------------------------------------------------------------------------- */
#define ORC_PI 3.14159265358979323846f
#define ORC_INV_PI 0.318309886183790671538f

typedef struct { float cosH, sinH, tanH; } orc_light_consts;

/* What DirectionalLight::shadowMap gives the lighting pass (Renderer.cpp:336, 427) [DONUT-RECOLLECTION of
 * EvaluateShadowGather16]: world -> light clip -> uv; outside the map: outOfBoundsShadow; else the 4x4 texel
 * footprint around the sample, each texel compared LessEqual (receiver depth - bias <= stored depth),
 * weighted [1-fx, 1, 1, fx] x [1-fy, 1, 1, fy], normalised by 9. */
typedef struct { const vr_view* light_view; const float* depth; int res; int light_index; float depth_bias; } orc_shadow;

static float shadow_factor(const orc_shadow* s, const float wp[3], float out_of_bounds)
{
    float p4[4] = { wp[0], wp[1], wp[2], 1.0f }, c[4];
    mul_row4(p4, s->light_view->world_to_clip, c);
    float xc = c[0] / c[3], yc = c[1] / c[3], zc = c[2] / c[3];
    float u = xc * 0.5f + 0.5f, v = 0.5f - yc * 0.5f;
    if (!(u >= 0.0f && u <= 1.0f && v >= 0.0f && v <= 1.0f && zc >= 0.0f && zc <= 1.0f)) return out_of_bounds;
    float z = zc - s->depth_bias;
    float tx = u * (float)s->res - 0.5f, ty = v * (float)s->res - 0.5f;
    float fxl = floorf(tx), fyl = floorf(ty);
    float fx = tx - fxl, fy = ty - fyl;
    int ix = (int)fxl - 1, iy = (int)fyl - 1;
    float wx[4] = { 1.0f - fx, 1.0f, 1.0f, fx }, wy[4] = { 1.0f - fy, 1.0f, 1.0f, fy };
    float sum = 0.0f;
    for (int j = 0; j < 4; j++) {
        int y = iy + j; y = y < 0 ? 0 : (y > s->res - 1 ? s->res - 1 : y);
        float row = 0.0f;
        for (int i = 0; i < 4; i++) {
            int x = ix + i; x = x < 0 ? 0 : (x > s->res - 1 ? s->res - 1 : x);
            float lit = z <= s->depth[(size_t)y * s->res + x] ? 1.0f : 0.0f;
            row = row + lit * wx[i];
        }
        sum = sum + row * wy[j];
    }
    return sum / 9.0f;
}

static void shade_pixel(const vr_view* v, int w, int h, int px, int py,
                        float depth, uint32_t diff, uint32_t spec, const uint16_t nrm[4], const uint16_t emi[4],
                        const vr_light* lights, const orc_light_consts* lc, int nl,
                        const float amb_top[3], const float amb_bot[3], const orc_shadow* shadow, float out[4])
{
    float albedo[3] = { g_srgb_lut[diff & 255u], g_srgb_lut[(diff >> 8) & 255u], g_srgb_lut[(diff >> 16) & 255u] };
    float F0[3] = { g_srgb_lut[spec & 255u], g_srgb_lut[(spec >> 8) & 255u], g_srgb_lut[(spec >> 16) & 255u] };
    float occlusion = (float)(spec >> 24) / 255.0f;
    float N[3] = { snorm16_decode(nrm[0]), snorm16_decode(nrm[1]), snorm16_decode(nrm[2]) };
    float rough = snorm16_decode(nrm[3]);
    float E[3] = { orc_half_to_float(emi[0]), orc_half_to_float(emi[1]), orc_half_to_float(emi[2]) };
    /* ReconstructWorldPosition: window -> clip -> world */
    float sx = 2.0f / (float)w, sy = -2.0f / (float)h;
    float clip[4] = { ((float)px + 0.5f) * sx + -1.0f, ((float)py + 0.5f) * sy + 1.0f, depth, 1.0f };
    float wp4[4];
    mul_row4(clip, v->clip_to_world, wp4);
    float wp[3] = { wp4[0] / wp4[3], wp4[1] / wp4[3], wp4[2] / wp4[3] };
    float d[3] = { wp[0] - v->camera_pos[0], wp[1] - v->camera_pos[1], wp[2] - v->camera_pos[2] };
    float dl = 1.0f / sqrtf(dot3(d, d));
    float vi[3] = { d[0] * dl, d[1] * dl, d[2] * dl };      /* viewIncident */
    float V[3] = { -vi[0], -vi[1], -vi[2] };
    float diffuseTerm[3] = { 0, 0, 0 }, specularTerm[3] = { 0, 0, 0 };
    float NdotVi = dot3(vi, N);
    float two = 2.0f * NdotVi;
    float R[3] = { vi[0] - N[0] * two, vi[1] - N[1] * two, vi[2] - N[2] * two };   /* reflect(viewIncident, N) */
    float NdotV = saturatef(dot3(N, V));
    float alpha = fmaxx(0.01f, rough * rough);
    float a2 = alpha * alpha;
    float kk = ((rough + 1.0f) * (rough + 1.0f)) / 8.0f;
    for (int i = 0; i < nl; i++) {
        const vr_light* L_ = &lights[i];
        float Lin[3], irr;
        float cosH = lc[i].cosH, sinH = lc[i].sinH, tanH = lc[i].tanH;
        if (L_->type == VR_LIGHT_DIRECTIONAL) {
            Lin[0] = L_->direction[0]; Lin[1] = L_->direction[1]; Lin[2] = L_->direction[2];
            irr = L_->intensity;
        } else {   /* spot or point (ShadeSurface) */
            float lts[3] = { wp[0] - L_->position[0], wp[1] - L_->position[1], wp[2] - L_->position[2] };
            float dist = sqrtf(dot3(lts, lts));
            float rd = 1.0f / dist;
            Lin[0] = lts[0] * rd; Lin[1] = lts[1] * rd; Lin[2] = lts[2] * rd;
            float att = 1.0f;
            if (L_->angular_size_or_inv_range > 0.0f) {
                float a = dist * L_->angular_size_or_inv_range;
                float aa = a * a;
                float s = saturatef(1.0f - aa * aa);
                att = s * s;
                if (att == 0.0f) continue;
            }
            float spotlight = 1.0f;
            if (L_->type == VR_LIGHT_SPOT) {
                float LdotD = fminx(fmaxx(dot3(Lin, L_->direction), -1.0f), 1.0f);
                float directionAngle = acosf(LdotD);
                float ts = saturatef((directionAngle - L_->inner_angle) / (L_->outer_angle - L_->inner_angle));
                spotlight = 1.0f - ts * ts * (3.0f - 2.0f * ts);        /* 1 - smoothstep(inner, outer, angle) */
                if (spotlight == 0.0f) continue;
            }
            if (L_->radius > 0.0f) {
                /* spherical source: half angle atan(min(radius / distance, 1)); its sin/cos/tan in closed form */
                float x = fminx(L_->radius * rd, 1.0f);
                float halfAng = atanf(x);
                float radianceTimesPi = L_->intensity / (L_->radius * L_->radius);
                irr = radianceTimesPi * (halfAng * halfAng);
                tanH = x; cosH = 1.0f / sqrtf(1.0f + x * x); sinH = x * cosH;
            } else {
                irr = L_->intensity * (rd * rd);
            }
            irr = irr * (spotlight * att);
        }
        if (shadow && i == shadow->light_index) {
            float sf = shadow_factor(shadow, wp, L_->out_of_bounds_shadow);
            if (sf == 0.0f) continue;
            irr = irr * sf;
        }
        float L[3] = { -Lin[0], -Lin[1], -Lin[2] };
        /* diffuse: Lambert */
        float NdotLd = fmaxx(dot3(N, L), 0.0f);
        float kd = (NdotLd * ORC_INV_PI) * irr;
        /* specular: GGX with area-light correction */
        float cosT = fminx(fmaxx(dot3(R, L), -1.0f), 1.0f);
        float CL[3];
        if (cosT >= cosH) { CL[0] = R[0]; CL[1] = R[1]; CL[2] = R[2]; }
        else {
            float sinT = sqrtf(fmaxx(1.0f - cosT * cosT, 1e-12f));
            float k2 = sinH / sinT;
            float k1 = cosH - cosT * k2;
            for (int c = 0; c < 3; c++) CL[c] = L[c] * k1 + R[c] * k2;
        }
        float Hv[3] = { CL[0] + V[0], CL[1] + V[1], CL[2] + V[2] };
        float hl2 = dot3(Hv, Hv);
        float hs = hl2 > 0.0f ? 1.0f / sqrtf(hl2) : 0.0f;
        Hv[0] *= hs; Hv[1] *= hs; Hv[2] *= hs;
        float NdotH = saturatef(dot3(N, Hv)), NdotL = saturatef(dot3(N, CL)), VdotH = saturatef(dot3(V, Hv));
        float corrAlpha = saturatef(alpha + 0.5f * tanH);
        float sn = alpha / corrAlpha; sn = sn * sn;
        float dd = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
        float D = (a2 / (ORC_PI * (dd * dd))) * sn;
        float G = 1.0f / ((NdotL * (1.0f - kk) + kk) * (NdotV * (1.0f - kk) + kk));
        float om = 1.0f - VdotH;
        float om2 = om * om;
        float fw = (om2 * om2) * om;
        float ks = (((D * G) * NdotL) / 4.0f) * irr;
        for (int c = 0; c < 3; c++) {
            float F = F0[c] + (1.0f - F0[c]) * fw;
            diffuseTerm[c] = diffuseTerm[c] + (albedo[c] * kd) * L_->color[c];
            specularTerm[c] = specularTerm[c] + (F * ks) * L_->color[c];
        }
    }
    float tt = N[1] * 0.5f + 0.5f;
    for (int c = 0; c < 3; c++) {
        float amb = amb_bot[c] + (amb_top[c] - amb_bot[c]) * tt;        /* lerp(bottom, top, N.y*0.5+0.5) */
        diffuseTerm[c] = diffuseTerm[c] + (amb * albedo[c]) * occlusion;
        specularTerm[c] = specularTerm[c] + (amb * F0[c]) * occlusion;
        out[c] = (diffuseTerm[c] + specularTerm[c]) + E[c];
    }
    out[3] = 0.0f;
}

/* optional shadow binding of the next orc_deferred* call (set by orc_deferred_set_shadow, test harness is single-threaded) */
static const orc_shadow* g_shadow = NULL;
static orc_shadow g_shadow_store;
void orc_deferred_set_shadow(const vr_view* light_view, const float* shadow_depth, int res, int light_index, float depth_bias)
{
    if (!light_view || !shadow_depth) { g_shadow = NULL; return; }
    g_shadow_store.light_view = light_view; g_shadow_store.depth = shadow_depth; g_shadow_store.res = res;
    g_shadow_store.light_index = light_index; g_shadow_store.depth_bias = depth_bias;
    g_shadow = &g_shadow_store;
}

static orc_light_consts* light_consts(const vr_light* lights, int n)
{
    orc_light_consts* lc = (orc_light_consts*)malloc(sizeof(orc_light_consts) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        double half = lights[i].type == VR_LIGHT_DIRECTIONAL ? 0.5 * (double)lights[i].angular_size_or_inv_range : 0.0;
        lc[i].cosH = (float)cos(half); lc[i].sinH = (float)sin(half); lc[i].tanH = (float)tan(half);
    }
    return lc;
}

void orc_deferred_f32(const vr_view* v, int w, int h, const float* depth, const uint32_t* diffuse, const uint32_t* specular,
                      const uint16_t* normals, const uint16_t* emissive, const vr_light* lights, int nl,
                      const float amb_top[3], const float amb_bottom[3], float* rgba)
{
    init_tables();
    orc_light_consts* lc = light_consts(lights, nl);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        size_t i = (size_t)y * w + x;
        shade_pixel(v, w, h, x, y, depth[i], diffuse[i], specular[i], &normals[i*4], &emissive[i*4], lights, lc, nl, amb_top, amb_bottom, g_shadow, &rgba[i*4]);
    }
    free(lc);
}
void orc_deferred(const vr_view* v, int w, int h, const float* depth, const uint32_t* diffuse, const uint32_t* specular,
                  const uint16_t* normals, const uint16_t* emissive, const vr_light* lights, int nl,
                  const float amb_top[3], const float amb_bottom[3], uint16_t* hdr)
{
    init_tables();
    orc_light_consts* lc = light_consts(lights, nl);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        size_t i = (size_t)y * w + x;
        float o[4];
        shade_pixel(v, w, h, x, y, depth[i], diffuse[i], specular[i], &normals[i*4], &emissive[i*4], lights, lc, nl, amb_top, amb_bottom, g_shadow, o);
        for (int c = 0; c < 4; c++) hdr[i*4+c] = orc_float_to_half(o[c]);
    }
    free(lc);
}

/* ------------------------------------------------------------------------- */
/* synthetic inputs (media/ is git-ignored in the reference: .gitignore:36,52)  */
/* integer-only so that any implementation reproduces them bit for bit.        */
/* ------------------------------------------------------------------------- */
static inline uint32_t hash32(uint32_t x, uint32_t y, uint32_t s)
{
    uint32_t h = (x * 0x9E3779B1u) ^ (y * 0x85EBCA77u) ^ (s * 0xC2B2AE3Du);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
static uint32_t value_noise16(uint32_t x, uint32_t y, uint32_t period, uint32_t seed)
{
    uint32_t ix = x / period, iy = y / period;
    uint32_t fx = ((x % period) << 16) / period, fy = ((y % period) << 16) / period;   /* 16.16 fraction */
    uint64_t sx = (((uint64_t)fx * fx) >> 16) * (3u * 65536u - 2u * fx) >> 16;          /* smoothstep  */
    uint64_t sy = (((uint64_t)fy * fy) >> 16) * (3u * 65536u - 2u * fy) >> 16;
    uint64_t h00 = hash32(ix, iy, seed) >> 16, h10 = hash32(ix + 1, iy, seed) >> 16;
    uint64_t h01 = hash32(ix, iy + 1, seed) >> 16, h11 = hash32(ix + 1, iy + 1, seed) >> 16;
    uint64_t top = (h00 * (65536u - sx) + h10 * sx) >> 16;
    uint64_t bot = (h01 * (65536u - sx) + h11 * sx) >> 16;
    return (uint32_t)((top * (65536u - sy) + bot * sy) >> 16);
}
void orc_synth_heightmap(int size, uint32_t seed, uint8_t* out)
{
    static const uint32_t wgt[5] = { 16, 8, 4, 2, 1 };
    for (int y = 0; y < size; y++) for (int x = 0; x < size; x++) {
        uint32_t total = 0;
        for (int o = 0; o < 5; o++) {
            uint32_t period = (uint32_t)size >> (2 + o); if (period < 1) period = 1;
            total += wgt[o] * value_noise16((uint32_t)x, (uint32_t)y, period, seed + (uint32_t)o);
        }
        uint32_t v16 = total / 31u;
        int32_t s = ((int32_t)v16 - 9000) * 3 / 2;
        if (s < 0) s = 0;
        if (s > 65535) s = 65535;
        uint32_t h16 = ((uint32_t)s * (uint32_t)s) >> 16;
        out[(size_t)y * size + x] = (uint8_t)(h16 >> 8);
    }
}
void orc_synth_albedo(int size, uint32_t seed, const uint8_t* height, uint8_t* out)
{
    static const int32_t hs[6] = { 0, 20, 40, 110, 180, 255 };
    static const int32_t cs[6][3] = { {40,70,110}, {60,90,120}, {180,165,120}, {70,120,50}, {110,100,90}, {235,235,240} };
    for (int y = 0; y < size; y++) for (int x = 0; x < size; x++) {
        int32_t hgt = height[(size_t)y * size + x];
        int seg = 0; while (seg < 4 && hgt >= hs[seg + 1]) seg++;
        int32_t h0 = hs[seg], h1 = hs[seg + 1];
        uint32_t n = hash32((uint32_t)x, (uint32_t)y, seed) & 255u;
        uint8_t* o = out + 4 * ((size_t)y * size + x);
        for (int c = 0; c < 3; c++) {
            int32_t v = (cs[seg][c] * (h1 - hgt) + cs[seg + 1][c] * (hgt - h0)) / (h1 - h0);
            v += (int32_t)(n >> 4) - 8;
            o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
        o[3] = 255;
    }
}

/* ------------------------------------------------------------------------- */
/* ToneMappingPass (f3) [DONUT-RECOLLECTION]                                    */
/* Called as SimpleRender(cmd, ToneMappingParameters(), view, HdrColor)         */
/* (Renderer.cpp:430-431) on a pass made with default CreateParameters (:256).  */
/* fp32, fixed operation order, no FMA; log2 / exp2 are pinned polynomials so   */
/* that the integer results are the same on every implementation.               */
/* ------------------------------------------------------------------------- */
float orc_log2_pinned(float x)            /* x > 0, finite, normal; same cubic as the LOD computation */
{
    uint32_t bits; memcpy(&bits, &x, 4);
    int e = (int)((bits >> 23) & 255u) - 127;
    uint32_t mb = (bits & 0x7fffffu) | 0x3f800000u;
    float m; memcpy(&m, &mb, 4);
    float tt = m - 1.0f;
    float p = tt * (1.4208646f + tt * (-0.57725066f + tt * 0.1563861f));
    return (float)e + p;
}
float orc_exp2_pinned(float x)            /* cubic for 2^frac, exponent by bit construction; x clamped to [-126, 127] */
{
    if (!(x == x)) return 0.0f;
    if (x < -126.0f) x = -126.0f;
    if (x > 127.0f) x = 127.0f;
    float n = floorf(x), f = x - n;
    float p = 1.0f + f * (0.69583356f + f * (0.22606716f + f * 0.07809929f));
    uint32_t sb = (uint32_t)((int)n + 127) << 23;
    float s; memcpy(&s, &sb, 4);
    return p * s;
}
static inline float tm_luminance(float r, float g, float b) { return (r * 0.2126f + g * 0.7152f) + b * 0.0722f; }
static inline float tm_saturate(float v) { if (!(v > 0.0f)) return 0.0f; return v > 1.0f ? 1.0f : v; }

void orc_tonemap_histogram(const vr_tonemap_params* p, const uint16_t* hdr, int w, int h, const vr_partition* part,
                           uint32_t hist[VR_TONEMAP_BINS])
{
    const float scale = 1.0f / (p->max_log_luminance - p->min_log_luminance), bias = (0.0f - p->min_log_luminance) * scale;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        if (!owns_pixel(part, x, y)) continue;
        const uint16_t* px = hdr + ((size_t)y * w + x) * 4;
        float lum = tm_luminance(orc_half_to_float(px[0]), orc_half_to_float(px[1]), orc_half_to_float(px[2]));
        float t;                                        /* saturate(log2(lum) * scale + bias) */
        uint32_t lb; memcpy(&lb, &lum, 4);
        if (!(lum > 0.0f)) t = 0.0f;                    /* log2(0) = -inf, negatives and NaN -> 0 */
        else if ((lb >> 23) == 0u) t = 0.0f;            /* denormal: far below the range */
        else if ((lb >> 23) == 255u) t = 1.0f;          /* +inf */
        else t = tm_saturate(orc_log2_pinned(lum) * scale + bias);
        float hb = t * (float)(VR_TONEMAP_BINS - 1);
        float lf = floorf(hb);
        int left = (int)lf;
        uint32_t rw = (uint32_t)((hb - lf) * 64.0f), lw = 64u - rw;   /* 6-bit fixed-point weights */
        if (lw != 0u && left < VR_TONEMAP_BINS) hist[left] += lw;
        if (rw != 0u && left + 1 < VR_TONEMAP_BINS) hist[left + 1] += rw;
    }
}

/* Written so that 256 lanes can evaluate it with the identical result: the running totals are exact
 * integer prefix sums (rounded to float once), and the two 256-term float sums are pairwise
 * (stride-halving) reductions in a fixed order. */
float orc_tonemap_exposure(const vr_tonemap_params* p, const uint32_t hist[VR_TONEMAP_BINS], float frame_time, float old_lum)
{
    const float scale = 1.0f / (p->max_log_luminance - p->min_log_luminance), bias = (0.0f - p->min_log_luminance) * scale;
    uint64_t prefix[VR_TONEMAP_BINS + 1];
    prefix[0] = 0;
    for (int i = 0; i < VR_TONEMAP_BINS; i++) prefix[i + 1] = prefix[i] + hist[i];
    const float ftotal = (float)prefix[VR_TONEMAP_BINS];
    const float lo = ftotal * p->histogram_low_percentile, hi = ftotal * p->histogram_high_percentile;
    float acc[VR_TONEMAP_BINS], wgt[VR_TONEMAP_BINS];
    for (int i = 0; i < VR_TONEMAP_BINS; i++) {
        float below = (float)prefix[i], running = (float)prefix[i + 1];
        float ca = running < lo ? lo : (running > hi ? hi : running);
        float cb = below < lo ? lo : (below > hi ? hi : below);
        wgt[i] = ca - cb;                                /* part of this bin between the two percentiles */
        float log_lum = ((float)i / (float)(VR_TONEMAP_BINS - 1) - bias) / scale;
        acc[i] = log_lum * wgt[i];
    }
    for (int stride = VR_TONEMAP_BINS / 2; stride >= 1; stride >>= 1)
        for (int i = 0; i < stride; i++) { acc[i] = acc[i] + acc[i + stride]; wgt[i] = wgt[i] + wgt[i + stride]; }
    float accum = acc[0], wsum = wgt[0];
    float avg_log = wsum > 0.0f ? accum / wsum : p->min_log_luminance;
    float target = orc_exp2_pinned(avg_log);
    if (target < p->min_adapted_luminance) target = p->min_adapted_luminance;
    if (target > p->max_adapted_luminance) target = p->max_adapted_luminance;
    if (!(old_lum > 0.0f)) return target;                /* unset: jump */
    float diff = target - old_lum;
    float speed = diff > 0.0f ? p->eye_adaptation_speed_up : p->eye_adaptation_speed_down;
    if (!(speed > 0.0f)) return target;
    float k = (float)(1.0 - exp(-(double)frame_time * (double)speed));
    return old_lum + diff * k;
}

void orc_tonemap_apply(const vr_tonemap_params* p, float adapted, const uint16_t* hdr, int w, int h, uint8_t* ldr)
{
    const float exposure_scale = exp2f(p->exposure_bias);
    const float wp_inv2 = 1.0f / (p->white_point * p->white_point);
    if (!(adapted > 0.0f)) adapted = p->min_adapted_luminance;
    const float inv_adapted = 1.0f / adapted;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        float c[3] = { orc_half_to_float(hdr[i*4]), orc_half_to_float(hdr[i*4+1]), orc_half_to_float(hdr[i*4+2]) };
        float src = tm_luminance(c[0], c[1], c[2]);
        float k = 0.0f;
        if (src > 0.0f) {
            /* mapped = scaled (1 + scaled / white^2) / (1 + scaled);  k = mapped / src, as one division */
            float scaled = (exposure_scale * src) * inv_adapted;
            k = (scaled * (1.0f + scaled * wp_inv2)) / ((1.0f + scaled) * src);
        }
        for (int ch = 0; ch < 3; ch++) ldr[i*4+ch] = orc_linear_to_srgb8(src > 0.0f ? c[ch] * k : 0.0f);   /* SRGBA8 target: saturate + OETF */
        ldr[i*4+3] = 255;
    }
}

/* ------------------------------------------------------------------------- */
/* CPU baseline timings                                        */
/* ------------------------------------------------------------------------- */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

double orc_time_tree_build(const vr_terrain_params* p, const uint8_t* height_r8, int w, int h)
{
    /* QuadTree::Init: malloc+memcpy of the heightmap, root, Split (QuadTree.cpp:19-44) */
    double t0 = now_s();
    orc_terrain t; memset(&t, 0, sizeof(t));
    t.p = *p;
    uint8_t* copy = (uint8_t*)malloc((size_t)w * h);
    memcpy(copy, height_r8, (size_t)w * h);
    int l2 = p->surface_size >= 1.0f ? ilog2_floor_f(p->surface_size) : 0;
    t.num_lods = (VR_MAX_LODS - 1) < l2 ? (VR_MAX_LODS - 1) : l2;
    float ext[3] = { p->surface_size / 2.0f, 0.0f, p->surface_size / 2.0f };
    t.root = node_new(p->location, ext, 0);
    split(&t, t.root, 1, 0, 0, 0);
    double t1 = now_s();
    free_tree(t.root); free(copy);
    return t1 - t0;
}
double orc_time_select(orc_terrain* t, const vr_view* views, int nv, float max_height, int repeats, int* selected_total)
{
    int cap = t->p.max_instances;
    vr_instance* inst = (vr_instance*)malloc(sizeof(vr_instance) * cap);
    long total = 0;
    double t0 = now_s();
    for (int r = 0; r < repeats; r++) for (int i = 0; i < nv; i++) total += orc_select(t, &views[i], max_height, 0, NULL, inst, cap);
    double t1 = now_s();
    free(inst);
    if (selected_total) *selected_total = (int)(total / (repeats > 0 ? repeats : 1));
    return t1 - t0;
}
double orc_time_set_height(orc_terrain* t)
{
    double t0 = now_s();
    orc_set_height(t);
    double t1 = now_s();
    return t1 - t0;
}
