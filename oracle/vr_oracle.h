/*
 * vr_oracle.h — CPU oracle for the vrenderer terrain + deferred-shading hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load
 * or call it, and there only as the checker.  The product (libvrterrain.so) never
 * links or calls this code.
 *
 * PARITY UNPINNED.  The reference (Viictor/vrenderer) has no tests, golden vectors
 * or fixtures, its HLSL cannot be compiled or run here, and everything it gets from
 * the NVIDIA Donut submodule (math library, frustum, texture sampling, G-buffer
 * formats, DeferredLightingPass) is absent from the checkout (empty submodule, SHA
 * unknown).  QuadTree.cpp includes Donut headers, so under this round's rules (no
 * stand-in headers) it is unbuildable here.  This oracle is therefore a restatement
 * pinned only by (i) the reference source text it cites line by line and (ii) the
 * node counts the survey recorded from the reference's own QuadTree.cpp
 * (87,381 / 5,592,405 nodes; 28 / 562 selected at the default
 * camera with the frustum test stubbed to "always intersects").
 */
#ifndef VR_ORACLE_H
#define VR_ORACLE_H

#include "../include/vrterrain.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_terrain orc_terrain;

/* textures + quadtree (TerrainPass::Init, QuadTree::Init/Split) */
orc_terrain* orc_terrain_create(const vr_terrain_params* p,
                                const uint8_t* height_r8, int hm_w, int hm_h,
                                const uint8_t* albedo_srgba8, int al_w, int al_h);
void   orc_terrain_destroy(orc_terrain* t);
int    orc_terrain_num_lods(const orc_terrain* t);
void   orc_terrain_lod_ranges(const orc_terrain* t, float out[VR_MAX_LODS]);
long   orc_terrain_num_nodes(const orc_terrain* t);
int    orc_terrain_height_levels(const orc_terrain* t);
int    orc_terrain_albedo_levels(const orc_terrain* t);
const uint8_t* orc_terrain_height_mip(const orc_terrain* t, int level, int* w, int* h);
const uint8_t* orc_terrain_albedo_mip(const orc_terrain* t, int level, int* w, int* h);

/* QuadTree::NodeSelect + TerrainPass::UpdateTransforms.  stub_frustum != 0 makes
 * frustum.intersectsWith() always true (the survey's probe configuration). */
int    orc_select(orc_terrain* t, const vr_view* v, float max_height, int stub_frustum,
                  uint32_t* node_ids, vr_instance* inst, int capacity);
/* QuadTree::SetHeight (QuadTree.cpp:191-208; disabled in the reference at :46-51).
 * Returns per-node (position.y, extents.y) for node ids < max_ids, for checks. */
void   orc_set_height(orc_terrain* t);
int    orc_node_height(const orc_terrain* t, uint32_t node_id, float* pos_y, float* ext_y);
void   orc_set_height_loaded(orc_terrain* t, int loaded);            /* m_HeightLoaded */
void   orc_reset_height(orc_terrain* t);                             /* tree as built, m_HeightLoaded = false */
void   orc_node_heights(const orc_terrain* t, float* out, long max_ids);   /* (pos.y, ext.y) per node id */

/* FirstPersonCamera::LookAt + perspProjD3DStyle + PlanarView::UpdateCache. */
void   orc_view_from_camera(const float eye[3], const float target[3], const float up[3],
                            float vfov, float z_near, float z_far, int w, int h, vr_view* out);

/* CascadedShadowMap::SetupForPlanarViewStable, one cascade (Renderer.cpp:345-352) [DONUT-RECOLLECTION]. */
void   orc_shadow_view(const vr_light* light, const vr_view* camera_view, const vr_shadow_params* p, vr_view* out);
/* Shadow binding used by the following orc_deferred / orc_deferred_f32 calls (NULL view or depth = none):
 * the light `light_index` is multiplied by the 4x4 tent-PCF lookup into the res^2 depth map. */
void   orc_deferred_set_shadow(const vr_view* light_view, const float* shadow_depth, int res, int light_index,
                               float depth_bias);

/* main_vs for one vertex of one instance (terrain_vs.hlsl:35-62). */
void   orc_vertex(const orc_terrain* t, const vr_view* v, float max_height,
                  const vr_instance* inst, int vx, int vz, float clip[4], float world[3]);

/* TerrainPass::Render (select + draw) into host G-buffer planes, which must hold
 * valid contents (e.g. cleared by orc_gbuffer_clear).  If part != NULL only pixels
 * of owned tiles are touched.  Returns the number of selected nodes. */
void   orc_gbuffer_clear(int w, int h, float* depth, uint32_t* diffuse, uint32_t* specular,
                         uint16_t* normals, uint16_t* emissive);
int    orc_render(orc_terrain* t, const vr_view* v, const vr_render_params* rp,
                  const vr_partition* part, int w, int h,
                  float* depth, uint32_t* diffuse, uint32_t* specular,
                  uint16_t* normals, uint16_t* emissive);

/* DeferredLightingPass::Render -> RGBA16F (4 halfs / pixel, row-major). */
void   orc_deferred(const vr_view* v, int w, int h,
                    const float* depth, const uint32_t* diffuse, const uint32_t* specular,
                    const uint16_t* normals, const uint16_t* emissive,
                    const vr_light* lights, int num_lights,
                    const float amb_top[3], const float amb_bottom[3],
                    uint16_t* hdr_out);
/* same, fp32 output before the half conversion (for RMS reporting) */
void   orc_deferred_f32(const vr_view* v, int w, int h,
                    const float* depth, const uint32_t* diffuse, const uint32_t* specular,
                    const uint16_t* normals, const uint16_t* emissive,
                    const vr_light* lights, int num_lights,
                    const float amb_top[3], const float amb_bottom[3],
                    float* rgba_out);

/* ToneMappingPass (Renderer.cpp:256-257,430-431) [DONUT-RECOLLECTION].
 * histogram: adds this frame's (owned) pixels to hist; exposure: returns the new adapted luminance;
 * apply: RGBA16F -> SRGBA8 (alpha 255). */
void   orc_tonemap_histogram(const vr_tonemap_params* p, const uint16_t* hdr_rgba16f, int w, int h,
                             const vr_partition* part, uint32_t hist[VR_TONEMAP_BINS]);
float  orc_tonemap_exposure(const vr_tonemap_params* p, const uint32_t hist[VR_TONEMAP_BINS],
                            float frame_time_seconds, float old_adapted_luminance);
void   orc_tonemap_apply(const vr_tonemap_params* p, float adapted_luminance, const uint16_t* hdr_rgba16f,
                         int w, int h, uint8_t* ldr_srgba8);
float  orc_log2_pinned(float x);
float  orc_exp2_pinned(float x);

/* synthetic inputs */
void   orc_synth_heightmap(int size, uint32_t seed, uint8_t* out_r8);
void   orc_synth_albedo(int size, uint32_t seed, const uint8_t* height_r8, uint8_t* out_srgba8);

/* Revision of the raster / sampler model (vr_oracle.c header), frozen from round 4 on; and a debug tap for the test that
 * bounds the model against a float64 evaluation: plane = w*h*3 floats (implicit LOD before the sampler's clamp, world x,
 * world z of every shaded pixel of the following orc_render calls), NULL = off.  Not part of the model. */
int    orc_model_revision(void);
void   orc_debug_set_pixel_plane(float* plane);

/* small helpers exposed for tests */
float    orc_half_to_float(uint16_t h);
uint16_t orc_float_to_half(float f);
float    orc_srgb8_to_linear(uint8_t c);
uint8_t  orc_linear_to_srgb8(float x);
void     orc_linear_to_srgb8_batch(const float* in, size_t n, uint8_t* out);

/* CPU baseline legs (bench.py cpu_baseline, kind "port"): the reference's CPU-side
 * terrain work, single-threaded like the reference's main thread. Return seconds. */
double orc_time_tree_build(const vr_terrain_params* p, const uint8_t* height_r8, int w, int h);
double orc_time_select(orc_terrain* t, const vr_view* views, int num_views, float max_height,
                       int repeats, int* selected_total);
double orc_time_set_height(orc_terrain* t);

#ifdef __cplusplus
}
#endif
#endif
