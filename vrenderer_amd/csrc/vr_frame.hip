// vr_frame_submit: one frame of the hot path as ONE call - the terrain part of Renderer::RecordCommand (Renderer.cpp:321-446:
// one command list per frame): Clear (fused, assume_cleared) -> TerrainPass::Render (:401-415) -> DeferredLightingPass::Render
// (:417-428) -> ToneMappingPass::SimpleRender (:430-431) [-> the N-rank exchange], plus the geometry of up to two upcoming
// frames built ahead (vr_terrain_prepare).  It queues exactly what the per-call entry points queue, in the same order on the
// same streams; what it removes is the host's per-call cost - a frame is about nine calls otherwise, and a rank of an 8-way
// split of the 8K frame has a period near 0.1 ms (the host needed 98 us per frame through the Python wrappers).
// The cross-stream order between the lighting pass (terrain's context) and a tone-map stage on another context's stream is
// the library's here: the stage waits for the lighting pass's stop event, and an image remembers who still reads it, so a
// later lighting pass into the same vr_image waits for that reader (a host that rotates two images keeps both streams busy).
#include "vr_internal.h"

extern "C" VR_API int vr_frame_submit(vr_terrain* t, vr_gbuffer* gb, const vr_frame_desc* f)
{
    VR_REQUIRE(t && gb && f && f->view && f->render && f->hdr_out, "NULL argument");
    VR_REQUIRE(f->num_lights >= 0 && (f->num_lights == 0 || f->lights), "lights is NULL");
    vr_context* ctx = t->ctx;
    VR_REQUIRE(gb->ctx == ctx && f->hdr_out->ctx->device == ctx->device, "G-buffer / image belong to another context");
    vr_image* hdr = f->hdr_out;
    int rc;
    // TerrainPass::Render, then the chains of the next frames under its tile pass (a frame already prepared is a no-op)
    // (a sticky code - VR_ERR_TOO_MANY_INSTANCES / VR_ERR_OVERFLOW of an EARLIER frame - does not stop this one: it is returned at the end)
    int earlier = vr_terrain_render(t, f->view, f->view, gb, f->render, f->part);
    if (earlier != VR_OK && earlier != VR_ERR_TOO_MANY_INSTANCES && earlier != VR_ERR_OVERFLOW) return earlier;
    for (int k = 0; k < 2; k++)
        if (f->prepare_views[k] && (rc = vr_terrain_prepare(t, f->prepare_views[k], gb, f->render, f->part))) return rc;
    // whoever still reads the image this lighting pass overwrites (the tone-map stage of two frames ago, on another stream)
    if (hdr->read_pending) { VR_HIP(hipStreamWaitEvent(ctx->stream, hdr->ev_read_done, 0)); hdr->read_pending = false; }
    if (f->tiled) rc = vr_deferred_light_tiled(ctx, f->view, gb, f->lights, f->num_lights, f->ambient_top, f->ambient_bottom, hdr, f->part);
    else if (f->shadow) rc = vr_deferred_light_shadowed(ctx, f->view, gb, f->lights, f->num_lights, f->ambient_top, f->ambient_bottom, hdr, f->part, f->shadow);
    else rc = vr_deferred_light(ctx, f->view, gb, f->lights, f->num_lights, f->ambient_top, f->ambient_bottom, hdr, f->part);
    if (rc) return rc;
    if (!f->tonemap) return earlier;

    // ToneMappingPass::SimpleRender on the tone mapper's own context (another stream: it runs under the next frame's rendering)
    VR_REQUIRE(f->tonemap_params && f->ldr_out, "tone-map stage: params / ldr_out missing");
    vr_context* tc = vr_tonemap_context(f->tonemap);
    VR_REQUIRE(tc->device == ctx->device, "tone mapper lives on another device");
    const bool cross = tc->stream != ctx->stream;
    if (cross) {
        // behind the lighting pass: its dispatch-stamped stop event when there is one, else an explicit record
        hipEvent_t done = (ctx->dispatch_events && ctx->last_stop) ? ctx->last_stop : nullptr;
        if (!done) {
            if (!hdr->ev_written) VR_HIP(hipEventCreateWithFlags(&hdr->ev_written, hipEventDisableTiming));
            VR_HIP(hipEventRecord(hdr->ev_written, ctx->stream));
            done = hdr->ev_written;
        }
        VR_HIP(hipStreamWaitEvent(tc->stream, done, 0));
    }
    const int w = gb->w, h = gb->h;
    if ((rc = vr_tonemap_reset_histogram(f->tonemap))) return rc;
    if ((rc = vr_tonemap_add_frame_to_histogram(f->tonemap, f->tonemap_params, hdr, w, h, f->part))) return rc;
    if (f->nccl_comm && (rc = vr_tonemap_allreduce_histogram(f->tonemap, f->nccl_comm))) return rc;
    if ((rc = vr_tonemap_compute_exposure(f->tonemap, f->tonemap_params, f->frame_time_seconds))) return rc;
    if ((rc = vr_tonemap_render(f->tonemap, f->tonemap_params, hdr, w, h, f->ldr_out, f->ldr_capacity, f->part))) return rc;
    if (cross) {
        if (!hdr->ev_read_done) VR_HIP(hipEventCreateWithFlags(&hdr->ev_read_done, hipEventDisableTiming));
        VR_HIP(hipEventRecord(hdr->ev_read_done, tc->stream));
        hdr->read_pending = true;
    }
    if (f->nccl_comm) {
        VR_REQUIRE(f->part && f->gathered && f->ldr_frame, "exchange: partition / gathered / ldr_frame missing");
        if ((rc = vr_frame_allgather_ldr(tc, f->nccl_comm, f->ldr_out, f->gathered, f->part->world_size, w, h, f->ldr_frame))) return rc;
    }
    return earlier;
}
