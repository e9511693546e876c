"""Scratch: per-kernel times of the geometry chain when nothing overlaps it (single stream), at several resolutions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vrenderer_amd import capi
if os.environ.get("VARIANT"):
    capi.LIB_PATH = os.path.join(ROOT, "vrenderer_amd", "lib", "variants", os.environ["VARIANT"], "libvrterrain.so")
    capi.EXPORTS = [e for e in capi.EXPORTS if e not in ("vr_debug_fastmath_check", "vr_deferred_tiled_status", "vr_frame_allgather", "vr_frame_allgather_ldr",
                                                           "vr_tonemap_allreduce_histogram", "vr_ldr_image_create", "vr_ldr_image_destroy", "vr_ldr_image_device_ptr",
                                                           "vr_ldr_image_capacity", "vr_ldr_image_download")] if os.environ["VARIANT"] == "r1" else capi.EXPORTS
import vrenderer_amd as vr
from vrenderer_amd.scene import params, flythrough_camera
size = 2048
ctx = vr.Context(0); ctx.set_async_geometry(False)
if os.environ.get("TILE"): ctx.set_raster_tile(int(os.environ["TILE"]))      # pin the raster tile edge (32 / 64)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
for (W, H) in ((1920, 1080), (3840, 2160), (7680, 4320)):
    rt = vr.RenderTargets(ctx).Init(W, H)
    rp = vr.default_render_params(400.0, assume_cleared=1)
    for i in range(3):
        v = vr.make_view(*flythrough_camera(i), W, H); tp.Render(v, v, rt, rp)
    ctx.synchronize(); ctx.timing_enable(True)
    for i in range(0, 120, 6):
        v = vr.make_view(*flythrough_camera(i), W, H); tp.Render(v, v, rt, rp)
    ctx.synchronize()
    t = ctx.timing_collect(); ctx.timing_enable(False)
    k = {n: round(ms / c * 1e3, 1) for n, (ms, c) in t.items()}
    print(W, H, k, "chain", round(sum(v for n, v in k.items() if n != "k_raster"), 1), flush=True)
    rt.close()
