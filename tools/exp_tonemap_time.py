"""Scratch (GPU box): the tone-map stage alone on a lit 8K frame (whole frame and one rank's packed tiles of an N-way split)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vrenderer_amd import capi
if os.environ.get("VARIANT"):
    capi.LIB_PATH = os.path.join(ROOT, "vrenderer_amd", "lib", "variants", os.environ["VARIANT"], "libvrterrain.so")
import vrenderer_amd as vr
from vrenderer_amd.passes import partition_info
from vrenderer_amd.scene import params, AMBIENT_TOP, AMBIENT_BOTTOM, flythrough_camera
size, w, h = 2048, 7680, 4320
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(w, h)
v = vr.make_view(*flythrough_camera(30), w, h)
tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1))
dl = vr.DeferredLightingPass(ctx)
tm = vr.ToneMappingPass(ctx); tmp = vr.default_tonemap_params(); tm.AdvanceFrame(1.0 / 60.0)
for world in (1, 8):
    part = None if world == 1 else vr.Partition(0, world)
    if part is None:
        hdr = vr.HdrImage(ctx, w, h); ldr = vr.LdrImage(ctx, w, h)
    else:
        info = partition_info(w, h, 0, world)
        rows = (info["packed_bytes"] + vr.VR_OWNER_TILE * 8 - 1) // (vr.VR_OWNER_TILE * 8)
        hdr = vr.HdrImage(ctx, vr.VR_OWNER_TILE, rows); ldr = vr.LdrImage(ctx, w, h)
    dl.Render(v, rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr, part)
    for it in range(3):
        if it == 1: ctx.timing_enable(True)
        for _ in range(10):
            tm.ResetHistogram(); tm.AddFrameToHistogram(tmp, hdr, w, h, part); tm.ComputeExposure(tmp); tm.Render(tmp, hdr, ldr, w, h, part)
        ctx.synchronize()
    t = ctx.timing_collect(); ctx.timing_enable(False)
    print("world", world, {k: round(ms / n * 1e3, 1) for k, (ms, n) in t.items()})
