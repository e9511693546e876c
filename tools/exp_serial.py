"""Scratch: tile pass + lighting per frame WITHOUT asynchronous geometry (the chain runs in front of the tile pass on the
same stream): separates what the lighting pass's cache traffic does to the next tile pass from what the overlapped
geometry does.  VARIANT=<name> picks a library variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vrenderer_amd import capi
if os.environ.get("VARIANT"):
    capi.LIB_PATH = os.path.join(ROOT, "vrenderer_amd", "lib", "variants", os.environ["VARIANT"], "libvrterrain.so")
import vrenderer_amd as vr
from vrenderer_amd.scene import params, AMBIENT_TOP, AMBIENT_BOTTOM, flythrough_camera
W, H, size = 7680, 4320, 2048
ctx = vr.Context(0); ctx.set_async_geometry(False)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H); hdr = vr.HdrImage(ctx, W, H)
rp = vr.default_render_params(400.0, assume_cleared=1)
dl = vr.DeferredLightingPass(ctx); sun = [vr.reference_sun()]
views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(0, 120, 5)]
for light in (True, False):
    for it in range(2):
        if it == 1: ctx.timing_enable(True)
        for v in views:
            tp.Render(v, v, rt, rp)
            if light: dl.Render(v, rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        ctx.synchronize()
    t = ctx.timing_collect(); ctx.timing_enable(False)
    print("with lighting   " if light else "tile passes only", {k: round(ms / n * 1e3, 1) for k, (ms, n) in t.items()}, flush=True)
