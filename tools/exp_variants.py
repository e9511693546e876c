"""Scratch: k_raster time of library variants (timing only; experimental variants may render garbage)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2:
    for v in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, v])
    sys.exit(0)
sys.path.insert(0, ROOT)
from vrenderer_amd import capi
v = sys.argv[1]
if v != "default":
    capi.LIB_PATH = os.path.join(ROOT, "vrenderer_amd", "lib", "variants", v, "libvrterrain.so")
import vrenderer_amd as vr
from tests.common import params
from bench import flythrough_camera
W, H, size = int(os.environ.get("W", 7680)), int(os.environ.get("H", 4320)), 2048
ctx = vr.Context(0); ctx.set_async_geometry(False)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
out = []
for mode in (dict(assume_cleared=1, depth_only=1), dict(assume_cleared=1)):
    rp = vr.default_render_params(400.0, **mode)
    views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(0, 120, 10)]
    for it in range(2):
        if it == 1: ctx.timing_enable(True)
        for vw in views: tp.Render(vw, vw, rt, rp)
        ctx.synchronize()
    t = ctx.timing_collect(); ctx.timing_enable(False)
    out.append(round(t["k_raster"][0] / t["k_raster"][1] * 1e3, 1))
print(f"{v:12s} depth-only {out[0]} us   full {out[1]} us", flush=True)
