"""Scratch: tiled deferred pass timing (config 5) for library variants."""
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
from tests.common import AMBIENT_BOTTOM, AMBIENT_TOP, params
from bench import flythrough_camera
W, H, size = 7680, 4320, 2048
ctx = vr.Context(0); ctx.set_async_geometry(False)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H); hdr = vr.HdrImage(ctx, W, H)
view = vr.make_view(*flythrough_camera(30), W, H)
tp.Render(view, view, rt, vr.default_render_params(400.0, assume_cleared=1))
td = vr.TiledDeferredLightingPass(ctx)
res = []
for n in (1, 256, 1024, 4096):
    lights = [vr.reference_sun()] + vr.synthetic_point_lights(n - 1, float(size), hm)
    for it in range(2):
        if it == 1: ctx.timing_enable(True)
        for _ in range(5): td.Render(view, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        ctx.synchronize()
    t = ctx.timing_collect(); ctx.timing_enable(False)
    res.append((n, round(t["k_deferred_tiled"][0] / t["k_deferred_tiled"][1] * 1e3, 1)))
print(f"{v:10s}", res, flush=True)
