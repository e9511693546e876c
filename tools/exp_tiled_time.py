"""Scratch (GPU box): time of the tiled lighting pass alone at 8K over 8 flythrough frames, per kernel (rocprofv3 not
needed: context kernel timing).  VARIANT=<name> picks vrenderer_amd/lib/variants/<name>/libvrterrain.so.  LIGHTS=n."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vrenderer_amd import capi
if os.environ.get("VARIANT"):
    capi.LIB_PATH = os.path.join(ROOT, "vrenderer_amd", "lib", "variants", os.environ["VARIANT"], "libvrterrain.so")
import torch
import vrenderer_amd as vr
from vrenderer_amd.scene import params, AMBIENT_TOP, AMBIENT_BOTTOM, flythrough_camera
size = 2048
w, h = int(os.environ.get("W", 7680)), int(os.environ.get("H", 4320))
nl = int(os.environ.get("LIGHTS", 1024))
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(w, h)
hdr = vr.HdrImage(ctx, w, h)
lights = [vr.reference_sun()] + vr.synthetic_point_lights(nl - 1, 2048.0, hm, 400.0, seed=9001)
tl = vr.TiledDeferredLightingPass(ctx)
res = []
for f in range(0, 120, 15):
    v = vr.make_view(*flythrough_camera(f), w, h)
    rt.Clear(); tp.Render(v, v, rt, vr.default_render_params(400.0))
    for _ in range(2): tl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    ctx.synchronize()
    ctx.timing_enable(True)
    for _ in range(5): tl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    ctx.synchronize()
    t = ctx.timing_collect(); ctx.timing_enable(False)
    res.append(sum(ms for ms, n in t.values()) / 5 * 1e3)
print("tiled pass us per frame:", " ".join("%.0f" % r for r in res), "mean %.1f" % (sum(res) / len(res)), "status", tl.Status())
