"""Scratch: a fixed set of 8K flythrough frames through the tile pass only, for rocprofv3 --pmc passes.
VARIANT=<name> picks vrenderer_amd/lib/variants/<name>/libvrterrain.so (default: the product library)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vrenderer_amd import capi
if os.environ.get("VARIANT"):
    capi.LIB_PATH = os.path.join(ROOT, "vrenderer_amd", "lib", "variants", os.environ["VARIANT"], "libvrterrain.so")
import vrenderer_amd as vr
from vrenderer_amd.scene import params
from bench import flythrough_camera

W, H, size = int(os.environ.get("W", 7680)), int(os.environ.get("H", 4320)), 2048
ctx = vr.Context(0); ctx.set_async_geometry(False)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
rp = vr.default_render_params(400.0, assume_cleared=1)
ctx.timing_enable(True)
for i in (0, 30, 60, 90, 15, 45, 75, 105):
    v = vr.make_view(*flythrough_camera(i), W, H)
    tp.Render(v, v, rt, rp)
ctx.synchronize()
t = ctx.timing_collect()
print({k: round(ms / n * 1e3, 1) for k, (ms, n) in t.items()})
