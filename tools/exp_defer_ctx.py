"""Scratch: k_deferred duration in different surroundings (after the tile pass, after an idle gap, back to back)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrenderer_amd as vr
from tests.common import AMBIENT_BOTTOM, AMBIENT_TOP, params
from bench import flythrough_camera
W, H, size = 7680, 4320, 2048
ctx = vr.Context(0); ctx.set_async_geometry(False)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H); hdr = vr.HdrImage(ctx, W, H)
dl = vr.DeferredLightingPass(ctx)
views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(120)]
rp = vr.default_render_params(400.0, assume_cleared=1)
sun = [vr.reference_sun()]
def report(name):
    t = ctx.timing_collect(); ctx.timing_enable(False)
    print(name, {k: round(ms / n * 1e3, 1) for k, (ms, n) in t.items() if k in ("k_raster", "k_deferred")}, flush=True)
def warm():
    for i in range(3):
        tp.Render(views[i], views[i], rt, rp); dl.Render(views[i], rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    ctx.synchronize()
warm(); ctx.timing_enable(True)
for i in range(20):
    tp.Render(views[i], views[i], rt, rp); dl.Render(views[i], rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
ctx.synchronize(); report("in frame (raster -> deferred)       ")
warm(); ctx.timing_enable(True)
for i in range(20):
    tp.Render(views[i], views[i], rt, rp); ctx.synchronize()
    dl.Render(views[i], rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr); ctx.synchronize()
report("host sync between the two passes    ")
warm(); ctx.timing_enable(True)
for i in range(20):
    tp.Render(views[i], views[i], rt, rp); ctx.synchronize(); time.sleep(0.002)
    dl.Render(views[i], rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr); ctx.synchronize()
report("2 ms idle between the two passes    ")
warm(); ctx.timing_enable(True)
for i in range(20):
    dl.Render(views[0], rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
ctx.synchronize(); report("deferred back to back               ")
warm(); ctx.timing_enable(True)
for i in range(20):
    dl.Render(views[0], rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr); ctx.synchronize(); time.sleep(0.002)
report("deferred alone, 2 ms idle in between")
