"""Scratch: async geometry on/off, interleaved rounds in one process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrenderer_amd as vr
from tests.common import AMBIENT_BOTTOM, AMBIENT_TOP, params
from bench import flythrough_camera
W, H, size = 7680, 4320, 2048
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H); hdr = vr.HdrImage(ctx, W, H)
dl = vr.DeferredLightingPass(ctx)
views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(120)]
rp = vr.default_render_params(400.0, assume_cleared=1)
res = {0: [], 1: []}
for rnd in range(5):
    for mode in (0, 1):
        ctx.set_async_geometry(mode)
        for i in range(5):
            tp.Render(views[i], views[i], rt, rp); dl.Render(views[i], rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        ctx.synchronize(); ctx.timing_enable(True)
        t0 = time.perf_counter()
        for i in range(40):
            v = views[(5 + i) % 120]
            tp.Render(v, v, rt, rp); dl.Render(v, rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        ctx.synchronize(); dt = (time.perf_counter() - t0) / 40
        t = ctx.timing_collect(); ctx.timing_enable(False)
        res[mode].append((round(dt * 1e3, 4), round(t["k_deferred"][0] / t["k_deferred"][1] * 1e3, 1), round(t["k_raster"][0] / t["k_raster"][1] * 1e3, 1)))
for m in (0, 1): print("async" if m else "sync ", res[m])
