"""Scratch experiment: per-kernel timings of the 8K frame in different modes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrenderer_amd as vr
from tests.common import AMBIENT_BOTTOM, AMBIENT_TOP, DEFAULT_EYE, DEFAULT_TARGET, params
from bench import flythrough_camera

W, H, size = int(os.environ.get("W", 7680)), int(os.environ.get("H", 4320)), 2048
ctx = vr.Context(0)
ctx.set_async_geometry(False)   # clean per-kernel numbers
ctx.set_async_geometry(False)   # clean per-kernel numbers
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
hdr = vr.HdrImage(ctx, W, H)
dl = vr.DeferredLightingPass(ctx)
views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(120)] + [vr.make_view(DEFAULT_EYE, DEFAULT_TARGET, W, H)]
def run(name, rp, frames, deferred=True, clear=False):
    for it in range(2):
        if it == 1: ctx.timing_enable(True)
        for i in frames:
            if clear: rt.Clear()
            tp.Render(views[i], views[i], rt, rp)
            if deferred: dl.Render(views[i], rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        ctx.synchronize()
    t = ctx.timing_collect(); ctx.timing_enable(False)
    print(name, {k: round(ms / n * 1e3, 1) for k, (ms, n) in t.items()}, tp.render_stats(), flush=True)
fr = list(range(0, 120, 12))
run("fused-clear      ", vr.default_render_params(400.0, assume_cleared=1), fr)
run("clear+render     ", vr.default_render_params(400.0), fr, clear=True)
run("depth-only       ", vr.default_render_params(400.0, assume_cleared=1, depth_only=1), fr, deferred=False)
run("default cam      ", vr.default_render_params(400.0, assume_cleared=1), [120] * 5)
for i in (0, 30, 60, 90):
    run(f"fly frame {i:3d}    ", vr.default_render_params(400.0, assume_cleared=1), [i] * 3)
