"""Scratch: fixed per-tile overhead of k_raster (camera looking at the sky: empty bins) vs the flythrough."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrenderer_amd as vr
from tests.common import params
from bench import flythrough_camera

W, H, size = 7680, 4320, 2048
ctx = vr.Context(0); ctx.set_async_geometry(False)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
sky = vr.make_view((0, 250, 0), (10, 2000, 0), W, H)
fly = vr.make_view(*flythrough_camera(30), W, H)
def run(name, view, **kw):
    rp = vr.default_render_params(400.0, **kw)
    for it in range(2):
        if it == 1: ctx.timing_enable(True)
        for _ in range(5): tp.Render(view, view, rt, rp)
        ctx.synchronize()
    t = ctx.timing_collect(); ctx.timing_enable(False)
    print(name, {k: round(ms / n * 1e3, 1) for k, (ms, n) in t.items() if k in ("k_raster", "k_select")}, tp.render_stats()["bin_entries"], flush=True)
run("sky  cleared depth-only", sky, assume_cleared=1, depth_only=1)
run("sky  cleared full      ", sky, assume_cleared=1)
run("sky  keep    full      ", sky)
run("fly  cleared depth-only", fly, assume_cleared=1, depth_only=1)
run("fly  cleared full      ", fly, assume_cleared=1)
run("fly  keep    full      ", fly)
