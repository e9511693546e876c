"""Scratch: config 5 timing — 8K terrain + 1024 point lights, tiled deferred."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrenderer_amd as vr
from tests.common import AMBIENT_BOTTOM, AMBIENT_TOP, params
from bench import flythrough_camera
W, H, size = 7680, 4320, 2048
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
hdr = vr.HdrImage(ctx, W, H)
views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(0, 120, 24)]
rp = vr.default_render_params(400.0, assume_cleared=1)
for n in (1, 16, 256, 1024, 4096):
    lights = [vr.reference_sun()] + vr.synthetic_point_lights(n - 1, 2048.0, hm)
    dl = vr.TiledDeferredLightingPass(ctx)
    for it in range(2):
        if it == 1: ctx.timing_enable(True)
        for v in views:
            tp.Render(v, v, rt, rp)
            dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        ctx.synchronize()
    t = ctx.timing_collect(); ctx.timing_enable(False)
    print(n, 'lights', {k: round(ms / c * 1e3, 1) for k, (ms, c) in t.items() if 'deferred' in k}, flush=True)
