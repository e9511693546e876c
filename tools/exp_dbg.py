import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrenderer_amd as vr
from tests.common import params
from bench import flythrough_camera
W, H, size = 7680, 4320, 2048
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(0, 120, 12)]
rp = vr.default_render_params(400.0, assume_cleared=1)
for it in range(2):
    if it == 1: ctx.timing_enable(True)
    for v in views: tp.Render(v, v, rt, rp)
    ctx.synchronize()
t = ctx.timing_collect()
print(os.environ.get("VR_DBG", "0"), {k: round(ms / n * 1e3, 1) for k, (ms, n) in t.items() if k in ("k_raster",)}, flush=True)
