"""Scratch: deferred kernel timing only (G-buffer rendered once), interleaved rounds."""
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
v = vr.make_view(*flythrough_camera(0), W, H)
tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1))
dl = vr.DeferredLightingPass(ctx)
res = {}
for rnd in range(6):
    for mode in ("base", "nt"):
        if mode == "nt": os.environ["VR_NT"] = "1"
        else: os.environ.pop("VR_NT", None)
        for it in range(3): dl.Render(v, rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        ctx.timing_enable(True)
        for it in range(10): dl.Render(v, rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        t = ctx.timing_collect(); ctx.timing_enable(False)
        res.setdefault(mode, []).append(round(t["k_deferred"][0] / t["k_deferred"][1] * 1e3, 1))
print(res)
