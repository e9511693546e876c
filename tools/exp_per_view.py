"""Scratch: tile pass / lighting pass time per view of the flythrough (each view rendered REP times back to back), to tell view
dependence from time dependence: python3 tools/exp_per_view.py [first last step]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vrenderer_amd as vr
from vrenderer_amd.scene import params, AMBIENT_TOP, AMBIENT_BOTTOM, flythrough_camera
W, H, size = int(os.environ.get("W", 7680)), int(os.environ.get("H", 4320)), 2048
first, last, step = (int(a) for a in (sys.argv[1:4] + ["0", "120", "5"])[:3])
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
hdr = vr.HdrImage(ctx, W, H)
dl = vr.DeferredLightingPass(ctx)
sun = [vr.reference_sun()]
rp = vr.default_render_params(400.0, assume_cleared=1)
for lap in range(2):
    out = []
    for i in range(first, last, step):
        v = vr.make_view(*flythrough_camera(i), W, H)
        for _ in range(3):
            tp.Render(v, v, rt, rp); dl.Render(v, rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        ctx.synchronize(); ctx.timing_enable(2)
        for _ in range(6):
            tp.Render(v, v, rt, rp); dl.Render(v, rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        ctx.synchronize(); t = ctx.timing_collect(); ctx.timing_enable(False)
        out.append((i, round(t["k_raster"][0] / t["k_raster"][1] * 1e3), round(t["k_deferred"][0] / t["k_deferred"][1] * 1e3)))
    print("lap", lap, " ".join("%d:%d/%d" % o for o in out), flush=True)
