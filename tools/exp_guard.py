"""Scratch: find a camera whose triangles cross the guard band (clipped_tris > 0 without near-plane crossing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vrenderer_amd as vr
from tests.common import params
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, 256); al = vr.synth_albedo(ctx, 256, hm)
tp = vr.TerrainPass(ctx, params(256)).Init(hm, al)
w, h = 512, 288
rt = vr.RenderTargets(ctx).Init(w, h)
hgt = float(hm[128 - 2, 128 + 1]) / 255.0 * 400.0
for eye, tgt in (((1.3, hgt + 0.5, -2.2), (1.32, hgt - 5.0, -2.18)), ((1.3, hgt + 3.0, -2.2), (2.3, hgt - 5.0, -1.2))):
    for fov in (3.0, 1.0, 0.3, 0.1, 0.03, 0.01):
        v = vr.make_view(eye, tgt, w, h, vfov_deg=fov)
        tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1))
        d = rt.download("depth")
        print(eye[1] - hgt, fov, tp.render_stats(), float((d < 1).mean()), flush=True)
