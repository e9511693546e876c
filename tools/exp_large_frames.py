"""Scratch: bin entries / status of very large targets on 32-pixel tiles (the scratch's bins must grow, never overflow for good)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vrenderer_amd as vr
from vrenderer_amd.scene import params, flythrough_camera
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, 2048, 1337); al = vr.synth_albedo(ctx, 2048, hm, 4242)
tp = vr.TerrainPass(ctx, params(2048)).Init(hm, al)
for (w, h) in ((15360, 8640), (16384, 16384)):
    rt = vr.RenderTargets(ctx).Init(w, h)
    for i in (0, 40, 40):
        v = vr.make_view(*flythrough_camera(i), w, h)
        try:
            tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1))
            print(w, h, i, tp.render_stats(), tp.memory_bytes(), flush=True)
        except Exception as e:
            print(w, h, i, "error:", e, flush=True)
    rt.close()
