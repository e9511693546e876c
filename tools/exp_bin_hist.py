"""Scratch: distribution of bin lengths (entries per raster tile) of flythrough frames at W x H."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vrenderer_amd as vr
from vrenderer_amd.scene import params
from bench import flythrough_camera
W, H, size = int(os.environ.get("W", 7680)), int(os.environ.get("H", 4320)), 2048
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
rp = vr.default_render_params(400.0, assume_cleared=1)
for i in (0, 30, 60, 90):
    v = vr.make_view(*flythrough_camera(i), W, H)
    tp.Render(v, v, rt, rp); ctx.synchronize()
    tiles, lens = tp.tile_order()
    edges = [0, 1, 5, 9, 17, 33, 65, 129, 257, 1 << 30]
    h = np.histogram(lens, edges)[0]
    print(f"frame {i}: {len(lens)} tiles, {int(lens.sum())} entries, max {int(lens.max())}; tiles by bin length "
          + ", ".join(f"{a}..{b - 1}: {c} ({100.0 * c / len(lens):.1f}%)" for a, b, c in zip(edges[:-1], edges[1:], h)), flush=True)
