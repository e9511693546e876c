"""debug: scratch growth, case 1 of test_scratch_grows_by_high_water_mark_and_reports_sticky_conditions"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vrenderer_amd as vr
from oracle import pyoracle as oracle
from tests.common import CAMERAS, params
oracle.build(); oracle.lib()
size = 2048
h = oracle.synth_heightmap(size); a = oracle.synth_albedo(size, h)
ctx = vr.Context(0)
w, hh = 960, 540
v = vr.make_view(*CAMERAS[0], w, hh)
rp = vr.default_render_params(400.0, assume_cleared=1)
ref = vr.TerrainPass(ctx, params(size)).Init(h, a)
rt0 = vr.RenderTargets(ctx).Init(w, hh)
ref.Render(v, v, rt0, rp); ctx.synchronize()
want = rt0.download("depth").view(np.uint32)
print("ref chunks", ref.num_chunks())
os.environ["VR_SCRATCH_INITIAL_NODES"] = sys.argv[1] if len(sys.argv) > 1 else "64"
tp = vr.TerrainPass(ctx, params(size)).Init(h, a)
rt = vr.RenderTargets(ctx).Init(w, hh)
for k in range(4):
    rc = ctx.lib.vr_terrain_render(tp.handle, C.byref(v), C.byref(v), rt.handle, C.byref(rp), None)
    ctx.synchronize()
    d = rt.download("depth").view(np.uint32)
    mis = np.argwhere(d != want)
    box = (mis.min(0).tolist(), mis.max(0).tolist()) if len(mis) else None
    print(f"frame {k}: rc={rc} err={ctx.lib.vr_last_error() if rc else b''} mismatches={len(mis)} box={box} scratch={tp.memory_bytes()['scratch']} zero={int((d == d[0,0]).sum())}")
