"""Scratch: host time per API call of the N-rank frame loop (rank 0 of an emulated 8-way split, one GPU): where do the
~160 us per frame go that the host needs to queue a frame?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import vrenderer_amd as vr
from vrenderer_amd.scene import params, AMBIENT_TOP, AMBIENT_BOTTOM, flythrough_camera
W, H, size = 7680, 4320, 2048
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
rp = vr.default_render_params(400.0, assume_cleared=1)
part = vr.Partition(0, 8)
from vrenderer_amd.passes import partition_info
info = partition_info(W, H, 0, 8)
rows = (info["packed_bytes"] + vr.VR_OWNER_TILE * 8 - 1) // (vr.VR_OWNER_TILE * 8)
hdr = vr.HdrImage(ctx, vr.VR_OWNER_TILE, rows)
dl = vr.DeferredLightingPass(ctx); sun = [vr.reference_sun()]
views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(120)]
acc = {}
def timed(name, fn, *a):
    t = time.perf_counter(); fn(*a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
N = 600
for it in range(2):
    acc.clear()
    t0 = time.perf_counter()
    for i in range(N):
        v = views[i % 120]
        timed("Render", tp.Render, v, v, rt, rp, part)
        timed("Prepare(i+1)", tp.Prepare, views[(i + 1) % 120], rt, rp, part)
        timed("Prepare(i+2)", tp.Prepare, views[(i + 2) % 120], rt, rp, part)
        timed("Light", dl.Render, v, rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr, part)
    t_issue = time.perf_counter() - t0
    ctx.synchronize()
    t_all = time.perf_counter() - t0
print("per frame: issue %.1f us, device period %.1f us" % (t_issue / N * 1e6, t_all / N * 1e6))
for k, v in acc.items():
    print("  %-14s %.1f us" % (k, v / N * 1e6))
