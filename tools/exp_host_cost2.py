import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import ctypes as C
import vrenderer_amd as vr
from vrenderer_amd.scene import params, AMBIENT_TOP, AMBIENT_BOTTOM, flythrough_camera
from vrenderer_amd.passes import partition_info
import torch
W, H, size = 7680, 4320, 2048
ctx = vr.Context(0)
ms = torch.cuda.current_stream(); ctx.set_stream(ms.cuda_stream)
cs = torch.cuda.Stream(); ctx2 = vr.Context(0); ctx2.set_stream(cs.cuda_stream)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
rp = vr.default_render_params(400.0, assume_cleared=1)
world = int(os.environ.get("WORLD", 8))
part = vr.Partition(0, world) if world > 1 else None
info = partition_info(W, H, 0, world)
rows = (info["packed_bytes"] + vr.VR_OWNER_TILE * 8 - 1) // (vr.VR_OWNER_TILE * 8)
hdr = [vr.HdrImage(ctx, vr.VR_OWNER_TILE, rows) if part else vr.HdrImage(ctx, W, H) for _ in range(2)]
sun = [vr.reference_sun()]
views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(120)]
tm = vr.ToneMappingPass(ctx2); tm.AdvanceFrame(1/60); tmp = vr.default_tonemap_params(); ldr = vr.LdrImage(ctx2, W, H)
lib = ctx.lib
def loop(name, fn, N=600):
    for it in range(2):
        t0 = time.perf_counter()
        for i in range(N): fn(i)
        ti = time.perf_counter() - t0
        ctx.synchronize(); ctx2.synchronize()
        ta = time.perf_counter() - t0
    print("%-40s issue %.1f us  period %.1f us" % (name, ti / N * 1e6, ta / N * 1e6), flush=True)
dl = vr.DeferredLightingPass(ctx)
loop("Render only", lambda i: tp.Render(views[i % 120], views[i % 120], rt, rp, part))
def rp2(i):
    tp.Render(views[i % 120], views[i % 120], rt, rp, part); tp.Prepare(views[(i + 1) % 120], rt, rp, part); tp.Prepare(views[(i + 2) % 120], rt, rp, part)
loop("Render + Prepare x2", rp2)
def rpl(i):
    rp2(i); dl.Render(views[i % 120], rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr[i % 2], part)
loop("Render + Prepare x2 + Light", rpl)
f1 = vr.Frame(tp, rt, rp, sun, AMBIENT_TOP, AMBIENT_BOTTOM, part)
loop("submit (no tone map)", lambda i: f1.submit(views[i % 120], hdr[i % 2], [views[(i + 1) % 120], views[(i + 2) % 120]]))
f2 = vr.Frame(tp, rt, rp, sun, AMBIENT_TOP, AMBIENT_BOTTOM, part, tonemap=tm, tonemap_params=tmp, ldr=ldr)
loop("submit + tone-map stage", lambda i: f2.submit(views[i % 120], hdr[i % 2], [views[(i + 1) % 120], views[(i + 2) % 120]]))
def tmonly(i):
    tm.ResetHistogram(); tm.AddFrameToHistogram(tmp, hdr[0], W, H, part); tm.ComputeExposure(tmp); tm.Render(tmp, hdr[0], ldr, W, H, part)
loop("tone-map stage alone (4 calls)", tmonly)
