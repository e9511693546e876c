"""Scratch: two frames in flight - frame i's lighting pass on a second stream (its own context) while frame i+1's tile pass
runs on the first; two G-buffers.  Does the bandwidth-bound lighting pass overlap usefully with the tile pass?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import vrenderer_amd as vr
from vrenderer_amd.scene import params, AMBIENT_TOP, AMBIENT_BOTTOM, flythrough_camera
W, H, size = 7680, 4320, 2048
torch.cuda.set_device(0)
PRIO = os.environ.get("PRIO", "none")      # none | main (tile pass on a high-priority stream) | light (lighting pass on one)
s_main = torch.cuda.Stream(priority=-1 if PRIO == "main" else 0)
s_light = torch.cuda.Stream(priority=-1 if PRIO == "light" else 0)
ctx = vr.Context(0); ctx.set_stream(s_main.cuda_stream)
ctx2 = vr.Context(0); ctx2.set_stream(s_light.cuda_stream)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rts = [vr.RenderTargets(ctx).Init(W, H) for _ in range(2)]
hdr = vr.HdrImage(ctx2, W, H)
rp = vr.default_render_params(400.0, assume_cleared=1)
sun = [vr.reference_sun()]
dl1 = vr.DeferredLightingPass(ctx); dl2 = vr.DeferredLightingPass(ctx2)
views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(120)]
ras_done = [torch.cuda.Event() for _ in range(2)]
lit_done = [torch.cuda.Event() for _ in range(2)]
def run(overlap, n):
    for i in range(n):
        b = i % 2 if overlap else 0
        v = views[i % 120]
        if overlap:
            s_main.wait_event(lit_done[b])                 # the G-buffer of two frames ago has been lit
        tp.Render(v, v, rts[b], rp)
        tp.Prepare(views[(i + 1) % 120], rts[(i + 1) % 2 if overlap else 0], rp)
        tp.Prepare(views[(i + 2) % 120], rts[i % 2 if overlap else 0], rp)
        if overlap:
            ras_done[b].record(s_main)
            s_light.wait_event(ras_done[b])
            dl2.Render(v, rts[b], sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
            lit_done[b].record(s_light)
        else:
            dl1.Render(v, rts[0], sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    torch.cuda.synchronize()
for overlap in (False, True, False, True):
    run(overlap, 10)
    t0 = time.perf_counter(); run(overlap, 120); dt = time.perf_counter() - t0
    print("prio", PRIO, "two frames in flight" if overlap else "one frame at a time ", "%.1f us per frame = %.1f Gpixels/s" % (dt / 120 * 1e6, W * H * 120 / dt / 1e9), flush=True)
