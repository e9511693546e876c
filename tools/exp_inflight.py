"""Scratch: two frames in flight on two streams (two contexts, two full pipelines) vs one."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrenderer_amd as vr
from tests.common import AMBIENT_BOTTOM, AMBIENT_TOP, params
from bench import flythrough_camera
W, H, size = 7680, 4320, 2048
hip = None
NF = int(os.environ.get("NF", 2))
ctxs = []
lib = vr.load_library()
hiprt = C.CDLL("libamdhip64.so.7") if False else None
pipes = []
import torch
streams = [torch.cuda.Stream() for _ in range(NF)]
for k in range(NF):
    ctx = vr.Context(0)
    ctx.set_stream(streams[k].cuda_stream)
    hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
    tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
    rt = vr.RenderTargets(ctx).Init(W, H)
    hdr = vr.HdrImage(ctx, W, H)
    pipes.append((ctx, tp, rt, hdr, vr.DeferredLightingPass(ctx)))
views = [vr.make_view(*flythrough_camera(i), W, H) for i in range(120)]
rp = vr.default_render_params(400.0, assume_cleared=1)
def frame(i):
    ctx, tp, rt, hdr, dl = pipes[i % NF]
    v = views[i % 120]
    tp.Render(v, v, rt, rp)
    dl.Render(v, rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
for i in range(6): frame(i)
torch.cuda.synchronize()
for p in pipes: p[0].timing_enable(True)
t0 = time.perf_counter()
N = 60
for i in range(N): frame(6 + i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tm = {}
for p in pipes:
    for k, (ms, n) in p[0].timing_collect().items():
        a = tm.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += n
print("frames in flight", NF, "ms/frame", round(dt / N * 1e3, 4), "Gpx/s", round(W * H * N / dt / 1e9, 2),
      {k: round(a[0] / a[1] * 1e3, 1) for k, a in tm.items()})
