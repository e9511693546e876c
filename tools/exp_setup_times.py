"""Scratch: how long the bench's set-up steps take on the host (what precedes the warm-up frames)."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t = [time.perf_counter()]
def mark(what):
    t.append(time.perf_counter()); print(f"{what:42s} {1e3 * (t[-1] - t[-2]):9.1f} ms", flush=True)
import numpy as np
import vrenderer_amd as vr
from vrenderer_amd.scene import params, flythrough_camera
mark("imports")
ctx = vr.Context(0); mark("context")
hm = vr.synth_heightmap(ctx, 2048, 1337); al = vr.synth_albedo(ctx, 2048, hm, 4242); mark("synthetic textures (device + download)")
tp = vr.TerrainPass(ctx, params(2048)).Init(hm, al); ctx.synchronize(); mark("TerrainPass.Init (tables, scratch)")
rt = vr.RenderTargets(ctx).Init(7680, 4320); ctx.synchronize(); mark("RenderTargets.Init (929 MB + clear)")
hdr = vr.HdrImage(ctx, 7680, 4320); mark("HdrImage")
views = [vr.make_view(*flythrough_camera(i), 7680, 4320) for i in range(120)]; mark("120 views")
gc.collect(); mark("gc.collect")
