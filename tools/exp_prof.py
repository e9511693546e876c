"""Scratch: per-phase wave cycles of k_raster from the VR_RASTER_PROFILE variant (tools/build_variant.py prof -DVR_RASTER_PROFILE)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vrenderer_amd import capi
capi.LIB_PATH = os.path.join(ROOT, "vrenderer_amd", "lib", "variants", os.environ.get("VARIANT", "prof"), "libvrterrain.so")
import vrenderer_amd as vr
from tests.common import params
from bench import flythrough_camera

W, H, size = int(os.environ.get("W", 7680)), int(os.environ.get("H", 4320)), 2048
ctx = vr.Context(0); ctx.set_async_geometry(False)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(W, H)
lib = capi.load_library()
names = ["init", "fetch records", "tiny sweeps", "row sweeps", "big sweeps", "barrier wait", "(unused)", "(counts)",
         "res: loop head + vis + record prefetch", "res: interp (waits for the record)", "res: uv, LOD, addresses, issue level 0", "res: wait for level 0", "res: filter level 0",
         "res: level 1 (if any)", "res: normal + encodes", "res: stores"]
def run(name, view, **kw):
    rp = vr.default_render_params(400.0, **kw)
    buf = (C.c_ulonglong * 16)()
    tp.Render(view, view, rt, rp); ctx.synchronize()
    lib.vr_debug_raster_prof(buf, 1)
    ctx.timing_enable(True)
    N = 5
    for _ in range(N): tp.Render(view, view, rt, rp)
    ctx.synchronize()
    t = ctx.timing_collect(); ctx.timing_enable(False)
    lib.vr_debug_raster_prof(buf, 1)
    tot = sum(buf[:7]) + sum(buf[8:16])
    kr = t.get("k_raster") or t.get("k_raster (depth only)")
    print(name, "k_raster %.1f us" % (kr[0] / kr[1] * 1e3),
          {n: "%.1f%%" % (100.0 * buf[i] / tot) for i, n in enumerate(names) if i != 7 and buf[i]},
          "wave-cycles/launch %.3g" % (tot / N),
          "triangles per launch: tiny %d, row %d, big %d" % ((buf[7] & 0xfffff) // N, ((buf[7] >> 20) & 0xfffff) // N, (buf[7] >> 40) // N), flush=True)
for i in (30, 90):
    fly = vr.make_view(*flythrough_camera(i), W, H)
    run(f"fly{i} depth-only", fly, assume_cleared=1, depth_only=1)
    run(f"fly{i} full      ", fly, assume_cleared=1)
