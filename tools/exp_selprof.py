import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from vrenderer_amd import capi
capi.LIB_PATH = os.path.join(ROOT, "vrenderer_amd", "lib", "variants", "selprof", "libvrterrain.so")
import vrenderer_amd as vr
from vrenderer_amd.scene import params, flythrough_camera
ctx = vr.Context(0); ctx.set_async_geometry(False)
hm = vr.synth_heightmap(ctx, 2048); al = vr.synth_albedo(ctx, 2048, hm)
tp = vr.TerrainPass(ctx, params(2048)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(1920, 1080)
rp = vr.default_render_params(400.0, assume_cleared=1)
for i in (0, 30, 60, 90):
    v = vr.make_view(*flythrough_camera(i), 1920, 1080); tp.Render(v, v, rt, rp); print(tp.render_stats())
