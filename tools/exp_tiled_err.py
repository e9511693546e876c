"""Scratch: error analysis of the tiled lighting pass vs the oracle's all-lights loop at config-5 scale."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vrenderer_amd as vr
from oracle import pyoracle as po
from vrenderer_amd.scene import params, AMBIENT_TOP, AMBIENT_BOTTOM, flythrough_camera
po.build()
size, w, h = 2048, 960, 540
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
v = vr.make_view(*flythrough_camera(30), w, h)
rt = vr.RenderTargets(ctx).Init(w, h)
tp.Render(v, v, rt, vr.default_render_params(400.0))
gb = po.GBufferHost(w, h)
for name, dst in (("depth", gb.depth), ("diffuse", gb.diffuse), ("specular", gb.specular), ("normals", gb.normals), ("emissive", gb.emissive)):
    dst[...] = rt.download(name).reshape(dst.shape)
lights = [vr.reference_sun()] + vr.synthetic_point_lights(1023, 2048.0, hm, 400.0, seed=9001)
ref = po.deferred(v, gb, lights, AMBIENT_TOP, AMBIENT_BOTTOM, f32=True).astype(np.float64)
hdr = vr.HdrImage(ctx, w, h)
vr.TiledDeferredLightingPass(ctx).Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
got = po.half_to_float(hdr.download()).astype(np.float64)
err = np.abs(got[..., :3] - ref[..., :3]).max(axis=2)
print("rms", np.sqrt(np.mean((got[..., 0] - ref[..., 0]) ** 2)), "max", err.max(), "at", np.unravel_index(err.argmax(), err.shape))
rel = err / np.maximum(np.abs(ref[..., :3]).max(axis=2), 1e-6)
print("pixels with rel err > 2^-10:", (rel > 2.0 ** -10).sum(), "of", err.size)
bad = rel > 2.0 ** -10
ys, xs = np.nonzero(bad)
if len(ys):
    tiles = set(zip((ys // 32).tolist(), (xs // 32).tolist()))
    print("bad pixels lie in", len(tiles), "32x32 tiles; sample", sorted(tiles)[:10])
    for y, x in list(zip(ys, xs))[:5]:
        print((y, x), "got", got[y, x, :3], "ref", ref[y, x, :3], "depth", gb.depth[y, x])
# streaming 16-light kernel on the 16 strongest lights at the worst pixel is not informative; compare half rounding level
print("rms of the format's own rounding (oracle half vs float):",
      np.sqrt(np.mean((po.half_to_float(po.deferred(v, gb, lights, AMBIENT_TOP, AMBIENT_BOTTOM)).astype(np.float64)[..., 0] - ref[..., 0]) ** 2)))
