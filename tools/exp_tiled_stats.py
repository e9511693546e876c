"""Scratch (GPU box): how long are the 32x32 light tiles' lists at config-5 scale, against how many lights actually reach a pixel."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vrenderer_amd as vr
from vrenderer_amd.scene import params, flythrough_camera
size, w, h = 2048, 7680, 4320
ctx = vr.Context(0)
hm = vr.synth_heightmap(ctx, size); al = vr.synth_albedo(ctx, size, hm)
tp = vr.TerrainPass(ctx, params(size)).Init(hm, al)
rt = vr.RenderTargets(ctx).Init(w, h)
lights = vr.synthetic_point_lights(1023, 2048.0, hm, 400.0, seed=9001)
lp = np.array([[l.position[0], l.position[1], l.position[2]] for l in lights], np.float64)
lr = np.array([1.0 / l.angular_size_or_inv_range for l in lights], np.float64)
for f in (30, 90):
    v = vr.make_view(*flythrough_camera(f), w, h)
    tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1))
    depth = rt.download("depth").astype(np.float64)
    c2w = np.array(list(v.clip_to_world), np.float64).reshape(4, 4)
    ys, xs = np.mgrid[0:h, 0:w]
    cx = (xs + 0.5) * (2.0 / w) - 1.0; cy = (ys + 0.5) * (-2.0 / h) + 1.0
    P = np.stack([cx, cy, depth, np.ones_like(depth)], -1) @ c2w
    wp = P[..., :3] / P[..., 3:4]
    cov = depth < 1.0
    # per pixel (every 8th): lights in range
    s = (slice(4, None, 8), slice(4, None, 8))
    q = wp[s][cov[s]]
    cnt = np.zeros(len(q))
    for i in range(0, len(q), 20000):
        d2 = ((q[i:i + 20000, None, :] - lp[None]) ** 2).sum(-1)
        cnt[i:i + 20000] = (d2 < lr[None] ** 2).sum(1)
    # per 32x32 tile: lights touching the box of its covered pixels' positions (what per-pixel culling could reach at best)
    th, tw = h // 32, w // 32
    wpt = wp[:th * 32, :tw * 32].reshape(th, 32, tw, 32, 3); ct = cov[:th * 32, :tw * 32].reshape(th, 32, tw, 32)
    big = 1e30
    lo = np.where(ct[..., None], wpt, big).min(axis=(1, 3)); hi = np.where(ct[..., None], wpt, -big).max(axis=(1, 3))
    anyc = ct.any(axis=(1, 3))
    d = np.maximum(np.maximum(lo[:, :, None, :] - lp[None, None], lp[None, None] - hi[:, :, None, :]), 0.0)
    touch = ((d ** 2).sum(-1) <= lr[None, None] ** 2) & anyc[..., None]
    per_tile = touch.sum(-1)
    npx = ct.sum(axis=(1, 3))
    # the cull kernel's box: the tile's frustum cell over its covered depth range (8 corners), padded
    dt = depth[:th * 32, :tw * 32].reshape(th, 32, tw, 32)
    dmin = np.where(ct, dt, 2.0).min(axis=(1, 3)); dmax = np.where(ct, dt, -1.0).max(axis=(1, 3))
    tyy, txx = np.mgrid[0:th, 0:tw]
    clo = np.full((th, tw, 3), big); chi = np.full((th, tw, 3), -big); far2 = np.zeros((th, tw))
    cam = np.array(list(v.camera_pos), np.float64)[:3]
    for cxi in (0, 1):
        for cyi in (0, 1):
            for dz in (dmin, dmax):
                wx = (txx + cxi) * 32.0; wy = (tyy + cyi) * 32.0
                Pc = np.stack([wx * (2.0 / w) - 1.0, wy * (-2.0 / h) + 1.0, dz, np.ones_like(dz)], -1) @ c2w
                pc = Pc[..., :3] / Pc[..., 3:4]
                clo = np.minimum(clo, pc); chi = np.maximum(chi, pc); far2 = np.maximum(far2, ((pc - cam) ** 2).sum(-1))
    pad = 4.0e-3 * np.sqrt(far2) + 1.0e-2
    clo -= pad[..., None]; chi += pad[..., None]
    d = np.maximum(np.maximum(clo[:, :, None, :] - lp[None, None], lp[None, None] - chi[:, :, None, :]), 0.0)
    touch2 = ((d ** 2).sum(-1) <= lr[None, None] ** 2 * 1.0001) & anyc[..., None]
    per_tile2 = touch2.sum(-1)
    print("   cull kernel's boxes: pixel-weighted mean %.2f" % ((per_tile2 * npx).sum() / npx.sum()), "max", per_tile2.max())
    print("frame", f, "lights in range per covered pixel: mean %.2f" % cnt.mean(), "| lights per tile (pixel-position box), pixel-weighted mean %.2f" % ((per_tile * npx).sum() / npx.sum()),
          "max", per_tile.max())
