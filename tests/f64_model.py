"""An exact-arithmetic evaluation of the terrain draw, written from the HLSL / D3D semantics - NOT from oracle/vr_oracle.c.

Test infrastructure: tests/test_oracle_cpu.py bounds the C oracle's raster / sampler model (revision 3: plane equations,
fused sampler arithmetic, a pinned cubic for log2) against this file.  What is taken as given, because the text of the
reference or of D3D pins it:
  - the vertex stage's fp32 outputs (terrain_vs.hlsl:35-62 in its written order; the oracle's vertex_shader, which the
    GPU suite compares with the HIP kernel bit for bit);
  - what the rasteriser is handed per vertex: the fp32 viewport transform and perspective divide and the snap to 8
    sub-pixel bits (D3D11 3.4.1 / 3.4.3): X, Y as 24.8 integers, z = z_clip * (1 / w), 1 / w;
  - D3D's coverage rules: pixel centres at +0.5, top-left rule, in-order LessOrEqual depth test.
Everything after that is evaluated here the way the specification states it, in float64 with exact integer edge
functions - none of the oracle's implementation choices:
  - barycentrics l_i = E_i / (2 area) as exact rationals (E_i and the area are integers below 2^53), depth and 1 / w
    interpolated linearly, attributes perspective-correctly: attr = sum(l_i a_i / w_i) / sum(l_i / w_i);
  - Texture2D::Sample's implicit level of detail from the analytic screen-space derivatives of that interpolant,
    lod = log2(max(|d uv / dx| * dims, |d uv / dy| * dims)) with the true log2;
  - bilinear / trilinear filtering (terrain_ps.hlsl:10-24: uv = (pos + half) / size; clamp addressing) on texels decoded
    in float64 (UNORM8 / 255; sRGB EOTF), main_ps's central differences and normalize (terrain_ps.hlsl:59-63), and the
    render targets' conversions (SRGBA8: round to nearest of the OETF; RGBA16_SNORM: round half away from zero).
Triangles that need the near-plane / guard-band clipper are clipped here in float64 (Sutherland-Hodgman, as D3D's
clipper); pixels they cover are reported separately (their new vertices depend on the clipper's own rounding).
"""
import numpy as np

GUARD_BAND = 100.0
G = 32
S = G + 1


def _tri_indices():
    ci, cj = np.meshgrid(np.arange(G), np.arange(G), indexing="ij")
    bl = (ci * S + cj).ravel()
    tl = bl + S
    tr = tl + 1
    br = bl + 1
    # index buffer (TerrainPass.cpp:68-87): per cell (BL, TL, TR), (BL, TR, BR); row = z, column = x
    return np.stack([np.stack([bl, tl, tr], 1), np.stack([bl, tr, br], 1)], 1).reshape(-1, 3)


_TRIS = _tri_indices()


def _to_screen_f32(clip, vp):
    """What the rasteriser is handed: fp32 perspective divide + viewport transform, snapped to 24.8 (round to nearest)."""
    f = np.float32
    c = clip.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        iw = f(1.0) / c[..., 3]
        nx, ny = c[..., 0] * iw, c[..., 1] * iw
        sx = (nx * f(0.5) + f(0.5)) * f(vp[2]) + f(vp[0])
        sy = (ny * f(-0.5) + f(0.5)) * f(vp[3]) + f(vp[1])
        X = np.floor(sx * f(256.0) + f(0.5))
        Y = np.floor(sy * f(256.0) + f(0.5))
        z = c[..., 2] * iw
    ok = np.isfinite(X) & np.isfinite(Y) & (np.abs(X) < 2.0 ** 40) & (np.abs(Y) < 2.0 ** 40)
    X = np.where(ok, X, 0.0).astype(np.int64)
    Y = np.where(ok, Y, 0.0).astype(np.int64)
    return X, Y, z.astype(np.float64), iw.astype(np.float64)


def _clip_poly64(poly, plane):
    """Sutherland-Hodgman against one plane in float64; poly = list of (clip[4], wx, wz)."""
    def dist(c):
        if plane == 0:
            return c[2]
        if plane == 1:
            return GUARD_BAND * c[3] + c[0]
        if plane == 2:
            return GUARD_BAND * c[3] - c[0]
        if plane == 3:
            return GUARD_BAND * c[3] + c[1]
        return GUARD_BAND * c[3] - c[1]
    out = []
    n = len(poly)
    d = [dist(p[0]) for p in poly]
    for i in range(n):
        j = (i + 1) % n
        ini, inj = d[i] >= 0.0, d[j] >= 0.0
        if ini:
            out.append(poly[i])
        if ini != inj:
            a, b = (poly[i], poly[j]) if ini else (poly[j], poly[i])
            da, db = (d[i], d[j]) if ini else (d[j], d[i])
            t = da / (da - db)
            out.append((a[0] + (b[0] - a[0]) * t, a[1] + (b[1] - a[1]) * t, a[2] + (b[2] - a[2]) * t))
    return out


class Frame:
    pass


def render(ot, view, w, h, max_height, world_size, hm_levels, al_levels, lod_for_sampling=None):
    """The frame TerrainPass::Render draws for `view` into a cleared w x h target, evaluated exactly.
    ot: the oracle terrain (used for NodeSelect and the vertex stage only); hm_levels / al_levels: the mip chains as
    uint8 arrays.  lod_for_sampling: an (h, w) array of levels of detail to sample with instead of this model's own
    (the test passes the oracle's, so that the filter arithmetic is compared at equal LOD and the LOD itself separately)."""
    n, ids, inst = ot.select(view, max_height)
    vp = (float(view.viewport_x), float(view.viewport_y), float(view.viewport_w), float(view.viewport_h))
    vx0, vy0 = max(view.viewport_x, 0), max(view.viewport_y, 0)
    vx1, vy1 = min(view.viewport_x + view.viewport_w - 1, w - 1), min(view.viewport_y + view.viewport_h - 1, h - 1)
    mirrored = bool(view.mirrored)

    depth = np.full((h, w), np.inf)                 # float64 depth of the winner; inf = nothing drawn
    ambiguous = np.zeros((h, w), bool)              # the depth test was decided by less than a few fp32 ulps
    from_clipper = np.zeros((h, w), bool)
    attr = np.zeros((h, w, 7))                      # q, wx, wz, dwx/dx, dwz/dx, dwx/dy, dwz/dy
    tri_count = dict(total=0, rasterised=0, clipped=0)

    def raster(X, Y, z, iw, wx, wz, clipped):
        """One triangle, vertices in submission order (three of each)."""
        area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (X[2] - X[0]) * (Y[1] - Y[0])
        if area2 == 0:
            return
        cw = area2 > 0
        if (not cw) if not mirrored else cw:        # back faces: front = clockwise unless mirrored
            return
        o = [0, 1, 2] if cw else [0, 2, 1]
        X, Y, z, iw, wx, wz = X[o], Y[o], z[o], iw[o], wx[o], wz[o]
        area2 = abs(int(area2))
        x0 = max((int(X.min()) - 128 + 255) >> 8, vx0); x1 = min((int(X.max()) - 128) >> 8, vx1)
        y0 = max((int(Y.min()) - 128 + 255) >> 8, vy0); y1 = min((int(Y.max()) - 128) >> 8, vy1)
        if x0 > x1 or y0 > y1:
            return
        PX = (np.arange(x0, x1 + 1, dtype=np.int64) * 256 + 128)[None, :]
        PY = (np.arange(y0, y1 + 1, dtype=np.int64) * 256 + 128)[:, None]

        def edge(a, b):
            return (X[b] - X[a]) * (PY - Y[a]) - (Y[b] - Y[a]) * (PX - X[a])

        def top_left(a, b):
            dx, dy = X[b] - X[a], Y[b] - Y[a]
            return dy < 0 or (dy == 0 and dx > 0)
        E0, E1, E2 = edge(1, 2), edge(2, 0), edge(0, 1)
        inside = (E0 - (0 if top_left(1, 2) else 1) >= 0) & (E1 - (0 if top_left(2, 0) else 1) >= 0) & (E2 - (0 if top_left(0, 1) else 1) >= 0)
        if not inside.any():
            return
        tri_count["rasterised"] += 1
        a2 = float(area2)
        l0, l1, l2 = E0 / a2, E1 / a2, E2 / a2                          # exact rationals, correctly rounded to float64
        zz = l0 * z[0] + l1 * z[1] + l2 * z[2]
        ok = inside & (zz >= 0.0) & (zz <= 1.0)                          # depth clip
        sub = depth[y0:y1 + 1, x0:x1 + 1]
        ulp = np.spacing(np.maximum(np.abs(zz), 2.0 ** -126).astype(np.float32)).astype(np.float64)
        near = ok & np.isfinite(sub) & (np.abs(zz - sub) <= 4.0 * ulp)
        ambiguous[y0:y1 + 1, x0:x1 + 1] |= near
        win = ok & (zz <= sub)                                           # LessOrEqual, in draw order
        if not win.any():
            return
        q = l0 * iw[0] + l1 * iw[1] + l2 * iw[2]
        nx = l0 * (wx[0] * iw[0]) + l1 * (wx[1] * iw[1]) + l2 * (wx[2] * iw[2])
        nz = l0 * (wz[0] * iw[0]) + l1 * (wz[1] * iw[1]) + l2 * (wz[2] * iw[2])
        # d l_i / dx = d E_i / d px * 256 / area2 (one pixel = 256 sub-pixel units)
        dldx = np.array([-(Y[2] - Y[1]), -(Y[0] - Y[2]), -(Y[1] - Y[0])], np.float64) * 256.0 / a2
        dldy = np.array([(X[2] - X[1]), (X[0] - X[2]), (X[1] - X[0])], np.float64) * 256.0 / a2
        qx, qy = float(dldx @ iw), float(dldy @ iw)
        nxx, nxy = float(dldx @ (wx * iw)), float(dldy @ (wx * iw))
        nzx, nzy = float(dldx @ (wz * iw)), float(dldy @ (wz * iw))
        with np.errstate(divide="ignore", invalid="ignore"):
            pwx, pwz = nx / q, nz / q
            vals = np.stack([q, pwx, pwz, (nxx - pwx * qx) / q, (nzx - pwz * qx) / q, (nxy - pwx * qy) / q, (nzy - pwz * qy) / q], -1)
        sub[win] = zz[win]
        attr[y0:y1 + 1, x0:x1 + 1][win] = vals[win]
        from_clipper[y0:y1 + 1, x0:x1 + 1][win] = clipped

    for i in range(n):
        clip = np.empty((S * S, 4), np.float32)
        world = np.empty((S * S, 3), np.float32)
        for vz in range(S):
            for vx in range(S):
                c, wv = ot.vertex(view, max_height, inst[i], vx, vz)
                clip[vz * S + vx] = c
                world[vz * S + vx] = wv
        X, Y, z, iw = _to_screen_f32(clip, vp)
        wx, wz = world[:, 0].astype(np.float64), world[:, 2].astype(np.float64)
        c3 = clip[_TRIS]                                            # (2048, 3, 4)
        cx, cy, cz, cw_ = c3[..., 0], c3[..., 1], c3[..., 2], c3[..., 3]
        rejected = ((cx < -cw_).all(1) | (cx > cw_).all(1) | (cy < -cw_).all(1) | (cy > cw_).all(1) | (cz < 0).all(1) | (cz > cw_).all(1))
        g = GUARD_BAND * cw_
        hard = ~rejected & ((cz < 0).any(1) | ((cx < -g) | (cx > g) | (cy < -g) | (cy > g)).any(1))
        tri_count["total"] += int((~rejected).sum())
        for t in np.nonzero(~rejected)[0]:
            idx = _TRIS[t]
            if not hard[t]:
                raster(X[idx], Y[idx], z[idx], iw[idx], wx[idx], wz[idx], False)
                continue
            tri_count["clipped"] += 1
            poly = [(clip[k].astype(np.float64), float(wx[k]), float(wz[k])) for k in idx]
            if (np.array([p[0][2] for p in poly]) < 0).any():
                poly = _clip_poly64(poly, 0)
            if len(poly) >= 3:
                gg = [GUARD_BAND * p[0][3] for p in poly]
                if any(p[0][0] < -a or p[0][0] > a or p[0][1] < -a or p[0][1] > a for p, a in zip(poly, gg)):
                    for pl in range(1, 5):
                        poly = _clip_poly64(poly, pl)
                        if len(poly) < 3:
                            break
            if len(poly) < 3:
                continue
            pc = np.array([p[0] for p in poly]).astype(np.float32)
            PXs, PYs, pz, piw = _to_screen_f32(pc, vp)
            pwx = np.array([np.float32(p[1]) for p in poly], np.float64)
            pwz = np.array([np.float32(p[2]) for p in poly], np.float64)
            for k in range(1, len(poly) - 1):
                sel = np.array([0, k, k + 1])
                raster(PXs[sel], PYs[sel], pz[sel], piw[sel], pwx[sel], pwz[sel], True)

    fr = Frame()
    fr.covered = np.isfinite(depth)
    fr.ambiguous = ambiguous
    fr.from_clipper = from_clipper
    fr.depth = np.where(fr.covered, depth, 1.0)
    fr.wx, fr.wz = attr[..., 1], attr[..., 2]
    fr.tri_count = tri_count
    fr.nodes = n
    half = world_size * 0.5
    u, v = (fr.wx + half) / world_size, (fr.wz + half) / world_size
    W0, H0 = hm_levels[0].shape[1], hm_levels[0].shape[0]
    dudx, dvdx, dudy, dvdy = (attr[..., k] / world_size for k in (3, 4, 5, 6))
    with np.errstate(divide="ignore", invalid="ignore"):
        rho = np.maximum(np.hypot(dudx * W0, dvdx * H0), np.hypot(dudy * W0, dvdy * H0))
        fr.lod = np.where(fr.covered, np.log2(rho), 0.0)             # D3D11 7.18.11, isotropic, exact log2; before the sampler's clamp
    lod_s = fr.lod if lod_for_sampling is None else np.asarray(lod_for_sampling, np.float64)

    def decode_height(level):
        return level.astype(np.float64) / 255.0

    def decode_albedo(level):
        c = level[..., :3].astype(np.float64) / 255.0
        return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)

    def bilinear(tex, uu, vv):
        th, tw = tex.shape[0], tex.shape[1]
        x, y = uu * tw - 0.5, vv * th - 0.5
        x0, y0 = np.floor(x), np.floor(y)
        fx, fy = x - x0, y - y0
        xi0 = np.clip(x0, 0, tw - 1).astype(np.int64); xi1 = np.clip(x0 + 1, 0, tw - 1).astype(np.int64)
        yi0 = np.clip(y0, 0, th - 1).astype(np.int64); yi1 = np.clip(y0 + 1, 0, th - 1).astype(np.int64)
        if tex.ndim == 3:
            fx, fy = fx[..., None], fy[..., None]
        top = tex[yi0, xi0] * (1 - fx) + tex[yi0, xi1] * fx
        bot = tex[yi1, xi0] * (1 - fx) + tex[yi1, xi1] * fx
        return top * (1 - fy) + bot * fy

    def trilinear(levels, lod, uu, vv):
        lod = np.clip(np.nan_to_num(lod, nan=0.0, posinf=64.0, neginf=0.0), 0.0, len(levels) - 1)
        l0 = np.floor(lod).astype(np.int64)
        f = lod - l0
        out = None
        for lv in range(len(levels)):
            m0, m1 = l0 == lv, (l0 + 1 == lv) & (f > 0)
            if not (m0.any() or m1.any()):
                continue
            s = bilinear(levels[lv], uu, vv)
            if out is None:
                out = np.zeros(s.shape)
            wgt = np.where(m0, 1 - f, 0.0) + np.where(m1, f, 0.0)
            out = out + s * (wgt[..., None] if s.ndim == 3 else wgt)
        return out

    uu, vv = np.where(fr.covered, u, 0.5), np.where(fr.covered, v, 0.5)
    hml = [decode_height(l) for l in hm_levels]
    all_ = [decode_albedo(l) for l in al_levels]
    off = 0.1                                                            # terrain_ps.hlsl:59
    hDx = trilinear(hml, lod_s, uu + off, vv) - trilinear(hml, lod_s, uu - off, vv)      # :60
    hDy = trilinear(hml, lod_s, uu, vv + off) - trilinear(hml, lod_s, uu, vv - off)      # :61
    nrm = np.stack([-hDx, np.full_like(hDx, 2.0 * off), -hDy], -1)                        # :63
    nrm = nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)
    s16 = np.clip(nrm, -1.0, 1.0) * 32767.0
    fr.normal_codes = np.where(s16 >= 0, np.floor(s16 + 0.5), np.ceil(s16 - 0.5)).astype(np.int64)      # RGBA16_SNORM, round half away
    fr.normal = nrm
    col = trilinear(all_, lod_s, uu, vv)                                                  # :68
    enc = np.where(col <= 0.0031308, col * 12.92, 1.055 * np.maximum(col, 0.0) ** (1.0 / 2.4) - 0.055)
    fr.albedo_codes = np.floor(np.clip(enc, 0.0, 1.0) * 255.0 + 0.5).astype(np.int64)     # SRGBA8, round to nearest
    fr.albedo = col
    return fr
