"""Generates tests/golden/*.npz from the CPU oracle (oracle/vr_oracle.c).

The reference (Viictor/vrenderer) ships no tests, fixtures or golden vectors and
cannot be built or run in this pipeline (Win32/D3D12 + the empty Donut submodule), so
these vectors are produced by the project's own CPU restatement — "parity unpinned" —
except for the node counts recorded in SURVEY.md §6/§8a, which the survey measured on
the reference's own QuadTree.cpp and which tests/test_oracle_cpu.py checks separately.

Run:  python tests/golden/make_golden.py      (rewrites the fixtures deterministically)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402
import vrenderer_amd as vr  # noqa: E402
from tests.common import AMBIENT_BOTTOM, AMBIENT_TOP, CAMERAS, params, scaled_camera  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    po.build()
    # 1. selected node-id lists: 8 cameras x {L=8 (256), L=11 (2048)}
    sel = {}
    for size in (256, 2048):
        h = po.synth_heightmap(size)
        a = po.synth_albedo(size, h)
        t = po.OracleTerrain(params(size), h, a)
        for ci, cam in enumerate(CAMERAS):
            eye, tgt = scaled_camera(cam, size)
            v = po.view_from_camera(eye, tgt, 1920, 1080)
            n, ids, _ = t.select(v, 400.0)
            sel[f"ids_{size}_{ci}"] = ids
        sel[f"height_crc_{size}"] = np.array([int(h.astype(np.uint64).sum()), int((h.astype(np.uint64) * np.arange(h.size, dtype=np.uint64).reshape(h.shape) % 65521).sum())], np.uint64)
        if size == 256:
            # 2. a 256x144 G-buffer + HDR frame of the 256 terrain
            w, hh = 256, 144
            eye, tgt = scaled_camera(CAMERAS[0], size)
            v = po.view_from_camera(eye, tgt, w, hh)
            gb = po.GBufferHost(w, hh)
            rp = vr.default_render_params(400.0)
            t.render(v, gb, rp)
            hdr = po.deferred(v, gb, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM)
            # The oracle's raster / sampler model is frozen per revision: one digest line per revision, never rewritten.
            import hashlib
            m = hashlib.sha256()
            for arr in (gb.depth, gb.diffuse, gb.specular, gb.normals, gb.emissive, hdr):
                m.update(np.ascontiguousarray(arr).tobytes())
            rev, digest = po.model_revision(), m.hexdigest()
            rev_file = os.path.join(OUT, "MODEL_REVISIONS.txt")
            known = {}
            if os.path.exists(rev_file):
                known = {int(l.split()[0]): l.split()[1] for l in open(rev_file) if l.strip() and not l.startswith("#")}
            if rev in known and known[rev] != digest:
                raise SystemExit(f"the oracle's output changed but ORC_MODEL_REVISION is still {rev}: bump it (oracle/vr_oracle.c), "
                                 "state the diff statistics against tests/f64_model.py in the commit, then run this again")
            if rev not in known:
                with open(rev_file, "a") as f:
                    if not known:
                        f.write("# model revision -> sha256 of the golden 256x144 frame's planes (depth, diffuse, specular, normals, emissive, hdr)\n")
                    f.write(f"{rev} {digest}\n")
            np.savez_compressed(os.path.join(OUT, "frame_256x144.npz"), depth=gb.depth, diffuse=gb.diffuse,
                                specular=gb.specular, normals=gb.normals, emissive=gb.emissive, hdr=hdr,
                                view=np.frombuffer(bytes(v), np.uint8), model_revision=np.int32(rev))
        t.close()
    np.savez_compressed(os.path.join(OUT, "select_ids.npz"), **sel)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
