"""N > 1 on CPU: two gloo ranks each render + light only their interleaved screen tiles (with the
oracle standing in for the device kernels), pack them tile-major, all-gather with equal send counts
and de-tile — the assembled frame must equal the unsplit frame byte for byte (SURVEY §4 iv, §8e); the same for the
tone-mapped exchange (histogram all-reduce + RGB8 tiles)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import vrenderer_amd as vr
    from oracle import pyoracle as po
    from tests.common import AMBIENT_BOTTOM, AMBIENT_TOP, CAMERAS, params, scaled_camera
    from vrenderer_amd import partition as pt

    dist.init_process_group("gloo", rank=rank, world_size=world)
    size = 256
    hm = po.synth_heightmap(size)
    al = po.synth_albedo(size, hm)
    t = po.OracleTerrain(params(size), hm, al)
    eye, tgt = scaled_camera(CAMERAS[5], size)
    v = po.view_from_camera(eye, tgt, w, h)
    gb = po.GBufferHost(w, h)
    t.render(v, gb, vr.default_render_params(400.0), vr.Partition(rank, world))
    hdr = po.deferred(v, gb, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM)
    packed_np = pt.pack(hdr, rank, world)                      # (max_owned, 128, 128, 4) uint16
    packed = torch.from_numpy(packed_np.view(np.uint8).reshape(-1))
    gathered = torch.empty(world * packed.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, packed)              # equal send counts on every rank
    frame = pt.detile(gathered.numpy().view(np.uint16).reshape((-1,) + packed_np.shape[1:]), w, h, world)
    np.save(os.path.join(out_dir, f"frame_{rank}.npy"), frame)
    # the tone-mapped exchange (row f3): own pixels -> histogram, all-reduce of the 256 bins, same exposure everywhere,
    # RGB8 tiles gathered, alpha restored
    tmp = vr.default_tonemap_params()
    tm = po.ToneMapper()
    tm.AdvanceFrame(1.0 / 60.0)
    tm.AddFrameToHistogram(tmp, hdr, vr.Partition(rank, world))
    hist = torch.from_numpy(tm.hist.astype(np.int64))
    dist.all_reduce(hist)
    tm.hist[:] = hist.numpy().astype(np.uint32)
    tm.ComputeExposure(tmp)
    ldr = tm.Render(tmp, hdr)                                   # only this rank's tiles of it are meaningful
    packed_ldr = torch.from_numpy(pt.pack(ldr, rank, world).reshape(-1))
    gathered_ldr = torch.empty(world * packed_ldr.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered_ldr, packed_ldr)
    np.save(os.path.join(out_dir, f"ldr_{rank}.npy"), pt.detile(gathered_ldr.numpy().reshape((-1,) + pt.packed_shape(w, h, world)[1:]), w, h, world, alpha=255))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tile_split_allgather_equals_unsplit(oracle, tmp_path):
    import torch.multiprocessing as mp
    import vrenderer_amd as vr
    from tests.common import AMBIENT_BOTTOM, AMBIENT_TOP, CAMERAS, params, scaled_camera

    w, h, world = 384, 300, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    size = 256
    hm = oracle.synth_heightmap(size)
    al = oracle.synth_albedo(size, hm)
    t = oracle.OracleTerrain(params(size), hm, al)
    eye, tgt = scaled_camera(CAMERAS[5], size)
    v = oracle.view_from_camera(eye, tgt, w, h)
    gb = oracle.GBufferHost(w, h)
    t.render(v, gb, vr.default_render_params(400.0))
    ref = oracle.deferred(v, gb, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM)
    tm = oracle.ToneMapper()
    tm.AdvanceFrame(1.0 / 60.0)
    ref_ldr = tm.SimpleRender(vr.default_tonemap_params(), ref)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"frame_{r}.npy"))
        assert np.array_equal(got, ref), f"rank {r}: assembled frame differs from the unsplit frame"
        got_ldr = np.load(os.path.join(str(tmp_path), f"ldr_{r}.npy"))
        assert np.array_equal(got_ldr, ref_ldr), f"rank {r}: assembled tone-mapped frame differs from the unsplit one"
