"""GPU parity: the HIP path through the C ABI vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import vrenderer_amd as vr
from tests.common import AMBIENT_BOTTOM, AMBIENT_TOP, CAMERAS, params, scaled_camera

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene256(oracle, gpu_ctx):
    size = 256
    h = oracle.synth_heightmap(size)
    a = oracle.synth_albedo(size, h)
    ot = oracle.OracleTerrain(params(size), h, a)
    tp = vr.TerrainPass(gpu_ctx, params(size)).Init(h, a)
    yield dict(size=size, h=h, a=a, ot=ot, tp=tp)
    tp.close()
    ot.close()


@pytest.fixture(scope="module")
def scene2048(oracle, gpu_ctx):
    size = 2048
    h = oracle.synth_heightmap(size)
    a = oracle.synth_albedo(size, h)
    ot = oracle.OracleTerrain(params(size), h, a)
    tp = vr.TerrainPass(gpu_ctx, params(size)).Init(h, a)
    yield dict(size=size, h=h, a=a, ot=ot, tp=tp)
    tp.close()
    ot.close()


def test_synth_inputs_bit_exact(oracle, gpu_ctx):
    for size in (256, 512):
        h_gpu = vr.synth_heightmap(gpu_ctx, size, 1337)
        h_cpu = oracle.synth_heightmap(size, 1337)
        assert np.array_equal(h_gpu, h_cpu)
        a_gpu = vr.synth_albedo(gpu_ctx, size, h_cpu, 4242)
        a_cpu = oracle.synth_albedo(size, h_cpu, 4242)
        assert np.array_equal(a_gpu, a_cpu)


def test_mip_chains_bit_exact(scene256):
    ot, tp = scene256["ot"], scene256["tp"]
    assert tp.mip_levels("height") == ot.height_levels()
    assert tp.mip_levels("albedo") == ot.albedo_levels()
    for l in range(ot.height_levels()):
        assert np.array_equal(tp.download_mip("height", l), ot.height_mip(l)), f"height level {l}"
    for l in range(ot.albedo_levels()):
        assert np.array_equal(tp.download_mip("albedo", l), ot.albedo_mip(l)), f"albedo level {l}"


@pytest.mark.parametrize("scene_name", ["scene256", "scene2048"])
def test_select_node_ids_and_instances_bit_exact(scene_name, request, oracle):
    sc = request.getfixturevalue(scene_name)
    ot, tp, size = sc["ot"], sc["tp"], sc["size"]
    assert tp.GetNumLods() == ot.num_lods
    assert np.array_equal(tp.GetLodRanges(), ot.lod_ranges())
    for cam in CAMERAS:
        eye, tgt = scaled_camera(cam, size)
        for (w, h) in ((1920, 1080), (640, 480)):
            v = vr.make_view(eye, tgt, w, h)
            n_o, ids_o, inst_o = ot.select(v, 400.0)
            n_g, ids_g, inst_g = tp.NodeSelect(v, 400.0)
            assert n_g == n_o, (cam, n_g, n_o)
            assert np.array_equal(ids_g, ids_o), cam
            assert np.array_equal(inst_g, inst_o), cam


def test_vertex_stage_bit_exact(scene2048, oracle, gpu_ctx):
    """main_vs on its own (terrain_vs.hlsl:10-25, :35-62): o_position and the world xz of every one of the 1,089 grid
    vertices of chosen instances - the nearest node, one whose vertices lie in the morph band (0 < morphK < 1), the
    coarsest, and the last of the draw - bit for bit against the oracle's vertex_shader, not through rasterised depth."""
    ot, tp = scene2048["ot"], scene2048["tp"]
    eye, tgt = CAMERAS[0]
    w, h = 960, 540
    v = vr.make_view(eye, tgt, w, h)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    tp.Render(v, v, rt, vr.default_render_params(400.0))
    n, ids, inst = ot.select(v, 400.0)
    assert tp.num_chunks() == n and n > 8
    ext = inst.view(np.float32)[:, 4]                       # transform[0] = the node's half-extent
    pos = inst.view(np.float32)[:, [7, 15]]                 # transform[3], transform[11] = position x, z
    dist = np.hypot(pos[:, 0] - eye[0], pos[:, 1] - eye[2])
    ranges = ot.lod_ranges()
    # a node that straddles its LOD range's morph start (0.85 r): some of its vertices morph partially
    lod = np.clip(np.floor(np.log2(2.0 * ext)).astype(int), 0, 11)
    band = np.abs(dist - 0.925 * ranges[lod]) - ext
    chosen = sorted({int(np.argmin(dist)), int(np.argmin(band)), int(np.argmax(ext)), n - 1, 0})
    partial = 0
    for i in chosen:
        got = tp.download_vertices(i, 1)[0]
        want = np.empty((1089, 6), np.float32)
        for vz in range(33):
            for vx in range(33):
                clip, world = ot.vertex(v, 400.0, inst[i], vx, vz)
                want[vz * 33 + vx, :4] = clip
                want[vz * 33 + vx, 4] = world[0]; want[vz * 33 + vx, 5] = world[2]
        mis = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        assert mis.size == 0, f"instance {i} (node {ids[i]}): {len(mis)} vertex components differ, first {mis[:4].tolist()}"
        # odd grid vertices move when morphK > 0: count vertices strictly between the two mesh resolutions
        base = pos[i, 0] + ext[i] * ((np.arange(33) - 16) / 16.0)
        moved = np.abs(got[:33, 4] - base.astype(np.float32))
        step = 2.0 * ext[i] / 32.0
        partial += int(np.sum((moved > 1e-6 * step) & (moved < step * (1.0 - 1e-6))))
    assert partial > 0, "no chosen instance had a partially morphed vertex - the morph band was not exercised"
    rt.close()


def test_config2_1080p_single_surface_at_its_own_size(scene256, oracle, gpu_ctx):
    """BASELINE config 2 at its own size: 1920x1080, the 256^2 single-surface scene, the reference camera
    (Renderer.cpp:97,315) scaled to the surface, 1 directional light: G-buffer bit-exact, HdrColor RMS <= 1e-4."""
    eye, tgt = scaled_camera(CAMERAS[0], 256)
    w, h = 1920, 1080
    v, gb_o, planes, n_o, n_g = _render_both(scene256, oracle, gpu_ctx, eye, tgt, w, h, assume_cleared=1)
    assert n_o == n_g
    _assert_gbuffer_equal(gb_o, planes, "config 2, 1920x1080")
    assert (gb_o.depth < 1.0).mean() > 0.05
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    tp = scene256["tp"]
    tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1))
    hdr = vr.HdrImage(gpu_ctx, w, h)
    vr.DeferredLightingPass(gpu_ctx).Render(v, rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    got = oracle.half_to_float(hdr.download()).astype(np.float64)
    want = oracle.half_to_float(oracle.deferred(v, gb_o, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM)).astype(np.float64)
    rms = np.sqrt(np.mean((got - want) ** 2, axis=(0, 1)))
    assert (rms <= 1e-4).all(), f"per-channel HDR RMS {rms} exceeds 1e-4"
    hdr.close(); rt.close()


def _render_both(sc, oracle, gpu_ctx, eye, tgt, w, h, assume_cleared=0, depth_only=0, part=None, wireframe=0):
    ot, tp = sc["ot"], sc["tp"]
    v = vr.make_view(eye, tgt, w, h)
    rp = vr.default_render_params(400.0, assume_cleared=assume_cleared, depth_only=depth_only, wireframe=wireframe)
    gb_o = oracle.GBufferHost(w, h)
    n_o = ot.render(v, gb_o, rp, part)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    tp.Render(v, v, rt, rp, part)
    n_g = tp.num_chunks()
    planes = {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}
    rt.close()
    return v, gb_o, planes, n_o, n_g


def _assert_gbuffer_equal(gb_o, planes, what=""):
    dmis = np.argwhere(planes["depth"].view(np.uint32) != gb_o.depth.view(np.uint32))
    assert dmis.size == 0, f"{what}: depth differs at {len(dmis)} pixels, first {dmis[:5].tolist()}"
    for name, ref in (("diffuse", gb_o.diffuse), ("specular", gb_o.specular), ("normals", gb_o.normals),
                      ("emissive", gb_o.emissive)):
        mis = np.argwhere(planes[name] != ref)
        assert mis.size == 0, f"{what}: {name} differs at {len(mis)} entries, first {mis[:5].tolist()}"


@pytest.mark.parametrize("cam_index", range(len(CAMERAS)))
def test_gbuffer_bit_exact_256(scene256, oracle, gpu_ctx, cam_index):
    eye, tgt = scaled_camera(CAMERAS[cam_index], 256)
    v, gb_o, planes, n_o, n_g = _render_both(scene256, oracle, gpu_ctx, eye, tgt, 640, 360)
    assert n_o == n_g
    _assert_gbuffer_equal(gb_o, planes, f"camera {cam_index}")


@pytest.mark.parametrize("cam_index", [0, 1, 5])
def test_gbuffer_bit_exact_2048(scene2048, oracle, gpu_ctx, cam_index):
    eye, tgt = CAMERAS[cam_index]
    v, gb_o, planes, n_o, n_g = _render_both(scene2048, oracle, gpu_ctx, eye, tgt, 960, 540)
    assert n_o == n_g
    _assert_gbuffer_equal(gb_o, planes, f"camera {cam_index}")


def test_gbuffer_assume_cleared_matches_clear_then_render(scene256, oracle, gpu_ctx):
    eye, tgt = scaled_camera(CAMERAS[7], 256)     # far camera: sky pixels present
    v, gb_o, planes, _, _ = _render_both(scene256, oracle, gpu_ctx, eye, tgt, 644, 362, assume_cleared=1)
    assert (gb_o.depth == 1.0).any(), "test needs uncovered pixels"
    _assert_gbuffer_equal(gb_o, planes, "assume_cleared")


def test_gbuffer_depth_only(scene256, oracle, gpu_ctx):
    eye, tgt = scaled_camera(CAMERAS[0], 256)
    v, gb_o, planes, _, _ = _render_both(scene256, oracle, gpu_ctx, eye, tgt, 320, 200, depth_only=1)
    _assert_gbuffer_equal(gb_o, planes, "depth_only")
    assert not planes["diffuse"].any()


@pytest.mark.parametrize("cam_index", [0, 3, 7])
def test_wireframe_bit_exact(scene256, oracle, gpu_ctx, cam_index):
    """RasterFillMode::Wireframe (EditorParams::m_Wireframe, TerrainPass.cpp:476)."""
    eye, tgt = scaled_camera(CAMERAS[cam_index], 256)
    v, gb_o, planes, n_o, n_g = _render_both(scene256, oracle, gpu_ctx, eye, tgt, 640, 360, wireframe=1)
    assert n_o == n_g
    _assert_gbuffer_equal(gb_o, planes, f"wireframe camera {cam_index}")
    drawn = planes["depth"] < 1.0
    assert drawn.any() and not drawn.all()


def test_wireframe_near_clipped_and_partitioned(scene256, oracle, gpu_ctx):
    hgt = float(scene256["h"][128 + 3, 128 + 5]) / 255.0 * 400.0
    eye, tgt = (5.3, hgt + 0.05, 3.2), (60.0, hgt - 5.0, 40.0)
    v, gb_o, planes, _, _ = _render_both(scene256, oracle, gpu_ctx, eye, tgt, 640, 360, wireframe=1)
    _assert_gbuffer_equal(gb_o, planes, "wireframe near-plane")
    part = vr.Partition(1, 2)
    v, gb_o, planes, _, _ = _render_both(scene256, oracle, gpu_ctx, eye, tgt, 640, 360, wireframe=1, part=part)
    _assert_gbuffer_equal(gb_o, planes, "wireframe rank 1 of 2")


def test_camera_inside_terrain_near_plane_clipping(scene256, oracle, gpu_ctx):
    # camera a fraction of a unit above the surface, looking along it: triangles cross z = 0
    size = 256
    hgt = float(scene256["h"][128 + 3, 128 + 5]) / 255.0 * 400.0
    eye = (5.3, hgt + 0.05, 3.2)
    tgt = (60.0, hgt - 5.0, 40.0)
    v, gb_o, planes, _, _ = _render_both(scene256, oracle, gpu_ctx, eye, tgt, 640, 360)
    _assert_gbuffer_equal(gb_o, planes, "near-plane")


@pytest.mark.parametrize("cam_index", [0, 5])
def test_deferred_rms_within_stated_tolerance(scene256, oracle, gpu_ctx, cam_index):
    eye, tgt = scaled_camera(CAMERAS[cam_index], 256)
    w, h = 640, 360
    v, gb_o, planes, _, _ = _render_both(scene256, oracle, gpu_ctx, eye, tgt, w, h)
    lights = [vr.reference_sun()]
    ref16 = oracle.deferred(v, gb_o, lights, AMBIENT_TOP, AMBIENT_BOTTOM)
    ref32 = oracle.deferred(v, gb_o, lights, AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    for k, arr in (("depth", gb_o.depth), ("diffuse", gb_o.diffuse), ("specular", gb_o.specular),
                   ("normals", gb_o.normals), ("emissive", gb_o.emissive)):
        rt.upload(k, arr)
    hdr = vr.HdrImage(gpu_ctx, w, h)
    vr.DeferredLightingPass(gpu_ctx).Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    out = hdr.download()
    hdr.close()
    rt.close()
    got = oracle.half_to_float(out)
    # stated tolerance (BASELINE.json north_star): per-channel RMS <= 1e-4 vs the fp32 oracle
    for c in range(3):
        rms = float(np.sqrt(np.mean((got[..., c].astype(np.float64) - ref32[..., c]) ** 2)))
        assert rms <= 1e-4, (c, rms)
    # the kernel uses v_rcp/v_rsq and FMA, so the half output may differ from the oracle's
    # by one half-ulp on a small fraction of values; bound both the fraction and the size
    mism = out != ref16
    assert mism.mean() < 0.02, f"{mism.mean():.4f} of the half values differ"
    assert np.abs(got.astype(np.float64) - oracle.half_to_float(ref16)).max() <= 2.0 ** -10 * max(1e-3, float(got.max()))


def test_golden_frame_on_gpu(scene256, oracle, gpu_ctx):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frame_256x144.npz"))
    w, h = 256, 144
    v = vr.View.from_buffer_copy(g["view"].tobytes())
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    scene256["tp"].Render(v, v, rt, vr.default_render_params(400.0))
    for name in ("depth", "diffuse", "specular", "normals", "emissive"):
        assert np.array_equal(rt.download(name), g[name]), name
    hdr = vr.HdrImage(gpu_ctx, w, h)
    vr.DeferredLightingPass(gpu_ctx).Render(v, rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    got = oracle.half_to_float(hdr.download()).astype(np.float64)
    want = oracle.half_to_float(g["hdr"]).astype(np.float64)
    assert np.sqrt(np.mean((got - want) ** 2)) <= 1e-4
    hdr.close()
    rt.close()


def test_lock_view_reuses_previous_selection(scene256, oracle, gpu_ctx):
    ot, tp = scene256["ot"], scene256["tp"]
    w, h = 480, 270
    va = vr.make_view(*scaled_camera(CAMERAS[0], 256), w, h)
    vb = vr.make_view(*scaled_camera(CAMERAS[5], 256), w, h)
    rp = vr.default_render_params(400.0)
    rpl = vr.default_render_params(400.0, lock_view=1)
    gb = oracle.GBufferHost(w, h)
    ot.render(va, gb, rp)
    gb.clear()
    n_o = ot.render(vb, gb, rpl)                 # selection of view A drawn from view B
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    tp.Render(va, va, rt, rp)
    rt.Clear()
    tp.Render(vb, vb, rt, rpl)
    assert tp.num_chunks() == n_o
    _assert_gbuffer_equal(gb, {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}, "lock_view")
    rt.close()


def test_render_over_existing_depth(scene256, oracle, gpu_ctx):
    """Without assume_cleared the pass depth-tests (LessOrEqual) against what the G-buffer already holds
    (the reference draws terrain after the glTF GBufferFill pass, Renderer.cpp:384-415)."""
    ot, tp = scene256["ot"], scene256["tp"]
    w, h = 400, 240
    v = vr.make_view(*scaled_camera(CAMERAS[0], 256), w, h)
    rp = vr.default_render_params(400.0)
    gb = oracle.GBufferHost(w, h)
    ot.render(v, gb, rp)
    # an occluder: the left half is closer than any terrain, a band equals the terrain depth exactly
    pre = oracle.GBufferHost(w, h)
    pre.depth[:, : w // 2] = 0.25
    pre.diffuse[:, : w // 2] = 0x11223344
    yy, xx = np.mgrid[0:h, 0:w]
    band = (gb.depth < 1.0) & (xx >= w // 2) & (yy % 5 == 0)
    assert band.sum() > 100
    pre.depth[band] = gb.depth[band]
    pre.diffuse[band] = 0x55667788
    want = oracle.GBufferHost(w, h)
    for a, b in zip(want.planes(), pre.planes()):
        a[...] = b
    ot.render(v, want, rp)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    for k, arr in (("depth", pre.depth), ("diffuse", pre.diffuse), ("specular", pre.specular), ("normals", pre.normals),
                   ("emissive", pre.emissive)):
        rt.upload(k, arr)
    tp.Render(v, v, rt, rp)
    planes = {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}
    _assert_gbuffer_equal(want, planes, "pre-filled depth")
    assert (planes["diffuse"][:, : w // 2] == 0x11223344).all()
    assert (planes["diffuse"][band] != 0x55667788).all(), "equal depth must pass LessOrEqual"
    rt.close()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_partition_emulated_on_one_gpu_equals_unsplit(scene256, oracle, gpu_ctx, world):
    """run every rank's share on one GPU, concatenate the packed buffers as the
    all-gather would, de-tile, and compare with the unsplit frame byte for byte."""
    from vrenderer_amd.passes import frame_detile, partition_info
    tp = scene256["tp"]
    w, h = 640, 400
    v = vr.make_view(*scaled_camera(CAMERAS[5], 256), w, h)
    rp = vr.default_render_params(400.0, assume_cleared=1)
    lights = [vr.reference_sun()]
    dl = vr.DeferredLightingPass(gpu_ctx)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    full = vr.HdrImage(gpu_ctx, w, h)
    tp.Render(v, v, rt, rp)
    dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, full)
    ref = full.download()
    ref_depth = rt.download("depth")
    info = partition_info(w, h, 0, world)
    gathered = np.zeros(world * info["packed_bytes"] // 2, np.uint16)
    seen = np.zeros((h, w), np.int32)
    for r in range(world):
        part = vr.Partition(r, world)
        rt.Clear()
        packed = vr.HdrImage(gpu_ctx, 128, info["max_owned"] * 128)
        tp.Render(v, v, rt, rp, part)
        dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, packed, part)
        gathered[r * info["packed_bytes"] // 2:(r + 1) * info["packed_bytes"] // 2] = packed.download(info["packed_bytes"])
        d = rt.download("depth")
        seen += (d.view(np.uint32) == ref_depth.view(np.uint32)) & (ref_depth < 1.0)
        packed.close()
    # every covered pixel was rendered by exactly one rank
    assert (seen[ref_depth < 1.0] == 1).all()
    # "all-gather": upload the concatenation, then de-tile on the device
    big = vr.HdrImage(gpu_ctx, 128, world * info["max_owned"] * 128)
    big.upload(gathered)
    out = vr.HdrImage(gpu_ctx, w, h)
    frame_detile(gpu_ctx, big.device_ptr, world, out)
    got = out.download()
    assert np.array_equal(got, ref)
    for o in (big, out, full, rt):
        o.close()


def test_device_srgb_encode_equals_threshold_search(oracle, gpu_ctx):
    """The render-target conversion used by the pixel shader (log/exp estimate + boundary re-check)
    must equal the oracle's threshold search for every input: dense sweep, every threshold +-4 ulp,
    random values, specials."""
    import ctypes as C
    rng = np.random.default_rng(11)
    # thresholds: smallest x that encodes to k, found by bisection on the oracle
    lo = np.zeros(255, np.float32)
    hi = np.ones(255, np.float32)
    for _ in range(40):
        mid = ((lo.astype(np.float64) + hi) / 2).astype(np.float32)
        ge = oracle.linear_to_srgb8(mid) >= np.arange(1, 256)
        hi = np.where(ge, mid, hi)
        lo = np.where(ge, lo, mid)
    near = np.concatenate([(hi.view(np.uint32).astype(np.int64) + d).astype(np.uint32).view(np.float32) for d in range(-4, 5)])
    x = np.concatenate([
        np.arange(0, 1 << 22, dtype=np.float32) / np.float32(1 << 22) * np.float32(1.05),
        near, rng.random(1 << 20, dtype=np.float32), (rng.random(1 << 18, dtype=np.float32) * 0.01).astype(np.float32),
        np.array([0.0, -0.0, 1.0, 1.5, -1.0, np.inf, -np.inf, np.nan, 1e-30, 0.0031308, 0.00313081, 0.0031307], np.float32)])
    got = np.empty(x.size, np.uint8)
    vr.capi.check(gpu_ctx.lib.vr_debug_srgb_encode(gpu_ctx.handle, x.ctypes.data_as(C.c_void_p), x.size, got.ctypes.data_as(C.c_void_p)),
                  "vr_debug_srgb_encode")
    want = oracle.linear_to_srgb8(x)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:5], x[bad[:5]], got[bad[:5]], want[bad[:5]])


def test_short_reciprocal_and_sqrt_equal_ieee(gpu_ctx):
    """The pixel shader's 1/x and sqrt(x) drop the compiler's out-of-range scaling; inside the range they are used for
    (|x| in [2^-60, 2^60)) they must equal 1.0f / x and sqrtf(x) for EVERY float: exhaustive sweep on the device."""
    import ctypes as C
    out = (C.c_ulonglong * 3)()
    vr.capi.check(gpu_ctx.lib.vr_debug_fastmath_check(gpu_ctx.handle, out), "vr_debug_fastmath_check")
    assert out[2] == 120 << 23, out[2]
    assert out[0] == 0 and out[1] == 0, (out[0], out[1])


def _scene_lights(scene, n):
    lights = [vr.reference_sun()] + vr.synthetic_point_lights(n - 1, float(scene["size"]), scene["h"], 400.0, seed=9001)
    for l in lights[1:]:                      # ranges authored for the 2048 world; scale to this scene
        l.angular_size_or_inv_range *= 2048.0 / scene["size"]
    return lights


def test_tiled_deferred_1024_lights_matches_oracle(scene256, oracle, gpu_ctx):
    """BASELINE config 5 in small: 1 sun + 1023 point lights; the culled per-tile sum must equal the
    oracle's all-lights loop (stated tolerance: per-channel RMS <= 1e-4)."""
    w, h = 256, 144
    eye, tgt = scaled_camera(CAMERAS[0], 256)
    v, gb_o, planes, _, _ = _render_both(scene256, oracle, gpu_ctx, eye, tgt, w, h)
    lights = _scene_lights(scene256, 1024)
    ref32 = oracle.deferred(v, gb_o, lights, AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
    lit = oracle.deferred(v, gb_o, lights[:1], AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
    assert np.abs(ref32 - lit).max() > 1e-3, "the point lights must contribute for the test to mean anything"
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    for k, arr in (("depth", gb_o.depth), ("diffuse", gb_o.diffuse), ("specular", gb_o.specular),
                   ("normals", gb_o.normals), ("emissive", gb_o.emissive)):
        rt.upload(k, arr)
    hdr = vr.HdrImage(gpu_ctx, w, h)
    vr.TiledDeferredLightingPass(gpu_ctx).Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    got = oracle.half_to_float(hdr.download()).astype(np.float64)
    for c in range(3):
        rms = float(np.sqrt(np.mean((got[..., c] - ref32[..., c]) ** 2)))
        assert rms <= 1e-4, (c, rms)
    assert np.abs(got[..., :3] - ref32[..., :3]).max() <= 2e-3 * max(1.0, float(ref32.max()))
    # the 16-light streaming kernel and the tiled kernel agree on a 16-light list
    hdr2 = vr.HdrImage(gpu_ctx, w, h)
    vr.DeferredLightingPass(gpu_ctx).Render(v, rt, lights[:16], AMBIENT_TOP, AMBIENT_BOTTOM, hdr2)
    vr.TiledDeferredLightingPass(gpu_ctx).Render(v, rt, lights[:16], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    a16 = oracle.half_to_float(hdr.download()).astype(np.float64)
    b16 = oracle.half_to_float(hdr2.download()).astype(np.float64)
    assert np.sqrt(np.mean((a16 - b16) ** 2)) <= 1e-5
    # partitioned (packed) output of the tiled kernel
    from vrenderer_amd.passes import frame_detile, partition_info
    world = 2
    info = partition_info(w, h, 0, world)
    gathered = np.zeros(world * info["packed_bytes"] // 2, np.uint16)
    vr.TiledDeferredLightingPass(gpu_ctx).Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    full = hdr.download()
    for r in range(world):
        packed = vr.HdrImage(gpu_ctx, 128, info["max_owned"] * 128)
        vr.TiledDeferredLightingPass(gpu_ctx).Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, packed, vr.Partition(r, world))
        gathered[r * info["packed_bytes"] // 2:(r + 1) * info["packed_bytes"] // 2] = packed.download(info["packed_bytes"])
        packed.close()
    big = vr.HdrImage(gpu_ctx, 128, world * info["max_owned"] * 128)
    big.upload(gathered)
    frame_detile(gpu_ctx, big.device_ptr, world, hdr2)
    assert np.array_equal(hdr2.download(), full)
    for o in (big, hdr, hdr2, rt):
        o.close()


def _gpu_gbuffer_as_oracle_input(oracle, gpu_ctx, tp, v, w, h):
    """Terrain G-buffer rendered by the HIP path (bit-exact vs the oracle elsewhere) in the oracle's host layout."""
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    tp.Render(v, v, rt, vr.default_render_params(400.0))
    gb = oracle.GBufferHost(w, h)
    for name, dst in (("depth", gb.depth), ("diffuse", gb.diffuse), ("specular", gb.specular), ("normals", gb.normals),
                      ("emissive", gb.emissive)):
        dst[...] = rt.download(name).reshape(dst.shape)
    return rt, gb


def test_tiled_deferred_config5_at_scale(scene2048, oracle, gpu_ctx):
    """BASELINE config 5 at its own scale: the 2048^2 scene, 1 sun + 1023 point lights of the seed-9001 set with their
    authored ranges (20-80 units), 960x540, flythrough frames.  The tiled pass (per-tile culled lists) must equal the
    oracle's all-lights loop within the stated per-channel RMS <= 1e-4; its packed (partitioned) output must be
    the same pixels."""
    from vrenderer_amd.scene import flythrough_camera
    from vrenderer_amd.passes import frame_detile, partition_info
    w, h = 960, 540
    lights = [vr.reference_sun()] + vr.synthetic_point_lights(1023, 2048.0, scene2048["h"], 400.0, seed=9001)
    tiled = vr.TiledDeferredLightingPass(gpu_ctx)
    for frame in (30, 75):
        v = vr.make_view(*flythrough_camera(frame), w, h)
        rt, gb = _gpu_gbuffer_as_oracle_input(oracle, gpu_ctx, scene2048["tp"], v, w, h)
        ref32 = oracle.deferred(v, gb, lights, AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
        sun_only = oracle.deferred(v, gb, lights[:1], AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
        touched = np.abs(ref32 - sun_only)[..., :3].max(axis=2) > 1e-4
        assert touched.mean() > 0.1, "the point lights must reach a good part of the frame for the test to mean anything"
        hdr = vr.HdrImage(gpu_ctx, w, h)
        tiled.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        tiled.Status()                                   # no tile keeps more than VR_TILE_LIGHT_CAP lights in this scene
        got = oracle.half_to_float(hdr.download()).astype(np.float64)
        # HdrColor is RGBA16F in the reference too: the stated tolerance is on that output (the format's own rounding
        # of this frame, oracle half vs oracle float, is already 1.3e-4 RMS)
        ref16 = oracle.half_to_float(oracle.deferred(v, gb, lights, AMBIENT_TOP, AMBIENT_BOTTOM)).astype(np.float64)
        for c in range(3):
            rms = float(np.sqrt(np.mean((got[..., c] - ref16[..., c]) ** 2)))
            assert rms <= 1e-4, (frame, c, rms)
        # and everywhere at the rounding level of the RGBA16F output (half: 11 significant bits)
        assert np.all(np.abs(got[..., :3] - ref32[..., :3]) <= 2.0 ** -10 * np.abs(ref32[..., :3]) + 1e-6)
        assert 0.3 < float(ref32[..., :3].max()) < 2.0, "the light set is meant to keep the frame in a sane HDR range"
        if frame == 30:
            world = 3
            info = partition_info(w, h, 0, world)
            gathered = np.zeros(world * info["packed_bytes"] // 2, np.uint16)
            full = hdr.download()
            for r in range(world):
                packed = vr.HdrImage(gpu_ctx, 128, info["max_owned"] * 128)
                tiled.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, packed, vr.Partition(r, world))
                gathered[r * info["packed_bytes"] // 2:(r + 1) * info["packed_bytes"] // 2] = packed.download(info["packed_bytes"])
                packed.close()
            big = vr.HdrImage(gpu_ctx, 128, world * info["max_owned"] * 128)
            big.upload(gathered)
            out = vr.HdrImage(gpu_ctx, w, h)
            frame_detile(gpu_ctx, big.device_ptr, world, out)
            assert np.array_equal(out.download(), full)
            big.close(); out.close()
        hdr.close(); rt.close()


def test_tiled_deferred_culling_is_conservative_where_positions_are_ill_conditioned(scene2048, oracle, gpu_ctx):
    """Adversarial case for k_light_cull's boxes (vr_deferred.hip: the pad is a derived fp32 rounding bound).
    Far plane at 10^5 near planes (0.1 / 10000, Renderer.cpp:315), a camera 1.4-2.8 k units from the terrain so that every
    covered pixel has depth >= 0.9995, most of them >= 0.9999 - where one rounding of w moves a reconstructed position by a world unit - and 1,000
    SMALL point lights (range 4-16) hung around the reconstructed positions of pixels with depth >= 0.9999 at 0.2-0.95 of their range:
    whether a light touches a 32x32 tile's box is marginal for thousands of (light, tile) pairs.  Judged per pixel by the
    largest absolute error against the oracle's all-lights loop, not by an RMS a dropped light would drown in."""
    w, h = 960, 540
    eye, tgt = CAMERAS[7]                                   # (900, 500, 900) -> origin
    v = vr.make_view(eye, tgt, w, h)
    rt, gb = _gpu_gbuffer_as_oracle_input(oracle, gpu_ctx, scene2048["tp"], v, w, h)
    covered = gb.depth < 1.0
    assert covered.mean() > 0.3 and float(gb.depth[covered].min()) >= 0.9995, float(gb.depth[covered].min())
    far = covered & (gb.depth >= 0.9999)                     # the lights hang around these
    assert far.mean() > 0.2
    # reconstructed world positions of the covered pixels (window -> clip -> world, float64 is fine for placing lights)
    c2w = np.array(v.clip_to_world, np.float64).reshape(4, 4)
    ys, xs = np.nonzero(far)
    rng = np.random.default_rng(20261004)
    pick = rng.choice(len(xs), 1000, replace=False)
    px, py, dz = xs[pick], ys[pick], gb.depth[ys[pick], xs[pick]].astype(np.float64)
    clip = np.stack([(px + 0.5) * (2.0 / w) - 1.0, 1.0 - (py + 0.5) * (2.0 / h), dz, np.ones_like(dz)], 1)
    wp = clip @ c2w
    wp = wp[:, :3] / wp[:, 3:4]
    lights = [vr.reference_sun()]
    for i in range(1000):
        r = float(rng.uniform(4.0, 16.0))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        pos = wp[i] + d * r * float(rng.uniform(0.2, 0.95))
        col = tuple(float(c) for c in rng.uniform(0.3, 1.0, 3))
        lights.append(vr.point_light(tuple(float(x) for x in pos), 40.0, r, col))
    ref32 = oracle.deferred(v, gb, lights, AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
    sun = oracle.deferred(v, gb, lights[:1], AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
    lit = np.abs(ref32 - sun)[..., :3].max(axis=2)
    assert (lit > 1e-2).sum() > 2000, "the point lights must reach far pixels for the test to mean anything"
    hdr = vr.HdrImage(gpu_ctx, w, h)
    tiled = vr.TiledDeferredLightingPass(gpu_ctx)
    tiled.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    tiled.Status()
    got = oracle.half_to_float(hdr.download()).astype(np.float64)
    err = np.abs(got[..., :3] - ref32[..., :3])
    bound = 2.0 ** -10 * np.abs(ref32[..., :3]) + 2e-6          # the RGBA16F output's own rounding
    bad = np.argwhere(err > bound)
    assert bad.size == 0, f"{len(bad)} pixel channels off by up to {err.max():.3e} (a culled light?), first {bad[:4].tolist()}"
    hdr.close(); rt.close()


def test_tiled_deferred_config5_at_full_size_properties(scene2048, gpu_ctx):
    """BASELINE config 5 at its own size - 7680x4320, 1 sun + 1023 point lights (seed 9001) - through size-independent
    properties: every pixel finite, no tile overflows its list, pixels no point light can reach (farther than its range
    from every one of them, by their reconstructed positions) equal the streaming sun-only pass to one half-precision ulp, and the 8-way
    packed output reassembles to the unsplit frame byte for byte."""
    from vrenderer_amd.scene import flythrough_camera
    from vrenderer_amd.passes import frame_detile, partition_info
    tp = scene2048["tp"]
    W, H = 7680, 4320
    v = vr.make_view(*flythrough_camera(30), W, H)
    rt = vr.RenderTargets(gpu_ctx).Init(W, H)
    tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1))
    lights = [vr.reference_sun()] + vr.synthetic_point_lights(1023, 2048.0, scene2048["h"], 400.0, seed=9001)
    tiled = vr.TiledDeferredLightingPass(gpu_ctx)
    full = vr.HdrImage(gpu_ctx, W, H)
    tiled.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, full)
    tiled.Status()
    ref = full.download()
    img = ref.view(np.float16)[..., :3].astype(np.float32)
    assert np.isfinite(img).all() and img.any()
    sun = vr.HdrImage(gpu_ctx, W, H)
    vr.DeferredLightingPass(gpu_ctx).Render(v, rt, lights[:1], AMBIENT_TOP, AMBIENT_BOTTOM, sun)
    sun_img = sun.download().view(np.float16)[..., :3].astype(np.float32)
    assert (img >= sun_img - 1e-3).all(), "point lights only add light"
    # pixels out of every point light's reach: reconstruct positions on a sub-grid (every 8th pixel) in float64
    depth = rt.download("depth")[::8, ::8].astype(np.float64)
    yy, xx = np.mgrid[0:H:8, 0:W:8]
    c2w = np.array(v.clip_to_world, np.float64).reshape(4, 4)
    clip = np.stack([(xx + 0.5) * (2.0 / W) - 1.0, 1.0 - (yy + 0.5) * (2.0 / H), depth, np.ones_like(depth)], -1)
    wp = clip @ c2w
    wp = wp[..., :3] / wp[..., 3:4]
    lp = np.array([[l.position[0], l.position[1], l.position[2]] for l in lights[1:]], np.float64)
    lr = np.array([1.0 / l.angular_size_or_inv_range for l in lights[1:]], np.float64)
    out_of_reach = np.ones(depth.shape, bool)
    for j in range(len(lr)):                                    # 1023 x 518 k distance tests
        out_of_reach &= ((wp - lp[j]) ** 2).sum(-1) > (lr[j] + 2.0) ** 2      # 2 units of slack for the fp32 reconstruction
    out_of_reach &= depth < 1.0
    assert out_of_reach.mean() > 0.02, "some of the frame must lie beyond every point light"
    # (the two passes reconstruct positions in different arithmetic: equal to one unit in the last place of the RGBA16F output)
    d = np.abs(img[::8, ::8] - sun_img[::8, ::8])[out_of_reach]
    assert (d <= 2.0 ** -10 * sun_img[::8, ::8][out_of_reach] + 1e-7).all(), float(d.max())
    # 8-way packed == unsplit
    world = 8
    info = partition_info(W, H, 0, world)
    gathered = np.zeros(world * info["packed_bytes"] // 2, np.uint16)
    for r in range(world):
        packed = vr.HdrImage(gpu_ctx, 128, info["max_owned"] * 128)
        tiled.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, packed, vr.Partition(r, world))
        gathered[r * info["packed_bytes"] // 2:(r + 1) * info["packed_bytes"] // 2] = packed.download(info["packed_bytes"])
        packed.close()
    tiled.Status()
    big = vr.HdrImage(gpu_ctx, 128, world * info["max_owned"] * 128)
    big.upload(gathered)
    out = vr.HdrImage(gpu_ctx, W, H)
    frame_detile(gpu_ctx, big.device_ptr, world, out)
    assert np.array_equal(out.download(), ref)
    for o in (big, out, full, sun, rt):
        o.close()


def test_tiled_deferred_reports_tile_overflow(scene256, oracle, gpu_ctx):
    """More than VR_TILE_LIGHT_CAP lights over one 32x32 tile: the excess is dropped in light order and
    vr_deferred_tiled_status returns VR_ERR_OVERFLOW once (the flag is cleared); a normal list reports VR_OK again."""
    w, h = 256, 144
    eye, tgt = scaled_camera(CAMERAS[0], 256)
    v = vr.make_view(eye, tgt, w, h)
    rt, gb = _gpu_gbuffer_as_oracle_input(oracle, gpu_ctx, scene256["tp"], v, w, h)
    hdr = vr.HdrImage(gpu_ctx, w, h)
    tiled = vr.TiledDeferredLightingPass(gpu_ctx)
    stacked = [vr.point_light((0.0, 60.0, 0.0), 1.0, 500.0, (1.0, 1.0, 1.0)) for _ in range(1100)]     # every tile sees all of them
    tiled.Render(v, rt, stacked, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    with pytest.raises(vr.VrError) as e:
        tiled.Status()
    assert e.value.code == vr.capi.VR_ERR_OVERFLOW
    # what was kept is exactly the first VR_TILE_LIGHT_CAP lights of the list
    got = oracle.half_to_float(hdr.download()).astype(np.float64)
    ref = oracle.deferred(v, gb, stacked[:1024], AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
    scale = max(1.0, float(ref[..., :3].max()))
    assert np.sqrt(np.mean(((got[..., :3] - ref[..., :3]) / scale) ** 2)) <= 1e-3     # 1024 terms of half-rounded light
    tiled.Status()                                       # cleared by the failing call
    tiled.Render(v, rt, stacked[:1000], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    tiled.Status()
    hdr.close(); rt.close()


def test_tiled_deferred_light_list_changes_between_calls(scene256, oracle, gpu_ctx):
    """The pass uploads a light list only when it differs from the one the device holds: same count with other colours,
    another count, the first list again and a prebuilt vr_light[] (light_array) each light the frame their own way."""
    w, h = 256, 144
    eye, tgt = scaled_camera(CAMERAS[0], 256)
    v = vr.make_view(eye, tgt, w, h)
    rt, gb = _gpu_gbuffer_as_oracle_input(oracle, gpu_ctx, scene256["tp"], v, w, h)
    hdr = vr.HdrImage(gpu_ctx, w, h)
    tiled = vr.TiledDeferredLightingPass(gpu_ctx)
    rng = np.random.default_rng(77)

    def some_lights(n, tint):
        return [vr.reference_sun()] + [vr.point_light((float(rng.uniform(-100, 100)), 40.0, float(rng.uniform(-100, 100))), 30.0, 60.0, tint)
                                       for _ in range(n - 1)]
    a = some_lights(40, (1.0, 0.2, 0.2))
    b = [vr.reference_sun()] + [vr.point_light(tuple(l.position), 30.0, 60.0, (0.2, 0.2, 1.0)) for l in a[1:]]   # same places, other colour
    c = a[:17]
    frames = []
    for lights in (a, b, c, a, vr.light_array(b)):
        tiled.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        frames.append(hdr.download().copy())
    tiled.Status()
    assert np.array_equal(frames[0], frames[3]) and np.array_equal(frames[1], frames[4])
    assert not np.array_equal(frames[0], frames[1]) and not np.array_equal(frames[0], frames[2])
    for got_bits, lights in ((frames[1], b), (frames[2], c)):
        got = oracle.half_to_float(got_bits).astype(np.float64)
        ref = oracle.deferred(v, gb, lights, AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
        assert np.sqrt(np.mean((got[..., :3] - ref[..., :3]) ** 2)) <= 1e-4
    hdr.close(); rt.close()


def test_cpp_allgather_example_through_rccl(product_lib, tmp_path):
    """SURVEY 8b's vr_frame_allgather through the C ABI with a real RCCL communicator (one rank per visible device;
    world size 1 on a one-GPU box): histogram all-reduce, all-gather of RGB8 tiles, de-tile; the assembled frame equals
    the unsplit one byte for byte."""
    import subprocess
    from tests.test_abi_cpu import _build_allgather_example
    exe = _build_allgather_example(tmp_path)
    r = subprocess.run([exe, "--require-gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "differing bytes=0" in r.stdout, r.stdout


def test_cpp_host_example(product_lib, tmp_path):
    """The C++ caller of tests/host/frame_example.cpp renders and lights a frame through the C ABI."""
    import subprocess
    from tests.test_abi_cpu import _build_host_example
    exe = _build_host_example(tmp_path)
    r = subprocess.run([exe, "--require-gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "chunks=" in r.stdout


@pytest.mark.parametrize("scene_name", ["scene256", "scene2048"])
def test_node_heights_and_height_aware_select(scene_name, request, oracle, gpu_ctx):
    """Row f2: QuadTree::SetHeight as a device reduction (bit-exact per node) and NodeSelect with
    m_HeightLoaded = true (tight y-bounds), plus the instance transforms that then carry y."""
    sc = request.getfixturevalue(scene_name)
    ot, tp, size = sc["ot"], sc["tp"], sc["size"]
    try:
        ot.set_height(True)
        tp.SetHeight(True)
        want = ot.node_heights()
        got = tp.node_heights(0, ot.num_nodes)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert want[0, 1] > 0, "root extents.y must be non-zero on a non-flat heightmap"
        fewer = 0
        for cam in CAMERAS:
            eye, tgt = scaled_camera(cam, size)
            v = vr.make_view(eye, tgt, 1920, 1080)
            n_o, ids_o, inst_o = ot.select(v, 400.0)
            n_g, ids_g, inst_g = tp.NodeSelect(v, 400.0)
            assert n_g == n_o and np.array_equal(ids_g, ids_o) and np.array_equal(inst_g, inst_o), cam
            ot.set_height(False)
            n_loose, _, _ = ot.select(v, 400.0)
            ot.set_height(True)
            fewer += n_loose != n_o
        assert fewer > 0, "tight bounds should change the selection for at least one camera"
        if size == 256:
            eye, tgt = scaled_camera(CAMERAS[5], size)
            v, gb_o, planes, n_o, n_g = _render_both(sc, oracle, gpu_ctx, eye, tgt, 480, 270)
            assert n_o == n_g
            _assert_gbuffer_equal(gb_o, planes, "height-aware frame")
    finally:
        ot.set_height(False)
        tp.SetHeight(False)


def test_multi_surface_world(oracle, gpu_ctx):
    """Row f4: WORLD_SIZE / SURFACE_SIZE = 2 -> four quadtrees sharing one instance buffer
    (TerrainPass.cpp:97-110,175-187), one heightmap over the whole world."""
    world, surface = 512, 256
    p = params(surface)
    p.world_size = float(world)
    h = oracle.synth_heightmap(world)
    a = oracle.synth_albedo(world, h)
    ot = oracle.OracleTerrain(p, h, a)
    tp = vr.TerrainPass(gpu_ctx, p).Init(h, a)
    try:
        assert ot.num_nodes == 4 * 87381 and tp.GetNumLods() == ot.num_lods == 8
        multi = 0
        for cam in CAMERAS:
            eye, tgt = scaled_camera(cam, world)
            v = vr.make_view(eye, tgt, 1280, 720)
            n_o, ids_o, inst_o = ot.select(v, 400.0)
            n_g, ids_g, inst_g = tp.NodeSelect(v, 400.0)
            assert n_g == n_o and np.array_equal(ids_g, ids_o) and np.array_equal(inst_g, inst_o), cam
            multi += len(set((ids_o // 87381).tolist())) > 1
        assert multi > 0, "at least one camera must select nodes from several surfaces"
        ot.set_height(True)
        tp.SetHeight(True)
        assert np.array_equal(tp.node_heights(0, ot.num_nodes).view(np.uint32), ot.node_heights().view(np.uint32))
        ot.set_height(False)
        tp.SetHeight(False)
        sc = dict(ot=ot, tp=tp)
        eye, tgt = scaled_camera(CAMERAS[1], world)
        v, gb_o, planes, n_o, n_g = _render_both(sc, oracle, gpu_ctx, eye, tgt, 640, 360)
        assert n_o == n_g
        _assert_gbuffer_equal(gb_o, planes, "multi-surface frame")
    finally:
        tp.close()
        ot.close()


def test_ragged_sizes_and_mismatched_textures(oracle, gpu_ctx):
    """Odd frame size (scalar store / scalar deferred paths), non-power-of-two world (the division
    form of the uv mapping), heightmap and albedo of different, non-square sizes (separate LODs)."""
    surface = 200
    p = params(surface)
    rng = np.random.default_rng(5)
    big = oracle.synth_heightmap(512)
    h = np.ascontiguousarray(big[:300, :420])                       # 420 x 300 heightmap
    a = np.ascontiguousarray(oracle.synth_albedo(512, big)[:96, :160])   # 160 x 96 albedo
    ot = oracle.OracleTerrain(p, h, a)
    tp = vr.TerrainPass(gpu_ctx, p).Init(h, a)
    try:
        assert tp.GetNumLods() == ot.num_lods == 7
        for l in range(ot.height_levels()):
            assert np.array_equal(tp.download_mip("height", l), ot.height_mip(l))
        for l in range(ot.albedo_levels()):
            assert np.array_equal(tp.download_mip("albedo", l), ot.albedo_mip(l))
        sc = dict(ot=ot, tp=tp)
        for cam in (CAMERAS[0], CAMERAS[5], CAMERAS[6]):
            eye, tgt = scaled_camera(cam, surface)
            v, gb_o, planes, n_o, n_g = _render_both(sc, oracle, gpu_ctx, eye, tgt, 333, 177)
            assert n_o == n_g and n_o > 0
            _assert_gbuffer_equal(gb_o, planes, "ragged")
            ref32 = oracle.deferred(v, gb_o, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
            rt = vr.RenderTargets(gpu_ctx).Init(333, 177)
            for k, arr in (("depth", gb_o.depth), ("diffuse", gb_o.diffuse), ("specular", gb_o.specular),
                           ("normals", gb_o.normals), ("emissive", gb_o.emissive)):
                rt.upload(k, arr)
            hdr = vr.HdrImage(gpu_ctx, 333, 177)
            vr.DeferredLightingPass(gpu_ctx).Render(v, rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
            got = oracle.half_to_float(hdr.download()).astype(np.float64)
            assert np.sqrt(np.mean((got - ref32) ** 2)) <= 1e-4
            hdr.close()
            rt.close()
    finally:
        tp.close()
        ot.close()


def test_scratch_grows_by_high_water_mark_and_reports_sticky_conditions(scene2048, oracle, gpu_ctx, monkeypatch):
    """The per-frame scratch holds cap nodes, not max_instances: it doubles BEFORE a frame can outgrow it (count > cap / 2, read
    from the pinned counters of completed chains, no wait), and a frame that does outgrow it is drawn without the excess, grows
    the scratch and is reported - once, sticky - by the next vr_terrain_render; the frame rendered after that is complete.
    NodeSelect's list never depends on the scratch.  VR_OPT_SCRATCH_WORST_CASE sizes for max_instances up front."""
    import ctypes as C
    h, a, ot = scene2048["h"], scene2048["a"], scene2048["ot"]
    w, hh = 960, 540
    v = vr.make_view(*CAMERAS[0], w, hh)
    rp = vr.default_render_params(400.0, assume_cleared=1)
    want = oracle.GBufferHost(w, hh)
    n_o = ot.render(v, want, vr.default_render_params(400.0))
    assert n_o > 200

    def planes(rt):
        return {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}
    # 1. far too small: truncated frame -> sticky VR_ERR_OVERFLOW on the next render -> complete frame
    monkeypatch.setenv("VR_SCRATCH_INITIAL_NODES", "64")
    tp = vr.TerrainPass(gpu_ctx, params(2048)).Init(h, a)
    rt = vr.RenderTargets(gpu_ctx).Init(w, hh)
    try:
        small = tp.memory_bytes()["scratch"]
        n_g, ids_g, _ = tp.NodeSelect(v, 400.0)              # the selection itself is complete whatever the scratch holds
        assert n_g == n_o
        small = min(small, tp.memory_bytes()["scratch"])
        tp2 = vr.TerrainPass(gpu_ctx, params(2048)).Init(h, a)     # (NodeSelect above has grown tp's scratch already: a fresh one)
        try:
            tp2.Render(v, v, rt, rp)                           # 64 of the nodes drawn
            gpu_ctx.synchronize()
            assert not np.array_equal(rt.download("depth").view(np.uint32), want.depth.view(np.uint32))
            rc = gpu_ctx.lib.vr_terrain_render(tp2.handle, C.byref(v), C.byref(v), rt.handle, C.byref(rp), None)
            assert rc == vr.capi.VR_ERR_OVERFLOW and b"scratch" in gpu_ctx.lib.vr_last_error()
            _assert_gbuffer_equal(want, planes(rt), "the frame queued by the call that reported the earlier one")
            assert gpu_ctx.lib.vr_terrain_render(tp2.handle, C.byref(v), C.byref(v), rt.handle, C.byref(rp), None) == vr.capi.VR_OK   # reported once
            assert tp2.memory_bytes()["scratch"] > 3 * small          # 64 -> 1024 nodes (the bins, clipper lists and selection arrays do not scale with the node count)
        finally:
            tp2.close()
    finally:
        tp.close()
    # 2. big enough for the frame but more than half full: grows silently, every frame complete
    monkeypatch.setenv("VR_SCRATCH_INITIAL_NODES", str(n_o + 8))
    tp = vr.TerrainPass(gpu_ctx, params(2048)).Init(h, a)
    try:
        before = tp.memory_bytes()["scratch"]
        for k in range(4):
            tp.Render(v, v, rt, rp)
            _assert_gbuffer_equal(want, planes(rt), f"frame {k}")
        assert tp.memory_bytes()["scratch"] > 1.5 * before and tp.num_chunks() == n_o
    finally:
        tp.close()
    monkeypatch.delenv("VR_SCRATCH_INITIAL_NODES")
    # 2b. the bins (one entry per triangle and raster tile) grow the same way: too few -> dropped triangles, reported once, then complete
    monkeypatch.setenv("VR_SCRATCH_INITIAL_BINS", "20000")
    tp = vr.TerrainPass(gpu_ctx, params(2048)).Init(h, a)
    monkeypatch.delenv("VR_SCRATCH_INITIAL_BINS")
    try:
        tp.Render(v, v, rt, rp)
        gpu_ctx.synchronize()
        assert not np.array_equal(rt.download("depth").view(np.uint32), want.depth.view(np.uint32))
        rc = gpu_ctx.lib.vr_terrain_render(tp.handle, C.byref(v), C.byref(v), rt.handle, C.byref(rp), None)
        assert rc == vr.capi.VR_ERR_OVERFLOW
        _assert_gbuffer_equal(want, planes(rt), "bins grown: the frame queued by the reporting call")
        assert gpu_ctx.lib.vr_terrain_render(tp.handle, C.byref(v), C.byref(v), rt.handle, C.byref(rp), None) == vr.capi.VR_OK
    finally:
        tp.close()
    # 3. defaults: 1024 nodes; worst case on request
    tp = vr.TerrainPass(gpu_ctx, params(2048)).Init(h, a)
    default_bytes = tp.memory_bytes()["scratch"]
    tp.close()
    gpu_ctx.set_scratch_worst_case(True)
    try:
        tp = vr.TerrainPass(gpu_ctx, params(2048)).Init(h, a)
        worst = tp.memory_bytes()["scratch"]
        tp.close()
    finally:
        gpu_ctx.set_scratch_worst_case(False)
    assert default_bytes < 1.6e9 < 4.0e9 < worst
    rt.close()


def test_frame_submit_passes_a_sticky_condition_on_and_still_queues_its_frame(scene2048, oracle, gpu_ctx, monkeypatch):
    """vr_frame_submit with a scratch that is far too small: the first frame is drawn without the excess; the second call returns
    VR_ERR_OVERFLOW once - behind its own launches: its G-buffer and HdrColor are complete - and the third is clean."""
    import ctypes as C
    h, a, ot = scene2048["h"], scene2048["a"], scene2048["ot"]
    w, hh = 640, 360
    v = vr.make_view(*CAMERAS[0], w, hh)
    want = oracle.GBufferHost(w, hh)
    assert ot.render(v, want, vr.default_render_params(400.0)) > 100
    monkeypatch.setenv("VR_SCRATCH_INITIAL_NODES", "32")
    tp = vr.TerrainPass(gpu_ctx, params(2048)).Init(h, a)
    monkeypatch.delenv("VR_SCRATCH_INITIAL_NODES")
    rt = vr.RenderTargets(gpu_ctx).Init(w, hh)
    hdr, hdr_ref = vr.HdrImage(gpu_ctx, w, hh), vr.HdrImage(gpu_ctx, w, hh)
    sun = [vr.reference_sun()]
    fr = vr.Frame(tp, rt, vr.default_render_params(400.0, assume_cleared=1), sun, AMBIENT_TOP, AMBIENT_BOTTOM)
    try:
        fr.submit(v, hdr)                                        # truncated, not reported yet
        gpu_ctx.synchronize()
        d = fr.desc
        rc = gpu_ctx.lib.vr_frame_submit(tp.handle, rt.handle, C.byref(d))
        assert rc == vr.capi.VR_ERR_OVERFLOW, rc
        _assert_gbuffer_equal(want, {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}, "the reporting call's own frame")
        vr.DeferredLightingPass(gpu_ctx).Render(v, rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr_ref)
        assert np.array_equal(hdr.download().view(np.uint16), hdr_ref.download().view(np.uint16))
        assert gpu_ctx.lib.vr_frame_submit(tp.handle, rt.handle, C.byref(d)) == vr.capi.VR_OK
    finally:
        hdr.close(); hdr_ref.close(); rt.close(); tp.close()


def test_too_many_instances_and_empty_selection(scene256, oracle, gpu_ctx):
    """MAX_INSTANCES overflow is an assert in the reference (TerrainPass.cpp:238): here an error code,
    with the first max_instances nodes still in order.  A camera that sees nothing selects nothing."""
    h, a = scene256["h"], scene256["a"]
    p = params(256, max_instances=16)
    tp = vr.TerrainPass(gpu_ctx, p).Init(h, a)
    ot = scene256["ot"]
    try:
        v = vr.make_view(*scaled_camera(CAMERAS[0], 256), 640, 360)
        n_o, ids_o, _ = ot.select(v, 400.0)
        assert n_o > 16
        with pytest.raises(vr.VrError) as e:
            tp.NodeSelect(v, 400.0)
        assert e.value.code == vr.capi.VR_ERR_TOO_MANY_INSTANCES
        ids = np.zeros(16, np.uint32)
        n = np.zeros(1, np.uint32)
        import ctypes as C
        rc = gpu_ctx.lib.vr_terrain_select(tp.handle, C.byref(v), 400.0, ids.ctypes.data_as(C.c_void_p), None,
                                           n.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert rc == vr.capi.VR_ERR_TOO_MANY_INSTANCES and n[0] == 16 and np.array_equal(ids, ids_o[:16])
    finally:
        tp.close()
    # camera far above and away, looking up: nothing is in range of the root
    tp = scene256["tp"]
    v = vr.make_view((5000.0, 9000.0, 5000.0), (5000.0, 9900.0, 5100.0), 320, 200)
    n_o, _, _ = ot.select(v, 400.0)
    n_g, ids_g, _ = tp.NodeSelect(v, 400.0)
    assert n_o == 0 and n_g == 0
    rt = vr.RenderTargets(gpu_ctx).Init(320, 200)
    tp.Render(v, v, rt, vr.default_render_params(400.0))
    assert (rt.download("depth") == 1.0).all() and not rt.download("diffuse").any()
    rt.close()


def test_full_size_8k_flythrough_is_the_same_with_and_without_plane_tracking(scene2048, gpu_ctx):
    """The bench's own frames (7680x4320, flythrough views 0, 1, 2, 40, 41 - sky above the horizon, terrain below) rendered and lit
    with VR_OPT_PLANE_TRACKING on (regions skipped, planes not re-read, lazy clear) and off: every G-buffer plane and HdrColor
    byte for byte, frame after frame on the same targets; and a fair share of the regions really was skipped."""
    from vrenderer_amd.scene import flythrough_camera
    tp = scene2048["tp"]
    W, H = 7680, 4320
    lights = [vr.reference_sun()]
    dl = vr.DeferredLightingPass(gpu_ctx)
    rp = vr.default_render_params(400.0, assume_cleared=1)
    frames = (0, 1, 2, 40, 41)
    digests = {}
    for tracking in (False, True):
        gpu_ctx.set_plane_tracking(tracking)
        rt = vr.RenderTargets(gpu_ctx).Init(W, H)
        hdr = vr.HdrImage(gpu_ctx, W, H)
        try:
            for i in frames:
                v = vr.make_view(*flythrough_camera(i), W, H)
                if i == 40:
                    rt.Clear()                                   # (lazy under the tracking) + a keep-what-is-there pass
                    tp.Render(v, v, rt, vr.default_render_params(400.0))
                else:
                    tp.Render(v, v, rt, rp)
                dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
                got = [rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")] + [hdr.download()]
                if tracking:
                    for name, a, b in zip(("depth", "diffuse", "specular", "normals", "emissive", "HdrColor"), digests[i], got):
                        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"frame {i}: {name} differs with the tracking on"
                else:
                    digests[i] = got
            if tracking:
                c = rt.region_census()
                assert c["clear"] > 0.1 * c["total"] and c["specular_constant"] > 0.5 * c["total"], c
        finally:
            gpu_ctx.set_plane_tracking(True)
            hdr.close(); rt.close()


def test_full_size_8k_properties(scene2048, gpu_ctx):
    """BASELINE's full size (7680x4320, heightmap 2048^2) through size-independent properties: the
    frame is deterministic, Clear+Render equals the fused-clear render, the default camera leaves no
    hole (any pixel at the clear depth would be a crack), every normal is unit length to 16-bit
    precision, and the 8-way screen-tile split reassembles to the unsplit frame byte for byte."""
    from vrenderer_amd.passes import frame_detile, partition_info
    tp = scene2048["tp"]
    W, H = 7680, 4320
    v = vr.make_view(CAMERAS[0][0], CAMERAS[0][1], W, H)
    rt = vr.RenderTargets(gpu_ctx).Init(W, H)
    rp = vr.default_render_params(400.0)
    rt.Clear()
    tp.Render(v, v, rt, rp)
    depth = rt.download("depth")
    diffuse = rt.download("diffuse")
    normals = rt.download("normals")
    assert (depth < 1.0).all(), "hole in an all-terrain view"
    n = normals.view(np.int16)[..., :3].astype(np.float64) / 32767.0
    assert np.abs(np.sqrt((n * n).sum(-1)) - 1.0).max() < 1e-4
    tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1))
    assert np.array_equal(rt.download("depth").view(np.uint32), depth.view(np.uint32))
    assert np.array_equal(rt.download("diffuse"), diffuse) and np.array_equal(rt.download("normals"), normals)
    lights = [vr.reference_sun()]
    dl = vr.DeferredLightingPass(gpu_ctx)
    full = vr.HdrImage(gpu_ctx, W, H)
    dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, full)
    ref = full.download()
    assert np.isfinite(ref.view(np.float16)).all() and ref[..., :3].any()
    world = 8
    info = partition_info(W, H, 0, world)
    gathered = np.zeros(world * info["packed_bytes"] // 2, np.uint16)
    rpa = vr.default_render_params(400.0, assume_cleared=1)
    for r in range(world):
        part = vr.Partition(r, world)
        packed = vr.HdrImage(gpu_ctx, 128, info["max_owned"] * 128)
        tp.Render(v, v, rt, rpa, part)
        dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, packed, part)
        gathered[r * info["packed_bytes"] // 2:(r + 1) * info["packed_bytes"] // 2] = packed.download(info["packed_bytes"])
        packed.close()
    big = vr.HdrImage(gpu_ctx, 128, world * info["max_owned"] * 128)
    big.upload(gathered)
    out = vr.HdrImage(gpu_ctx, W, H)
    frame_detile(gpu_ctx, big.device_ptr, world, out)
    assert np.array_equal(out.download(), ref)
    for o in (big, out, full, rt):
        o.close()


def test_full_size_8k_frame_against_the_oracle(scene2048, oracle, gpu_ctx):
    """One whole frame at BASELINE's full size (7680x4320, flythrough frame 30) against the CPU oracle: all 33.2 M
    pixels of every G-buffer plane bit-exact, HDR per-channel RMS <= 1e-4 (the oracle needs ~20 s for it)."""
    from vrenderer_amd.scene import flythrough_camera
    W, H = 7680, 4320
    eye, tgt = flythrough_camera(30)
    v = vr.make_view(eye, tgt, W, H)
    rp = vr.default_render_params(400.0, assume_cleared=1)
    rt = vr.RenderTargets(gpu_ctx).Init(W, H)
    scene2048["tp"].Render(v, v, rt, rp)
    hdr = vr.HdrImage(gpu_ctx, W, H)
    vr.DeferredLightingPass(gpu_ctx).Render(v, rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    gb = oracle.GBufferHost(W, H)
    n_o = scene2048["ot"].render(v, gb, rp)
    assert n_o == scene2048["tp"].num_chunks()
    for name, ref in (("depth", gb.depth.view(np.uint32)), ("diffuse", gb.diffuse), ("specular", gb.specular),
                      ("normals", gb.normals), ("emissive", gb.emissive)):
        got = rt.download(name)
        if name == "depth":
            got = got.view(np.uint32)
        assert np.array_equal(got, ref), f"8K {name}: {int((got != ref).sum())} entries differ"
        del got
    covered = int((gb.depth < 1.0).sum())
    assert covered > 0.5 * W * H
    ref = oracle.deferred(v, gb, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
    got = hdr.download().view(np.float16)
    se = np.zeros(3)
    for y0 in range(0, H, 540):                       # in slabs: keeps the float64 temporaries small
        d = got[y0:y0 + 540, :, :3].astype(np.float64) - ref[y0:y0 + 540, :, :3].astype(np.float64)
        se += (d * d).sum(axis=(0, 1))
    rms = np.sqrt(se / (W * H))
    assert (rms <= 1e-4).all(), rms
    hdr.close(); rt.close()


def test_odd_sized_frame_on_64_pixel_tiles_against_the_oracle(scene2048, oracle, gpu_ctx):
    """5001 x 3343 (neither a multiple of 4; 79 x 53 tiles of 64 pixels, the last column 9 and the last row 15 pixels
    wide): the tile pass's column-of-four resolve and its per-row stores at the frame's edges.  Every G-buffer plane
    bit-exact vs the oracle, once fused-clear and once over a cleared target whose depth plane already holds a near
    wall in the left half (what the wall hides must keep its cleared planes)."""
    from vrenderer_amd.scene import flythrough_camera
    W, H = 5001, 3343
    tp = scene2048["tp"]
    v = vr.make_view(*flythrough_camera(100), W, H)
    rt = vr.RenderTargets(gpu_ctx).Init(W, H)
    gb = oracle.GBufferHost(W, H)
    for fused in (True, False):
        rp = vr.default_render_params(400.0, assume_cleared=1 if fused else 0)
        rt.Clear()
        gb.clear()
        if not fused:
            wall = np.ones((H, W), np.float32)
            wall[:, :W // 2] = 0.9990
            rt.upload("depth", wall)
            gb.depth[...] = wall
        tp.Render(v, v, rt, rp)
        scene2048["ot"].render(v, gb, rp)
        for name, ref in (("depth", gb.depth.view(np.uint32)), ("diffuse", gb.diffuse), ("specular", gb.specular),
                          ("normals", gb.normals), ("emissive", gb.emissive)):
            got = rt.download(name)
            if name == "depth":
                got = got.view(np.uint32)
            assert np.array_equal(got, ref), f"{name} (fused={fused}): {int((got != ref).sum())} entries differ"
        assert (gb.depth < 1.0).mean() > 0.5
    rt.close()


def test_full_size_4k_frame_against_the_oracle(scene2048, oracle, gpu_ctx):
    """BASELINE config 3's resolution (3840x2160, 32-pixel raster tiles, full-quadtree terrain, flythrough frame 75):
    every G-buffer plane of the whole frame bit-exact vs the oracle, HDR per-channel RMS <= 1e-4, Clear+Render equals
    the fused-clear render, and the 4-way screen-tile split (tone-mapped RGB8 exchange format) reassembles to the
    unsplit LDR frame byte for byte."""
    from vrenderer_amd.scene import flythrough_camera
    from vrenderer_amd.passes import frame_detile_ldr, partition_info
    W, H = 3840, 2160
    tp = scene2048["tp"]
    v = vr.make_view(*flythrough_camera(75), W, H)
    rt = vr.RenderTargets(gpu_ctx).Init(W, H)
    rt.Clear()
    tp.Render(v, v, rt, vr.default_render_params(400.0))
    gb = oracle.GBufferHost(W, H)
    n_o = scene2048["ot"].render(v, gb, vr.default_render_params(400.0))
    assert n_o == tp.num_chunks()
    planes = {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}
    _assert_gbuffer_equal(gb, planes, "4K frame")
    rpa = vr.default_render_params(400.0, assume_cleared=1)
    tp.Render(v, v, rt, rpa)
    for k in planes:
        assert np.array_equal(rt.download(k).view(np.uint8), planes[k].view(np.uint8)), f"fused clear: {k}"
    lights = [vr.reference_sun()]
    dl = vr.DeferredLightingPass(gpu_ctx)
    hdr = vr.HdrImage(gpu_ctx, W, H)
    dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    ref = oracle.deferred(v, gb, lights, AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
    got = hdr.download().view(np.float16)[..., :3].astype(np.float64)
    rms = np.sqrt(((got - ref[..., :3]) ** 2).mean(axis=(0, 1)))
    assert (rms <= 1e-4).all(), rms
    # unsplit LDR frame, then the same through 4 ranks' packed RGB8 tiles
    tmp = vr.default_tonemap_params()
    tm = vr.ToneMappingPass(gpu_ctx); tm.AdvanceFrame(1.0 / 60.0)
    ldr = vr.LdrImage(gpu_ctx, W, H)
    tm.SimpleRender(tmp, hdr, ldr)
    want = ldr.download()
    assert np.array_equal(want, oracle.ToneMapper().SimpleRender(tmp, hdr.download()))
    world = 4
    info = partition_info(W, H, 0, world)
    nb = info["packed_bytes_ldr"]
    gathered = np.zeros(world * nb, np.uint8)
    tms = vr.ToneMappingPass(gpu_ctx); tms.AdvanceFrame(1.0 / 60.0)
    tms.ResetHistogram()
    packed = []
    for r in range(world):                                         # every rank's pixels enter the one histogram
        part = vr.Partition(r, world)
        ph = vr.HdrImage(gpu_ctx, 128, info["max_owned"] * 128)
        tp.Render(v, v, rt, rpa, part)
        dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, ph, part)
        tms.AddFrameToHistogram(tmp, ph, W, H, part)
        packed.append(ph)
    tms.ComputeExposure(tmp)
    for r in range(world):
        pl = vr.LdrImage(gpu_ctx, W, H, capacity_bytes=nb)
        tms.Render(tmp, packed[r], pl, W, H, vr.Partition(r, world))
        gathered[r * nb:(r + 1) * nb] = pl.download(nb)
        pl.close(); packed[r].close()
    gd = vr.LdrImage(gpu_ctx, W, H, capacity_bytes=world * nb)
    gd.upload(gathered)
    out = vr.LdrImage(gpu_ctx, W, H)
    frame_detile_ldr(gpu_ctx, gd.device_ptr, world, W, H, out)
    assert np.array_equal(out.download(), want)
    for o in (gd, out, ldr, tm, tms, hdr, rt):
        o.close()


def test_prepared_geometry_is_equivalent(scene256, oracle, gpu_ctx):
    """vr_terrain_prepare only moves work in time: a prepared frame, a frame whose prepared geometry does
    not match (discarded) and a frame rendered on one stream all equal the oracle."""
    ot, tp = scene256["ot"], scene256["tp"]
    w, h = 512, 288
    va = vr.make_view(*scaled_camera(CAMERAS[0], 256), w, h)
    vb = vr.make_view(*scaled_camera(CAMERAS[5], 256), w, h)
    rp = vr.default_render_params(400.0)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    want = {}
    for name, v in (("a", va), ("b", vb)):
        gb = oracle.GBufferHost(w, h)
        ot.render(v, gb, rp)
        want[name] = gb

    def check(v, key, what):
        planes = {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}
        _assert_gbuffer_equal(want[key], planes, what)

    rt.Clear(); tp.Prepare(va, rt, rp); tp.Render(va, va, rt, rp); check(va, "a", "prepared")
    rt.Clear(); tp.Prepare(va, rt, rp); tp.Render(vb, vb, rt, rp); check(vb, "b", "stale prepare discarded")
    rt.Clear(); tp.Prepare(vb, rt, rp); tp.Prepare(va, rt, rp); tp.Render(va, va, rt, rp); check(va, "a", "re-prepared")
    # a chain of frames, each preparing the next one (the bench's pattern)
    seq = [va, vb, va, vb, vb, va]
    for i, v in enumerate(seq):
        rt.Clear()
        tp.Render(v, v, rt, rp)
        if i + 1 < len(seq):
            tp.Prepare(seq[i + 1], rt, rp)
        check(v, "a" if v is va else "b", f"chain {i}")
    gpu_ctx.set_async_geometry(False)
    try:
        rt.Clear(); tp.Render(vb, vb, rt, rp); check(vb, "b", "single stream")
    finally:
        gpu_ctx.set_async_geometry(True)
    rt.close()


def test_context_stream_changed_between_prepare_and_render(scene2048, oracle, gpu_ctx):
    """vr_context_set_stream between vr_terrain_prepare and vr_terrain_render (per-stage streams): the wait
    for the prepared chain that prepare queued sits on the OLD stream; the tile pass on the new one must wait for the chain
    itself, or it reads vertices, bins and records that the geometry kernels are still writing."""
    import torch
    ot, tp = scene2048["ot"], scene2048["tp"]
    w, h = 960, 540
    rp = vr.default_render_params(400.0, assume_cleared=1)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    gpu_ctx.synchronize()                                 # (the new target's clear ran on the stream the context had until now)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    try:
        for k, cam in enumerate((CAMERAS[0], CAMERAS[3], CAMERAS[5])):
            v = vr.make_view(*cam, w, h)
            gb = oracle.GBufferHost(w, h)
            gb.clear(); ot.render(v, gb, rp)
            first, second = (sa, sb) if k % 2 == 0 else (sb, sa)
            gpu_ctx.set_stream(first.cuda_stream)
            tp.Prepare(v, rt, rp)                         # geometry on the terrain's stream; the context's stream (first) waits for it
            second.wait_stream(first)                     # what the host owes: its own work on the old stream is ordered before the new one
            gpu_ctx.set_stream(second.cuda_stream)
            tp.Render(v, v, rt, rp)                       # ... but the chain itself is the library's to wait for
            planes = {p: rt.download(p) for p in ("depth", "diffuse", "specular", "normals", "emissive")}
            _assert_gbuffer_equal(gb, planes, f"stream switched between prepare and render, camera {k}")
    finally:
        torch.cuda.synchronize()
        gpu_ctx.set_stream(0)
        rt.close()


def test_region_tracking_never_changes_the_gbuffer(scene256, oracle, gpu_ctx):
    """VR_OPT_PLANE_TRACKING per region (8 rows x 32 pixels): a sky region known to hold the clear values is not written by a
    pass over a 'cleared' target, a terrain region known to hold the specular constant keeps that plane.  Through a history of
    moving cameras (sky <-> terrain), clears, foreign writes, keep-what-is-there passes, other variants of the tile pass
    (wireframe, depth only, 64-pixel tiles, a partition) the five planes equal the oracle's after every step, with the tracking
    on and off; the census shows that regions really were skipped."""
    ot, tp = scene256["ot"], scene256["tp"]
    w, h = 544, 300                                            # partial tiles on both edges
    cams = [scaled_camera(c, 256) for c in (CAMERAS[0], CAMERAS[5], CAMERAS[7], CAMERAS[1])]
    cams.append(((10.0, 60.0, 10.0), (60.0, 200.0, 60.0)))     # looking up: sky only
    views = [vr.make_view(e, t, w, h) for e, t in cams]
    rp_c, rp_k = vr.default_render_params(400.0, assume_cleared=1), vr.default_render_params(400.0)
    rng = np.random.default_rng(5)
    junk32 = rng.integers(1, 2 ** 32 - 1, (h, w), dtype=np.uint32)
    names = ("depth", "diffuse", "specular", "normals", "emissive")

    def planes(rt):
        return {k: rt.download(k) for k in names}
    saw_clear = saw_spec = False
    for tracking in (True, False):
        gpu_ctx.set_plane_tracking(tracking)
        rt = vr.RenderTargets(gpu_ctx).Init(w, h)
        cur = [oracle.GBufferHost(w, h)]

        def over_cleared(i, part=None, what=""):
            cur[0] = oracle.GBufferHost(w, h) if part is None else cur[0]
            if part is None:
                ot.render(views[i], cur[0], rp_k)
            tp.Render(views[i], views[i], rt, rp_c, part)
            if part is None:
                _assert_gbuffer_equal(cur[0], planes(rt), f"{what} view {i} over a 'cleared' target (tracking {tracking})")

        def keep(i, what=""):
            ot.render(views[i], cur[0], rp_k)
            tp.Render(views[i], views[i], rt, rp_k)
            _assert_gbuffer_equal(cur[0], planes(rt), f"{what} view {i} over what is there (tracking {tracking})")
        try:
            c = rt.region_census()
            assert c["total"] == 4 * ((w + 31) // 32) * ((h + 31) // 32)
            assert c["clear"] == (c["total"] if tracking else 0)              # created cleared
            for i in (0, 0, 1, 4, 4, 2, 3, 0, 4, 1):
                over_cleared(i)
                c = rt.region_census()
                assert c["unknown"] + c["specular_constant"] + c["clear"] == c["total"]
                if tracking:
                    saw_clear |= c["clear"] > 0
                    saw_spec |= c["specular_constant"] > 0
                    if i == 4:
                        assert c["clear"] == c["total"], "a sky-only frame leaves every region clear"
                else:
                    assert c["unknown"] == c["total"]
            # foreign writes
            rt.upload("specular", junk32)
            assert rt.region_census()["unknown"] == rt.region_census()["total"]
            over_cleared(0, what="after a foreign write to the specular plane:")
            over_cleared(0)
            rt.upload("depth", junk32.view(np.float32))
            over_cleared(4, what="after a foreign write to the depth plane:")
            # clear, then keep-what-is-there passes (two views composed by the depth test)
            rt.Clear(); cur[0] = oracle.GBufferHost(w, h)
            keep(1); keep(0); keep(0); keep(4)
            over_cleared(0); keep(1, what="over a tracked frame:"); over_cleared(1)
            # other variants of the tile pass in between: they write planes without keeping the states
            over_cleared(0)
            tp.Render(views[1], views[1], rt, vr.default_render_params(400.0, assume_cleared=1, wireframe=1))
            over_cleared(0, what="after a wireframe pass:")
            tp.Render(views[4], views[4], rt, vr.default_render_params(400.0, assume_cleared=1, depth_only=1))
            over_cleared(0, what="after a depth-only pass:")
            gpu_ctx.set_raster_tile(64)
            over_cleared(1, what="64-pixel tiles:"); over_cleared(1, what="64-pixel tiles:")
            gpu_ctx.set_raster_tile(0)
            over_cleared(1, what="back on 32-pixel tiles:"); over_cleared(0)
            # a partition writes its own tiles only: the other tiles' states stay valid for what they hold
            over_cleared(1)
            over_cleared(0, vr.Partition(1, 2))
            got = planes(rt)
            ty, tx = np.indices((h, w))
            owned = ((tx // 128 + ty // 128) % 2) == 1
            want0 = oracle.GBufferHost(w, h); ot.render(views[0], want0, rp_k)
            for k in names:
                assert np.array_equal(got[k][owned], getattr(want0, k)[owned]) and np.array_equal(got[k][~owned], getattr(cur[0], k)[~owned]), k
            over_cleared(0, what="after a partitioned pass:"); over_cleared(4); over_cleared(0)
        finally:
            gpu_ctx.set_raster_tile(0)
            gpu_ctx.set_plane_tracking(True)
            rt.close()
    assert saw_clear and saw_spec


def test_clear_is_lazy_under_the_tracking_and_never_visible(scene256, oracle, gpu_ctx):
    """RenderTargets::Clear (Renderer.cpp:382) under VR_OPT_PLANE_TRACKING writes nothing by itself: the next whole-frame shaded tile
    pass runs as 'over a cleared target', anything else that looks at the planes first (download, a lighting pass, a rank's share,
    a depth-only pass, upload) gets the clear values written then.  Whatever the order, the planes read as Clear + what followed."""
    ot, tp = scene256["ot"], scene256["tp"]
    w, h = 416, 236
    va, vb = (vr.make_view(*scaled_camera(CAMERAS[i], 256), w, h) for i in (0, 5))
    rp_k = vr.default_render_params(400.0)
    names = ("depth", "diffuse", "specular", "normals", "emissive")
    want_a, want_b, cleared = oracle.GBufferHost(w, h), oracle.GBufferHost(w, h), oracle.GBufferHost(w, h)
    ot.render(va, want_a, rp_k); ot.render(vb, want_b, rp_k)

    def planes(rt):
        return {k: rt.download(k) for k in names}
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    hdr, hdr2 = vr.HdrImage(gpu_ctx, w, h), vr.HdrImage(gpu_ctx, w, h)
    dl = vr.DeferredLightingPass(gpu_ctx)
    sun = [vr.reference_sun()]
    try:
        # Clear + Render, the reference's own sequence, frame after frame: no clear kernel runs
        tp.Render(va, va, rt, rp_k)
        gpu_ctx.synchronize()
        gpu_ctx.timing_enable(1)
        for v, want in ((vb, want_b), (va, want_a), (vb, want_b)):
            rt.Clear(); tp.Render(v, v, rt, rp_k)
        gpu_ctx.synchronize()
        kernels = gpu_ctx.timing_collect()
        gpu_ctx.timing_enable(False)
        assert "k_raster" in kernels and not any("clear" in k for k in kernels), sorted(kernels)
        _assert_gbuffer_equal(want_b, planes(rt), "Clear + Render x 3")
        # a reader first: the clear happens for it
        rt.Clear()
        _assert_gbuffer_equal(cleared, planes(rt), "Clear, then download")
        tp.Render(va, va, rt, rp_k); rt.Clear()
        dl.Render(va, rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)                    # lights a cleared G-buffer
        rt2 = vr.RenderTargets(gpu_ctx).Init(w, h)
        dl.Render(va, rt2, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr2)
        assert np.array_equal(hdr.download(), hdr2.download())
        rt2.close()
        # a rank's share / a depth-only pass over a pending clear
        tp.Render(vb, vb, rt, rp_k); rt.Clear()
        tp.Render(va, va, rt, rp_k, vr.Partition(0, 2))
        got = planes(rt)
        ty, tx = np.indices((h, w))
        owned = ((tx // 128 + ty // 128) % 2) == 0
        for k in names:
            assert np.array_equal(got[k][owned], getattr(want_a, k)[owned]) and np.array_equal(got[k][~owned], getattr(cleared, k)[~owned]), k
        tp.Render(vb, vb, rt, rp_k); rt.Clear()
        tp.Render(va, va, rt, vr.default_render_params(400.0, depth_only=1))
        got = planes(rt)
        assert np.array_equal(got["depth"].view(np.uint32), want_a.depth.view(np.uint32))
        for k in names[1:]:
            assert np.array_equal(got[k], getattr(cleared, k)), k
        # upload over a pending clear: the other planes are clear, the uploaded one holds the upload
        tp.Render(vb, vb, rt, rp_k); rt.Clear()
        junk = np.full((h, w), 0x01020304, np.uint32)
        rt.upload("diffuse", junk)
        got = planes(rt)
        assert np.array_equal(got["diffuse"], junk) and np.array_equal(got["depth"], cleared.depth) and np.array_equal(got["normals"], cleared.normals)
        # two clears, then a composite of two views by the depth test
        rt.Clear(); rt.Clear()
        both = oracle.GBufferHost(w, h)
        ot.render(va, both, rp_k); ot.render(vb, both, rp_k)
        tp.Render(va, va, rt, rp_k); tp.Render(vb, vb, rt, rp_k)
        _assert_gbuffer_equal(both, planes(rt), "Clear, Render a, Render b")
    finally:
        gpu_ctx.timing_enable(False)
        hdr.close(); hdr2.close(); rt.close()


def test_emissive_plane_tracking_never_changes_the_gbuffer(scene256, oracle, gpu_ctx):
    """VR_OPT_PLANE_TRACKING: main_ps writes 0 to the emissive target (terrain_ps.hlsl:80) and Clear writes 0, so the tile pass
    skips the plane while the library knows it is all zero.  Whatever the history - fresh target, foreign writes (upload),
    partitioned and keep-what-is-there passes, pointers handed out - the five planes equal the oracle's, and the library's
    knowledge follows the rules of include/vrterrain.h."""
    ot, tp = scene256["ot"], scene256["tp"]
    w, h = 512, 288
    v = vr.make_view(*scaled_camera(CAMERAS[0], 256), w, h)
    rp_c, rp_k = vr.default_render_params(400.0, assume_cleared=1), vr.default_render_params(400.0)
    want = oracle.GBufferHost(w, h)
    ot.render(v, want, rp_k)                                   # Clear + Render
    rng = np.random.default_rng(11)
    junk = rng.integers(1, 65535, (h, w, 4), dtype=np.uint16)
    names = ("depth", "diffuse", "specular", "normals", "emissive")

    def planes(rt):
        return {k: rt.download(k) for k in names}
    for tracking in (True, False):
        gpu_ctx.set_plane_tracking(tracking)
        try:
            rt = vr.RenderTargets(gpu_ctx).Init(w, h)
            assert rt.plane_known_zero() == tracking           # created cleared
            tp.Render(v, v, rt, rp_c)
            _assert_gbuffer_equal(want, planes(rt), f"fresh target (tracking {tracking})")
            assert rt.plane_known_zero() == tracking
            # a foreign write: the knowledge is gone; a partitioned pass over a 'cleared' target zeroes its own tiles only
            rt.upload("emissive", junk)
            assert not rt.plane_known_zero()
            part = vr.Partition(1, 3)
            tp.Render(v, v, rt, rp_c, part)
            got = planes(rt)
            ty, tx = np.indices((h, w))
            owned = ((tx // 128 + ty // 128) % 3) == 1
            assert np.array_equal(got["emissive"][owned], want.emissive[owned]) and np.array_equal(got["emissive"][~owned], junk[~owned])
            for k in ("diffuse", "specular", "normals"):
                assert np.array_equal(got[k][owned], getattr(want, k)[owned]), k
            assert not rt.plane_known_zero()
            # a whole-target pass over a 'cleared' target writes every pixel's emissive texel: known zero again
            tp.Render(v, v, rt, rp_c)
            _assert_gbuffer_equal(want, planes(rt), f"whole-target pass after a foreign write (tracking {tracking})")
            assert rt.plane_known_zero() == tracking
            tp.Render(v, v, rt, rp_c)                          # ... and this one skips the plane
            _assert_gbuffer_equal(want, planes(rt), f"second pass (tracking {tracking})")
            # keep-what-is-there pass on junk: zeros where the terrain is drawn, junk elsewhere; not known zero
            rt.Clear(); rt.upload("emissive", junk)
            tp.Render(v, v, rt, rp_k)
            got = planes(rt)
            cov = want.depth < 1.0
            assert np.array_equal(got["emissive"][cov], want.emissive[cov]) and np.array_equal(got["emissive"][~cov], junk[~cov])
            assert not rt.plane_known_zero()
            # ... on a cleared target it stays known
            rt.Clear(); assert rt.plane_known_zero() == tracking
            tp.Render(v, v, rt, rp_k)
            _assert_gbuffer_equal(want, planes(rt), f"Clear + Render (tracking {tracking})")
            assert rt.plane_known_zero() == tracking
            # pointers handed out: never known again, not even after a clear
            rt.describe(); assert not rt.plane_known_zero()
            rt.Clear(); assert not rt.plane_known_zero()
            rt.upload("emissive", junk)
            tp.Render(v, v, rt, rp_c)
            _assert_gbuffer_equal(want, planes(rt), f"after describe (tracking {tracking})")
            rt.close()
        finally:
            gpu_ctx.set_plane_tracking(True)


def test_deferred_spot_and_spherical_lights(scene256, oracle, gpu_ctx):
    """All three Donut light types through the streaming pass (ShadeSurface: cone falloff by
    1 - smoothstep(inner, outer, angle), spherical sources with a per-pixel half angle)."""
    w, h = 320, 180
    eye, tgt = scaled_camera(CAMERAS[0], 256)
    v, gb_o, planes, _, _ = _render_both(scene256, oracle, gpu_ctx, eye, tgt, w, h)
    lights = [vr.reference_sun(), vr.point_light((10.0, 40.0, -5.0), 3000.0, 120.0, (1.0, 0.5, 0.25)),
              vr.spot_light((-20.0, 60.0, 10.0), (0.3, -1.0, -0.2), 6000.0, 200.0, 12.0, 25.0, (0.2, 1.0, 0.4)),
              vr.point_light((30.0, 35.0, -30.0), 2000.0, 150.0, (0.9, 0.9, 1.0), radius=6.0),
              vr.spot_light((0.0, 80.0, -40.0), (0.0, -1.0, 0.3), 9000.0, 0.0, 5.0, 40.0, (1.0, 0.2, 0.2), radius=3.0)]
    ref32 = oracle.deferred(v, gb_o, lights, AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
    base = oracle.deferred(v, gb_o, lights[:2], AMBIENT_TOP, AMBIENT_BOTTOM, f32=True)
    assert np.abs(ref32 - base).max() > 1e-3
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    for k, arr in (("depth", gb_o.depth), ("diffuse", gb_o.diffuse), ("specular", gb_o.specular),
                   ("normals", gb_o.normals), ("emissive", gb_o.emissive)):
        rt.upload(k, arr)
    hdr = vr.HdrImage(gpu_ctx, w, h)
    vr.DeferredLightingPass(gpu_ctx).Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    got = oracle.half_to_float(hdr.download()).astype(np.float64)
    for c in range(3):
        assert np.sqrt(np.mean((got[..., c] - ref32[..., c]) ** 2)) <= 1e-4
    with pytest.raises(vr.VrError):           # the tiled pass takes directional + punctual point lights only
        vr.TiledDeferredLightingPass(gpu_ctx).Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
    hdr.close()
    rt.close()


def test_two_rank_frame_split_rehearsal_on_one_gpu():
    """bench.py's N>1 control flow (tile partition, packed RGB16F tiles, all-gather, de-tile) with two
    ranks sharing this GPU and gloo standing in for RCCL; every rank compares the assembled frame with
    its own unsplit render, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--width", "1920", "--height", "1080", "--rehearse-on-one-gpu", "--verify", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["frame_verified_against_unsplit"] is True


@pytest.mark.parametrize("mode", ["ldr", "hdr", "ldr-one-stream"])
def test_bench_n_rank_path_through_rccl_and_the_c_abi(mode):
    """bench.py's REAL N > 1 code path on this one GPU (--force-dist: a one-rank process group on the nccl backend): its own
    ncclComm_t (vrenderer_amd/rccl.py), packed tiles, the histogram all-reduce and the all-gather through the exported
    exchange (vr_tonemap_allreduce_histogram, vr_frame_allgather_tiles + vr_frame_detile[_ldr], or - on one stream -
    vr_frame_allgather[_ldr]), de-tile; the assembled frame equals the unsplit render and the line carries the `exchange`
    record.  What the first 8-GPU run executes, minus the peers."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "1", "--width", "1920", "--height", "1080",
           "--verify", "--no-cpu-baseline", "--exchange", mode.split("-")[0]] + (["--no-overlap"] if mode.endswith("one-stream") else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout (RCCL's banner must not land there)"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["frame_verified_against_unsplit"] is True
    x = out["exchange"]
    assert "C ABI" in x["through"] and ("vr_frame_allgather_ldr" in x["through"] if mode.endswith("one-stream") else "vr_frame_allgather_tiles" in x["through"])
    assert x["allgather_us"] > 0 and x["frame_period_us"] > 0 and x["bytes_received_per_rank"] == 0
    assert (x["allreduce_us"] > 0) == (mode != "hdr")
    assert "k_raster" in out["kernels"] and ("k_detile_ldr" if mode != "hdr" else "k_detile") in out["kernels"]
    assert out["sustained"]["value"] > 0


@pytest.mark.parametrize("cross_stream", [False, True])
def test_frame_submit_equals_the_per_call_sequence(scene256, oracle, gpu_ctx, cross_stream):
    """vr_frame_submit queues what Render + Prepare x 2 + Light + the tone-map stage queue: six frames in flight (no host
    synchronisation in between, two rotating tile/image buffers, the tone mapper on the terrain's stream or on its own) give
    the same HdrColor and LdrColor bytes as the per-call sequence - the first frame against the oracle as well."""
    import torch
    ot, tp = scene256["ot"], scene256["tp"]
    w, h = 512, 288
    views = [vr.make_view(*scaled_camera(CAMERAS[k], 256), w, h) for k in (0, 1, 5, 3, 0, 6)]
    rp = vr.default_render_params(400.0, assume_cleared=1)
    lights = [vr.reference_sun()]
    tmp = vr.default_tonemap_params()
    side = torch.cuda.Stream() if cross_stream else None
    tctx = vr.Context(0) if cross_stream else gpu_ctx
    if cross_stream:
        tctx.set_stream(side.cuda_stream)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    hdrs = [vr.HdrImage(gpu_ctx, w, h) for _ in range(2)]
    ldrs = [vr.LdrImage(tctx, w, h) for _ in range(2)]

    def run(submit):
        tm = vr.ToneMappingPass(tctx)
        tm.AdvanceFrame(1.0 / 60.0)
        fr = vr.Frame(tp, rt, rp, lights, AMBIENT_TOP, AMBIENT_BOTTOM, tonemap=tm, tonemap_params=tmp, ldr=ldrs[0]) if submit else None
        outs = []
        for i, v in enumerate(views):
            b = i % 2
            if i >= 2:                                  # the host reads a buffer back before its slot is used again
                gpu_ctx.synchronize(); tctx.synchronize()
                outs.append((hdrs[b].download().copy(), ldrs[b].download().copy()))
            if submit:
                fr.submit(v, hdrs[b], views[i + 1:i + 3], ldr=ldrs[b])
            else:
                tp.Render(v, v, rt, rp)
                for nv in views[i + 1:i + 3]:
                    tp.Prepare(nv, rt, rp)
                vr.DeferredLightingPass(gpu_ctx).Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdrs[b])
                if cross_stream:
                    gpu_ctx.synchronize()               # the per-call reference orders the two streams the blunt way
                tm.SimpleRender(tmp, hdrs[b], ldrs[b])
                if cross_stream:
                    tctx.synchronize()
        gpu_ctx.synchronize(); tctx.synchronize()
        for b in (0, 1):
            outs.append((hdrs[b].download().copy(), ldrs[b].download().copy()))
        tm.close()
        return outs
    try:
        want, got = run(False), run(True)
        assert len(want) == len(got) == len(views)
        for k, ((h0, l0), (h1, l1)) in enumerate(zip(want, got)):
            assert np.array_equal(h0, h1), f"HdrColor of frame slot {k} differs"
            assert np.array_equal(l0, l1), f"LdrColor of frame slot {k} differs"
        gb = oracle.GBufferHost(w, h)
        ot.render(views[0], gb, vr.default_render_params(400.0))
        ref = oracle.deferred(views[0], gb, lights, AMBIENT_TOP, AMBIENT_BOTTOM)
        a, b_ = oracle.half_to_float(got[0][0]).astype(np.float64), oracle.half_to_float(ref).astype(np.float64)
        assert np.sqrt(np.mean((a - b_) ** 2)) <= 1e-4
    finally:
        if cross_stream:
            torch.cuda.synchronize()
        for o in ldrs + hdrs + [rt]:
            o.close()
        if cross_stream:
            tctx.close()


def _fused_vs_unfused(tp, gpu_ctx, v, w, h, lights, part=None):
    from vrenderer_amd.passes import partition_info
    rp = vr.default_render_params(400.0, assume_cleared=1)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    if part is None:
        mk = lambda: vr.HdrImage(gpu_ctx, w, h)
        nbytes = None
    else:
        info = partition_info(w, h, part.rank, part.world_size)
        rows = (info["packed_bytes"] + vr.VR_OWNER_TILE * 8 - 1) // (vr.VR_OWNER_TILE * 8)
        mk = lambda: vr.HdrImage(gpu_ctx, vr.VR_OWNER_TILE, rows)
        nbytes = info["owned"] * 128 * 128 * 6            # the live tiles of the packed buffer
    a, b = mk(), mk()
    for img in (a, b):                                     # (pixels of a partial edge tile that lie outside the frame are never written)
        img.upload(np.zeros(img.width * img.height * 4, np.uint16))
    tp.Render(v, v, rt, rp, part)
    vr.DeferredLightingPass(gpu_ctx).Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, a, part)
    depth_a = rt.download("depth").copy()
    want = a.download(nbytes).copy()
    rt.Clear()                                             # (the fused pass must not depend on what the other planes hold)
    tp.RenderLit(v, rt, rp, lights, AMBIENT_TOP, AMBIENT_BOTTOM, b, part)
    got = b.download(nbytes)
    depth_b = rt.download("depth")
    for o in (a, b, rt):
        o.close()
    return want, got, depth_a, depth_b


def test_fused_render_lit_equals_render_plus_lighting(scene256, scene2048, oracle, gpu_ctx):
    """vr_terrain_render_lit (opt-in): the tile pass's resolve shades what it has just encoded and writes depth +
    HdrColor only.  Bit-identical to vr_terrain_render + vr_deferred_light - sun only, sun + point lights (the exact-position
    branch), whole frames and the packed tiles of a 3-way split, both raster tile sizes - and within the stated 1e-4 RMS of the
    oracle like the unfused pair; spot lights fall back to the two passes inside the call."""
    sun = [vr.reference_sun()]
    lamps = sun + [vr.point_light((10.0, 40.0, -5.0), 3000.0, 120.0, (1.0, 0.5, 0.25)), vr.point_light((-30.0, 35.0, 30.0), 2000.0, 150.0, (0.9, 0.9, 1.0))]
    spots = sun + [vr.spot_light((-20.0, 60.0, 10.0), (0.3, -1.0, -0.2), 6000.0, 200.0, 12.0, 25.0, (0.2, 1.0, 0.4))]
    cases = [(scene256, 256, 512, 288, CAMERAS[0], sun, None), (scene256, 256, 512, 288, CAMERAS[5], lamps, None),
             (scene256, 256, 640, 360, CAMERAS[1], lamps, vr.Partition(1, 3)), (scene2048, 2048, 1920, 1080, CAMERAS[0], sun, None),
             (scene2048, 2048, 1920, 1080, CAMERAS[3], lamps, vr.Partition(5, 8)), (scene256, 256, 512, 288, CAMERAS[0], spots, None)]
    for tile in (0, 64):
        gpu_ctx.set_raster_tile(tile)
        try:
            for sc, size, w, h, cam, lights, part in cases:
                v = vr.make_view(*scaled_camera(cam, size), w, h)
                want, got, da, db = _fused_vs_unfused(sc["tp"], gpu_ctx, v, w, h, lights, part)
                what = f"{w}x{h} scene {size} lights {len(lights)} part {None if part is None else (part.rank, part.world_size)} tile {tile}"
                assert np.array_equal(da.view(np.uint32), db.view(np.uint32)), "depth: " + what
                assert np.array_equal(want, got), f"HdrColor differs at {(want != got).sum()} halfs: " + what
        finally:
            gpu_ctx.set_raster_tile(0)
    # ... and against the oracle (one case; the unfused pair has its own tests)
    w, h = 512, 288
    v = vr.make_view(*scaled_camera(CAMERAS[0], 256), w, h)
    _, got, _, _ = _fused_vs_unfused(scene256["tp"], gpu_ctx, v, w, h, sun)
    gb = oracle.GBufferHost(w, h)
    scene256["ot"].render(v, gb, vr.default_render_params(400.0))
    ref = oracle.deferred(v, gb, sun, AMBIENT_TOP, AMBIENT_BOTTOM)
    d = oracle.half_to_float(got).astype(np.float64) - oracle.half_to_float(ref).astype(np.float64)
    assert np.sqrt(np.mean(d ** 2)) <= 1e-4


def test_fused_render_lit_at_8k_and_split_eight_ways(scene2048, gpu_ctx):
    """The fused variant at BASELINE's size: the whole 7680x4320 frame and all eight ranks' packed tiles equal the unfused pair
    byte for byte (a size-independent property: both sides are the HIP path)."""
    from vrenderer_amd.passes import frame_detile
    w, h = 7680, 4320
    v = vr.make_view(*CAMERAS[1], w, h)
    sun = [vr.reference_sun()]
    want, got, da, db = _fused_vs_unfused(scene2048["tp"], gpu_ctx, v, w, h, sun)
    assert np.array_equal(da.view(np.uint32), db.view(np.uint32)) and np.array_equal(want, got)
    assert (da < 1.0).mean() > 0.3
    for r in range(8):
        wp, gp, _, _ = _fused_vs_unfused(scene2048["tp"], gpu_ctx, v, w, h, sun, vr.Partition(r, 8))
        assert np.array_equal(wp, gp), f"rank {r} of 8"


def _render_view_both(sc, oracle, gpu_ctx, v, w, h, **rpkw):
    ot, tp = sc["ot"], sc["tp"]
    rp = vr.default_render_params(400.0, **rpkw)
    gb_o = oracle.GBufferHost(w, h)
    ot.render(v, gb_o, rp, None)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    tp.Render(v, v, rt, rp, None)
    planes = {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}
    rt.close()
    return gb_o, planes


@pytest.mark.parametrize("wireframe", [0, 1])
def test_sub_viewport_inside_a_larger_target(scene256, oracle, gpu_ctx, wireframe):
    """nvrhi::Viewport smaller than the framebuffer (ViewportState, TerrainPass.cpp:213-216): the view's
    viewport rectangle positions and clips the image; nothing outside it is touched."""
    w, h = 640, 360
    eye, tgt = scaled_camera(CAMERAS[0], 256)
    v = vr.make_view(eye, tgt, 500, 300)              # projection for a 500x300 viewport ...
    v.viewport_x, v.viewport_y = 37, 21                # ... placed at (37, 21): cuts through raster tiles on all four sides
    gb_o, planes = _render_view_both(scene256, oracle, gpu_ctx, v, w, h, wireframe=wireframe)
    _assert_gbuffer_equal(gb_o, planes, "sub-viewport")
    inside = np.zeros((h, w), bool)
    inside[21:321, 37:537] = True
    assert (planes["depth"][~inside] == 1.0).all() and not planes["diffuse"][~inside].any()
    assert (planes["depth"][inside] < 1.0).any()


def test_guard_band_clipping_of_huge_triangles(scene256, oracle, gpu_ctx):
    """Looking straight down from half a unit with a 0.01 degree field of view, two triangles fill the
    frame and their vertices project beyond the +-100 NDC guard band: they go through the clipper."""
    hgt = float(scene256["h"][128 - 2, 128 + 1]) / 255.0 * 400.0
    w, h = 512, 288
    for fov, expect_clip in ((0.01, True), (0.1, False)):
        v = vr.make_view((1.3, hgt + 0.5, -2.2), (1.32, hgt - 5.0, -2.18), w, h, vfov_deg=fov)
        gb_o, planes = _render_view_both(scene256, oracle, gpu_ctx, v, w, h)
        _assert_gbuffer_equal(gb_o, planes, f"guard band, fov {fov}")
        st = scene256["tp"].render_stats()
        assert (st["clipped_tris"] > 0) == expect_clip and st["flags"] == 0, st
        assert (planes["depth"] < 1.0).all()


def _lit_frame(sc, gpu_ctx, cam_index, w, h, part=None, hdr=None):
    eye, tgt = scaled_camera(CAMERAS[cam_index], sc["size"])
    v = vr.make_view(eye, tgt, w, h)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    sc["tp"].Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1), part)
    out = hdr if hdr is not None else vr.HdrImage(gpu_ctx, w, h)
    vr.DeferredLightingPass(gpu_ctx).Render(v, rt, [vr.reference_sun()], AMBIENT_TOP, AMBIENT_BOTTOM, out, part)
    rt.close()
    return out


@pytest.mark.parametrize("w,h", [(640, 360), (333, 187)])
def test_tonemap_simple_render_bit_exact(scene256, oracle, gpu_ctx, w, h):
    """ToneMappingPass::SimpleRender (Renderer.cpp:430-431) over three frames with eye adaptation: histogram,
    adapted luminance (bit pattern) and SRGBA8 LdrColor equal the oracle's on the same HdrColor input."""
    p = vr.default_tonemap_params()
    tm_g, tm_o = vr.ToneMappingPass(gpu_ctx), oracle.ToneMapper()
    ldr = vr.LdrImage(gpu_ctx, w, h)
    for frame, cam in enumerate((0, 7, 3)):
        hdr = _lit_frame(scene256, gpu_ctx, cam, w, h)
        tm_g.AdvanceFrame(1.0 / 60.0); tm_o.AdvanceFrame(1.0 / 60.0)
        tm_g.SimpleRender(p, hdr, ldr)
        want = tm_o.SimpleRender(p, hdr.download())
        hist, lum = tm_g.download()
        assert np.array_equal(hist, tm_o.hist), f"frame {frame}: histogram"
        assert hist.sum() == 64 * w * h
        assert np.float32(lum).view(np.uint32) == np.float32(tm_o.adapted).view(np.uint32), (frame, lum, tm_o.adapted)
        got = ldr.download()
        mis = np.argwhere(got != want)
        assert mis.size == 0, f"frame {frame}: LDR differs at {len(mis)} bytes, first {mis[:4].tolist()}"
        assert got[..., :3].max() > 100 and (got[..., 3] == 255).all()
        hdr.close()
    ldr.close(); tm_g.close()


@pytest.mark.parametrize("world", [2, 3])
def test_tonemap_on_packed_tiles_and_ldr_detile(scene256, oracle, gpu_ctx, world):
    """N ranks emulated on one GPU: each rank's packed RGB16F tiles -> shared histogram (what the all-reduce
    produces) -> exposure -> packed RGB8 tiles -> concatenation (the all-gather) -> vr_frame_detile_ldr.
    The assembled SRGBA8 frame equals the oracle's tone-mapped unsplit frame."""
    from vrenderer_amd.passes import frame_detile_ldr, partition_info, partition_prepare
    w, h = 640, 360
    p = vr.default_tonemap_params()
    full = _lit_frame(scene256, gpu_ctx, 5, w, h)
    tm_o = oracle.ToneMapper()
    want = tm_o.SimpleRender(p, full.download())
    info = partition_info(w, h, 0, world)
    rows = (info["packed_bytes"] + 8 * 128 - 1) // (8 * 128)
    tm = vr.ToneMappingPass(gpu_ctx)
    tm.ResetHistogram()
    packed = []
    for r in range(world):
        part = vr.Partition(r, world)
        buf = vr.HdrImage(gpu_ctx, 128, rows)
        _lit_frame(scene256, gpu_ctx, 5, w, h, part=part, hdr=buf)
        tm.AddFrameToHistogram(p, buf, w, h, part)             # accumulates: the sum over ranks
        packed.append(buf)
    hist, _ = tm.download()
    assert np.array_equal(hist, tm_o.hist)
    tm.ComputeExposure(p)
    gathered = np.zeros(world * info["packed_bytes_ldr"], np.uint8)
    for r in range(world):
        out = vr.LdrImage(gpu_ctx, w, h, capacity_bytes=info["packed_bytes_ldr"])
        tm.Render(p, packed[r], out, w, h, vr.Partition(r, world))
        gathered[r * info["packed_bytes_ldr"]:(r + 1) * info["packed_bytes_ldr"]] = out.download(info["packed_bytes_ldr"])
        out.close()
    g_rows = (gathered.nbytes + 8 * 1024 - 1) // (8 * 1024)
    g_dev = vr.HdrImage(gpu_ctx, 1024, g_rows)
    pad = np.zeros(g_rows * 8 * 1024, np.uint8); pad[:gathered.nbytes] = gathered
    g_dev.upload(pad)
    frame = vr.LdrImage(gpu_ctx, w, h)
    partition_prepare(gpu_ctx, w, h, vr.Partition(0, world))
    frame_detile_ldr(gpu_ctx, g_dev.device_ptr, world, w, h, frame)
    got = frame.download()
    mis = np.argwhere(got != want)
    assert mis.size == 0, f"assembled LDR frame differs at {len(mis)} bytes, first {mis[:4].tolist()}"
    for b in packed: b.close()
    g_dev.close(); frame.close(); tm.close(); full.close()


def _shadow_setup(sc, oracle, gpu_ctx, cam_index, w, h, res, bias):
    size = sc["size"]
    eye, tgt = scaled_camera(CAMERAS[cam_index], size)
    cam = vr.make_view(eye, tgt, w, h)
    sun = vr.reference_sun()
    sm = vr.CascadedShadowMap(gpu_ctx, vr.default_shadow_params(float(size), resolution=res, depth_bias=bias))
    lv = sm.SetupForPlanarViewStable(sun, cam)
    sm.Clear()
    sm.RenderTerrain(sc["tp"])                                  # "Terrain Shadow" (Renderer.cpp:356-372)
    # the same pass through the oracle
    gb_l = oracle.GBufferHost(res, res)
    sc["ot"].render(lv, gb_l, vr.default_render_params(400.0, depth_only=1))
    return cam, sun, sm, lv, gb_l


def test_terrain_shadow_pass_depth_bit_exact(scene256, oracle, gpu_ctx):
    """TerrainPass::Render(depthOnly) from the cascade's orthographic light view (Renderer.cpp:356-372)."""
    cam, sun, sm, lv, gb_l = _shadow_setup(scene256, oracle, gpu_ctx, 0, 640, 360, 512, 0.0)
    got = sm.download_depth()
    mis = np.argwhere(got.view(np.uint32) != gb_l.depth.view(np.uint32))
    assert mis.size == 0, f"shadow depth differs at {len(mis)} texels, first {mis[:4].tolist()}"
    assert (got < 1.0).mean() > 0.001 and scene256["tp"].num_chunks() > 0
    sm.close()


@pytest.mark.parametrize("cam_index,bias", [(0, 0.0), (5, 0.002), (7, 0.002)])
def test_shadowed_deferred_rms(scene256, oracle, gpu_ctx, cam_index, bias):
    """DeferredLightingPass with DirectionalLight::shadowMap set (Renderer.cpp:336, 427): per-channel RMS <= 1e-4
    vs the oracle, and no pixel flips its PCF comparisons (max error stays at rounding level)."""
    w, h, res = 640, 360, 512
    cam, sun, sm, lv, gb_l = _shadow_setup(scene256, oracle, gpu_ctx, cam_index, w, h, res, bias)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    scene256["tp"].Render(cam, cam, rt, vr.default_render_params(400.0, assume_cleared=1))
    hdr, hdr_plain = vr.HdrImage(gpu_ctx, w, h), vr.HdrImage(gpu_ctx, w, h)
    dl = vr.DeferredLightingPass(gpu_ctx)
    dl.Render(cam, rt, [sun], AMBIENT_TOP, AMBIENT_BOTTOM, hdr, shadow_map=sm)
    dl.Render(cam, rt, [sun], AMBIENT_TOP, AMBIENT_BOTTOM, hdr_plain)
    gb = oracle.GBufferHost(w, h)
    for name, arr in (("depth", gb.depth), ("diffuse", gb.diffuse), ("specular", gb.specular), ("normals", gb.normals), ("emissive", gb.emissive)):
        arr[...] = rt.download(name)
    want = oracle.deferred(cam, gb, [sun], AMBIENT_TOP, AMBIENT_BOTTOM, f32=True, shadow=(lv, gb_l.depth, 0, bias))
    got = oracle.half_to_float(hdr.download()).astype(np.float64)
    plain = oracle.half_to_float(hdr_plain.download()).astype(np.float64)
    err = got[..., :3] - want[..., :3].astype(np.float64)
    rms = np.sqrt((err ** 2).mean(axis=(0, 1)))
    assert (rms <= 1e-4).all(), rms
    assert np.abs(err).max() < 2e-3, np.abs(err).max()          # half rounding of values < 1; a flipped comparison would be ~0.1
    shadowed = (plain[..., 1] - got[..., 1]) > 1e-3
    assert shadowed.mean() > 0.01, "the frame has no shadowed pixels: the test would not exercise the lookup"
    assert (got <= plain + 1e-3).all()
    for o in (hdr, hdr_plain, rt, sm): o.close()


def test_shadowed_deferred_on_packed_tiles(scene256, oracle, gpu_ctx):
    w, h, res, world = 640, 360, 512, 2
    cam, sun, sm, lv, _ = _shadow_setup(scene256, oracle, gpu_ctx, 5, w, h, res, 0.002)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    dl = vr.DeferredLightingPass(gpu_ctx)
    scene256["tp"].Render(cam, cam, rt, vr.default_render_params(400.0, assume_cleared=1))
    full = vr.HdrImage(gpu_ctx, w, h)
    dl.Render(cam, rt, [sun], AMBIENT_TOP, AMBIENT_BOTTOM, full, shadow_map=sm)
    ref = full.download()
    from vrenderer_amd import partition as pt
    from vrenderer_amd.passes import partition_info
    info = partition_info(w, h, 0, world)
    rows = (info["packed_bytes"] + 8 * 128 - 1) // (8 * 128)
    for r in range(world):
        part = vr.Partition(r, world)
        buf = vr.HdrImage(gpu_ctx, 128, rows)
        dl.Render(cam, rt, [sun], AMBIENT_TOP, AMBIENT_BOTTOM, buf, part, shadow_map=sm)
        packed = buf.download(info["packed_bytes"]).reshape(-1, 128, 128, 3)
        tx, _ = pt.owner_grid(w, h)
        for lt, tile in enumerate(pt.owned_tiles(w, h, r, world)):
            y0, x0 = (tile // tx) * 128, (tile % tx) * 128
            hh, ww = min(128, h - y0), min(128, w - x0)
            assert np.array_equal(packed[lt, :hh, :ww], ref[y0:y0 + hh, x0:x0 + ww, :3]), (r, tile)
        buf.close()
    for o in (full, rt, sm): o.close()


def _fuzz_views(rng, n, heightmap, size):
    """Seeded random views: positions inside, above, below and far outside the terrain; any direction; rolled
    up-vectors; fields of view from 5 to 150 degrees; odd target sizes."""
    sizes = [(320, 180), (257, 131), (64, 64), (400, 96)]
    half, s = size / 2.0, size / 256.0
    for it in range(n):
        kind = it % 6
        if kind == 0:      # hovering close above the surface
            x, z = rng.uniform(-0.4 * size, 0.4 * size, 2)
            hx, hz = int(np.clip(x + half, 0, size - 1)), int(np.clip(z + half, 0, size - 1))
            eye = (x, float(heightmap[hz, hx]) / 255.0 * 400.0 + rng.uniform(0.2, 8.0), z)
        elif kind == 1:    # far outside, high up
            ang = rng.uniform(0, 2 * np.pi)
            eye = (1.6 * size * np.cos(ang), rng.uniform(100, 600), 1.6 * size * np.sin(ang))
        elif kind == 2:    # below the terrain
            eye = (rng.uniform(-100, 100) * s, rng.uniform(-50, 20), rng.uniform(-100, 100) * s)
        else:              # anywhere in a box around the world
            eye = tuple(rng.uniform((-0.8 * size, 0, -0.8 * size), (0.8 * size, 450, 0.8 * size)))
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        if kind in (0, 1):
            d = np.array([-eye[0], -eye[1] * rng.uniform(0.2, 1.5), -eye[2]]) + rng.normal(size=3) * 20     # roughly at the terrain
            d /= np.linalg.norm(d)
        tgt = tuple(np.array(eye) + d * 50.0)
        up = rng.normal(size=3) if it % 4 == 3 else np.array([0.0, 1.0, 0.0])
        if abs(np.dot(up / np.linalg.norm(up), d)) > 0.95:
            up = np.array([1.0, 0.0, 0.0])
        w, h = sizes[it % len(sizes)]
        fov = float(rng.choice([5.0, 30.0, 60.0, 90.0, 150.0]))
        yield it, eye, tgt, tuple(up), fov, w, h


@pytest.mark.parametrize("scene_name,views,seed", [("scene256", 48, 20260104), ("scene2048", 18, 7)])
def test_random_camera_fuzz_bit_exact(scene_name, views, seed, request, oracle, gpu_ctx):
    """Seeded random views with random render options (wireframe, depth-only, a rank of a 2..4-way tile
    partition): node lists, instance data and all G-buffer planes must equal the oracle's bit for bit on
    every one of them.  (This test found the bin-key-0 bug: the first triangle of the first node was never shaded.)"""
    sc = request.getfixturevalue(scene_name)
    rng = np.random.default_rng(seed)
    opt = np.random.default_rng(seed + 1)
    ot, tp = sc["ot"], sc["tp"]
    checked_pixels = 0
    for it, eye, tgt, up, fov, w, h in _fuzz_views(rng, views, sc["h"], sc["size"]):
        v = vr.make_view(eye, tgt, w, h, vfov_deg=fov, up=up)
        world = int(opt.integers(1, 5))
        part = vr.Partition(int(opt.integers(0, world)), world) if world > 1 and it % 3 == 1 else None
        rp = vr.default_render_params(400.0, assume_cleared=int(opt.integers(0, 2)), wireframe=int(it % 7 == 6),
                                      depth_only=int(it % 11 == 10))
        what = f"fuzz view {it}: eye {eye} target {tgt} up {up} fov {fov} {w}x{h} part {(part.rank, part.world_size) if part else None}"
        n_o, ids_o, inst_o = ot.select(v, 400.0)
        if n_o > tp.params.max_instances:
            continue
        n_g, ids_g, inst_g = tp.NodeSelect(v, 400.0)
        assert n_g == n_o and np.array_equal(ids_g, ids_o) and np.array_equal(inst_g, inst_o), what
        gb_o = oracle.GBufferHost(w, h)
        ot.render(v, gb_o, rp, part)
        rt = vr.RenderTargets(gpu_ctx).Init(w, h)
        tp.Render(v, v, rt, rp, part)
        planes = {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}
        assert tp.render_stats()["flags"] == 0, what
        rt.close()
        _assert_gbuffer_equal(gb_o, planes, what)
        checked_pixels += int((planes["depth"] < 1.0).sum())
    assert checked_pixels > 3000 * views


def test_tonemap_fuzz_all_half_values(oracle, gpu_ctx):
    """HdrColor filled with random half bit patterns (every class: zeros, denormals, negatives, huge values, inf, NaN):
    histogram, adapted luminance and SRGBA8 output must equal the oracle's bit for bit."""
    rng = np.random.default_rng(99)
    w, h = 512, 256
    p = vr.default_tonemap_params()
    for trial in range(3):
        img = rng.integers(0, 65536, size=(h, w, 4), dtype=np.uint16)
        if trial == 1:       # mostly plausible radiance with a sprinkle of specials
            img = np.abs(rng.normal(0.2, 0.3, size=(h, w, 4))).astype(np.float16).view(np.uint16)
            special = rng.integers(0, h * w, 2000)
            img.reshape(-1, 4)[special, rng.integers(0, 3, 2000)] = rng.choice(
                np.array([0x7c00, 0xfc00, 0x7e00, 0x0001, 0x8001, 0x7bff, 0xfbff, 0x8000], np.uint16), 2000)
        if trial == 2:       # every half value appears in the red channel
            img[:, :, 0].reshape(-1)[:65536] = np.arange(65536, dtype=np.uint16)
        hdr = vr.HdrImage(gpu_ctx, w, h)
        hdr.upload(img)
        tm_g, tm_o = vr.ToneMappingPass(gpu_ctx), oracle.ToneMapper()
        ldr = vr.LdrImage(gpu_ctx, w, h)
        tm_g.AdvanceFrame(0.02); tm_o.AdvanceFrame(0.02)
        for _ in range(2):
            tm_g.SimpleRender(p, hdr, ldr)
            want = tm_o.SimpleRender(p, img)
        hist, lum = tm_g.download()
        assert np.array_equal(hist, tm_o.hist), trial
        assert np.float32(lum).view(np.uint32) == np.float32(tm_o.adapted).view(np.uint32), (trial, lum, tm_o.adapted)
        got = ldr.download()
        mis = np.argwhere(got != want)
        assert mis.size == 0, f"trial {trial}: {len(mis)} bytes differ, first {mis[:4].tolist()}: hdr {img[tuple(mis[0][:2])]} got {got[tuple(mis[0][:2])]} want {want[tuple(mis[0][:2])]}"
        for o in (hdr, ldr, tm_g): o.close()


def test_deferred_fuzz_random_gbuffer_and_lights(scene256, oracle, gpu_ctx):
    """Lighting pass on random surface data (unit normals, any roughness / albedo / F0 / occlusion / emissive) over real
    depth, with random mixes of all light types: every pixel within half-precision rounding of the oracle's fp32 value."""
    rng = np.random.default_rng(4711)
    w, h = 320, 180
    eye, tgt = scaled_camera(CAMERAS[5], 256)
    v, gb, _, _, _ = _render_both(scene256, oracle, gpu_ctx, eye, tgt, w, h)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    hdr = vr.HdrImage(gpu_ctx, w, h)
    dl = vr.DeferredLightingPass(gpu_ctx)
    for trial in range(4):
        n = rng.normal(size=(h, w, 3)); n /= np.linalg.norm(n, axis=-1, keepdims=True)
        rough = rng.uniform(0.0, 1.0, size=(h, w, 1))
        gb.normals[...] = np.round(np.concatenate([n, rough], -1) * 32767.0).astype(np.int16).view(np.uint16)
        gb.diffuse[...] = rng.integers(0, 2 ** 32, size=(h, w), dtype=np.uint32)
        gb.specular[...] = rng.integers(0, 2 ** 32, size=(h, w), dtype=np.uint32)
        gb.emissive[...] = np.abs(rng.normal(0, 0.05, size=(h, w, 4))).astype(np.float16).view(np.uint16)
        if trial == 3:       # cleared pixels: zero normal, zero everything
            gb.normals[:40] = 0; gb.diffuse[:40] = 0; gb.specular[:40] = 0; gb.emissive[:40] = 0; gb.depth[:40] = 1.0
        lights = []
        for i in range(int(rng.integers(1, 9))):
            kind = int(rng.integers(0, 4))
            pos = tuple(rng.uniform((-60, 20, -60), (60, 120, 60)))
            col = tuple(rng.uniform(0.1, 1.0, 3))
            if kind == 0:
                lights.append(vr.directional_light(tuple(rng.normal(size=3)), float(rng.uniform(0.2, 2.0)), float(rng.uniform(0.1, 5.0)), col))
            elif kind == 1:
                lights.append(vr.point_light(pos, float(rng.uniform(500, 5000)), float(rng.choice([0.0, 80.0, 200.0])), col))
            elif kind == 2:
                lights.append(vr.point_light(pos, float(rng.uniform(500, 5000)), 150.0, col, radius=float(rng.uniform(0.5, 8.0))))
            else:
                inner = float(rng.uniform(3, 30))
                lights.append(vr.spot_light(pos, tuple(rng.normal(size=3) - np.array([0, 1.5, 0])), float(rng.uniform(1000, 9000)),
                                            float(rng.choice([0.0, 250.0])), inner, inner + float(rng.uniform(2, 30)), col,
                                            radius=float(rng.choice([0.0, 2.0]))))
        for k, arr in (("depth", gb.depth), ("diffuse", gb.diffuse), ("specular", gb.specular), ("normals", gb.normals), ("emissive", gb.emissive)):
            rt.upload(k, arr)
        dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        got = oracle.half_to_float(hdr.download()).astype(np.float64)[..., :3]
        ref = oracle.deferred(v, gb, lights, AMBIENT_TOP, AMBIENT_BOTTOM, f32=True).astype(np.float64)[..., :3]
        assert np.isfinite(got).all() and np.isfinite(ref).all(), trial
        # half rounding is 4.9e-4 relative; GGX highlights at low roughness are ill-conditioned in fp32 (dd = NdotH^2 (a2-1) + 1
        # cancels), so a handful of highlight pixels sit further apart: 99.9 % of the values tight, all but a few in 10^5 within 3 %
        err = np.abs(got - ref)
        loose = np.argwhere(err > 3e-2 * np.abs(ref) + 1e-3)
        assert len(loose) <= 5e-5 * err.size, f"trial {trial} ({len(lights)} lights): {len(loose)} values off, first {loose[:3].tolist()}: got {got[tuple(loose[0])]} want {ref[tuple(loose[0])]}"
        assert (err <= 0.25 * np.abs(ref) + 1e-2).all(), trial
        assert (err > 1.5e-3 * np.abs(ref) + 1e-4).mean() < 1e-3, trial
        assert np.sqrt(np.mean((err / (np.abs(ref) + 1.0)) ** 2)) < 2e-4, trial
    hdr.close(); rt.close()


def test_tiled_deferred_fuzz_random_views_and_lights(scene256, oracle, gpu_ctx):
    """The tiled pass over seeded random views (close to the ground, far outside, below the terrain, rolled cameras, 5 to
    150 degrees of field of view, frames with sky) with random sets of directional and punctual lights - ranges from a
    few texels to the whole world, some lights outside the world, an unbounded one: HDR RMS <= 1e-4 against the oracle's
    all-lights loop, no overflow, and a 3-way partition's packed tiles reassemble to the unsplit frame byte for byte."""
    from vrenderer_amd.passes import frame_detile, partition_info
    rng = np.random.default_rng(20261004)
    tiled = vr.TiledDeferredLightingPass(gpu_ctx)
    done = 0
    for it, eye, tgt, up, fov, w, h in _fuzz_views(rng, 16, scene256["h"], 256):
        if w % 4:
            continue                                             # the tiled pass needs a width that is a multiple of 4
        v = vr.make_view(eye, tgt, w, h, vfov_deg=fov, up=up)
        rt, gb = _gpu_gbuffer_as_oracle_input(oracle, gpu_ctx, scene256["tp"], v, w, h)
        lights = [vr.directional_light(tuple(rng.normal(size=3) - np.array([0, 1.0, 0])), 0.5, 1.0, (1.0, 0.9, 0.8))]
        for i in range(int(rng.integers(20, 200))):
            pos = tuple(rng.uniform((-180, 0, -180), (180, 300, 180)))
            rng_ = float(rng.choice([3.0, 10.0, 40.0, 150.0, 600.0]))
            lights.append(vr.point_light(pos, float(rng.uniform(5, 200)), rng_, tuple(rng.uniform(0.1, 1.0, 3))))
        if it % 3 == 0:
            lights.insert(int(rng.integers(0, len(lights))), vr.point_light((0.0, 150.0, 0.0), 300.0, 0.0, (0.3, 0.3, 0.3)))    # no range limit
        hdr = vr.HdrImage(gpu_ctx, w, h)
        tiled.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
        full = hdr.download().copy()
        got = oracle.half_to_float(full).astype(np.float64)[..., :3]
        ref = oracle.deferred(v, gb, lights, AMBIENT_TOP, AMBIENT_BOTTOM, f32=True).astype(np.float64)[..., :3]
        assert np.isfinite(got).all(), it
        scale = max(1.0, float(ref.max()))
        rms = np.sqrt(np.mean(((got - ref) / scale) ** 2, axis=(0, 1)))
        assert (rms <= 1e-4).all(), (it, rms, scale)
        if done % 3 == 0:                                        # packed output of a 3-way split == unsplit
            world = 3
            info = partition_info(w, h, 0, world)
            gathered = np.zeros(world * info["packed_bytes"] // 2, np.uint16)
            for r in range(world):
                packed = vr.HdrImage(gpu_ctx, 128, info["max_owned"] * 128)
                tiled.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, packed, vr.Partition(r, world))
                gathered[r * info["packed_bytes"] // 2:(r + 1) * info["packed_bytes"] // 2] = packed.download(info["packed_bytes"])
                packed.close()
            big = vr.HdrImage(gpu_ctx, 128, world * info["max_owned"] * 128)
            big.upload(gathered)
            out = vr.HdrImage(gpu_ctx, w, h)
            frame_detile(gpu_ctx, big.device_ptr, world, out)
            assert np.array_equal(out.download(), full), it
            big.close(); out.close()
        hdr.close(); rt.close()
        done += 1
    tiled.Status()
    assert done >= 8


def test_large_heightmap_8192(oracle, gpu_ctx):
    """The largest surface the reference's LOD cap makes sense for (8192^2 texels, numLods = min(11, log2) = 11,
    QuadTree.cpp:10-17): 335 MB of textures + quad tables on the device, frames bit-exact vs the oracle."""
    size = 8192
    h = vr.synth_heightmap(gpu_ctx, size)
    a = vr.synth_albedo(gpu_ctx, size, h)
    p = params(size)
    tp = vr.TerrainPass(gpu_ctx, p).Init(h, a)
    ot = oracle.OracleTerrain(p, h, a)
    try:
        assert tp.GetNumLods() == ot.num_lods == 11
        sc = dict(ot=ot, tp=tp, size=size)
        for cam in (0, 5):
            eye, tgt = scaled_camera(CAMERAS[cam], size)
            v, gb_o, planes, n_o, n_g = _render_both(sc, oracle, gpu_ctx, eye, tgt, 640, 360)
            assert n_o == n_g and n_o > 0
            _assert_gbuffer_equal(gb_o, planes, f"8192 camera {cam}")
        # a camera close to the ground: deep LODs of the big tree
        hgt = float(h[size // 2 + 40, size // 2 - 60]) / 255.0 * 400.0
        v, gb_o, planes, n_o, n_g = _render_both(sc, oracle, gpu_ctx, (-60.0, hgt + 30.0, 40.0), (200.0, hgt - 40.0, -150.0), 640, 360)
        assert n_o == n_g and n_o > 50
        _assert_gbuffer_equal(gb_o, planes, "8192 close camera")
    finally:
        tp.close(); ot.close()


def test_random_texture_fuzz_bit_exact(oracle, gpu_ctx):
    """White-noise heightmap and albedo (every texel differs from its neighbours: steep slopes, every LOD level and
    clamp of the filter gets exercised), non-square textures of different sizes, 16 random views: bit-exact."""
    rng = np.random.default_rng(31337)
    for (hw, hh, aw, ah) in ((256, 256, 256, 256), (192, 320, 512, 128)):
        h = rng.integers(0, 256, size=(hh, hw), dtype=np.uint8)
        a = rng.integers(0, 256, size=(ah, aw, 4), dtype=np.uint8)
        p = params(256)
        ot = oracle.OracleTerrain(p, h, a)
        tp = vr.TerrainPass(gpu_ctx, p).Init(h, a)
        sc = dict(ot=ot, tp=tp, size=256, h=np.zeros((256, 256), np.uint8))
        try:
            for l in range(ot.height_levels()):
                assert np.array_equal(tp.download_mip("height", l), ot.height_mip(l))
            for l in range(ot.albedo_levels()):
                assert np.array_equal(tp.download_mip("albedo", l), ot.albedo_mip(l))
            for it, eye, tgt, up, fov, w, hgt in _fuzz_views(rng, 8, sc["h"], 256):
                v = vr.make_view(eye, tgt, w, hgt, vfov_deg=fov, up=up)
                rp = vr.default_render_params(400.0, assume_cleared=1)
                gb_o = oracle.GBufferHost(w, hgt)
                n_o = ot.render(v, gb_o, rp, None)
                rt = vr.RenderTargets(gpu_ctx).Init(w, hgt)
                tp.Render(v, v, rt, rp, None)
                planes = {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}
                assert tp.num_chunks() == n_o
                rt.close()
                _assert_gbuffer_equal(gb_o, planes, f"noise textures {hw}x{hh}/{aw}x{ah}, view {it}: eye {eye} target {tgt} fov {fov}")
        finally:
            tp.close(); ot.close()


def _soak_seeds(default):
    """The committed seed, plus VR_FUZZ_SEEDS=a,b,c for a soak run (one pytest process, more sequences)."""
    import os
    extra = [int(x) for x in os.environ.get("VR_FUZZ_SEEDS", "").split(",") if x.strip()]
    return [default] + extra


@pytest.mark.parametrize("seed", _soak_seeds(20261005))
def test_plane_tracking_sequence_fuzz(scene256, oracle, gpu_ctx, seed):
    """A random sequence of everything that touches a G-buffer - passes over a 'cleared' target, keep-what-is-there passes, Clear
    (lazy), uploads of junk into single planes, wireframe / depth-only / partitioned passes, lighting passes - with the planes
    compared only now and then, so that region states, the zero-emissive flag and a pending clear live through several
    operations.  The planes must equal an oracle-side mirror of the same sequence, and a lighting pass over the tracked target
    must equal, bit for bit, the same pass over a second target that holds the mirror's planes and knows nothing about them."""
    ot, tp = scene256["ot"], scene256["tp"]
    rng = np.random.default_rng(seed)
    w, h = 352, 200
    cams = [scaled_camera(c, 256) for c in CAMERAS[:6]] + [((10.0, 60.0, 10.0), (60.0, 200.0, 60.0))]      # the last one: sky only
    views = [vr.make_view(e, t, w, h) for e, t in cams]
    names = ("depth", "diffuse", "specular", "normals", "emissive")
    rp_k = vr.default_render_params(400.0)
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    rt_ref = vr.RenderTargets(gpu_ctx).Init(w, h)
    hdr, hdr_ref = vr.HdrImage(gpu_ctx, w, h), vr.HdrImage(gpu_ctx, w, h)
    dl = vr.DeferredLightingPass(gpu_ctx)
    sun = [vr.reference_sun()]
    cur = oracle.GBufferHost(w, h)
    ty, tx = np.indices((h, w))
    ops = ["cleared", "cleared", "cleared", "keep", "clear", "upload", "wire", "depth_only", "part", "light", "light_tiled", "submit", "lit", "check"]
    many = _scene_lights(scene256, 48)
    tdl = vr.TiledDeferredLightingPass(gpu_ctx)
    frame_call = vr.Frame(tp, rt, vr.default_render_params(400.0, assume_cleared=1), sun, AMBIENT_TOP, AMBIENT_BOTTOM)
    try:
        for step in range(160):
            op = str(rng.choice(ops))
            v = views[int(rng.integers(len(views)))]
            if op == "cleared":
                cur = oracle.GBufferHost(w, h); ot.render(v, cur, rp_k)
                tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1, depth_ranges=int(rng.integers(2))))
            elif op == "keep":
                ot.render(v, cur, rp_k); tp.Render(v, v, rt, rp_k)
            elif op == "clear":
                cur = oracle.GBufferHost(w, h); rt.Clear()
            elif op == "upload":
                k = names[int(rng.integers(5))]
                arr = getattr(cur, k)
                junk = rng.integers(0, 256, arr.shape + (arr.dtype.itemsize,), dtype=np.uint8).view(arr.dtype).reshape(arr.shape)
                if k == "depth":
                    junk = rng.random(arr.shape, dtype=np.float32)        # (finite depths)
                arr[...] = junk; rt.upload(k, junk)
            elif op == "wire":
                cur = oracle.GBufferHost(w, h); rpw = vr.default_render_params(400.0, wireframe=1)
                ot.render(v, cur, rpw); tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1, wireframe=1))
            elif op == "depth_only":
                rpd = vr.default_render_params(400.0, depth_only=1)
                ot.render(v, cur, rpd); tp.Render(v, v, rt, rpd)
            elif op == "part":
                world = int(rng.integers(2, 4)); rank = int(rng.integers(world))
                want = oracle.GBufferHost(w, h); ot.render(v, want, rp_k)
                owned = ((tx // 128 + ty // 128) % world) == rank
                for k in names:
                    getattr(cur, k)[owned] = getattr(want, k)[owned]
                tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1), vr.Partition(rank, world))
            elif op == "light":
                for k in names:
                    rt_ref.upload(k, getattr(cur, k))                    # (an upload: nothing is known about rt_ref's planes)
                dl.Render(v, rt, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
                dl.Render(v, rt_ref, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr_ref)
                a, b = hdr.download(), hdr_ref.download()
                assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"step {step}: lighting differs on {np.argwhere(a.view(np.uint16) != b.view(np.uint16))[:4].tolist()}"
            elif op == "light_tiled":                                     # the many-light pass takes the same hints
                for k in names:
                    rt_ref.upload(k, getattr(cur, k))
                tdl.Render(v, rt, many, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
                tdl.Render(v, rt_ref, many, AMBIENT_TOP, AMBIENT_BOTTOM, hdr_ref)
                assert np.array_equal(hdr.download().view(np.uint16), hdr_ref.download().view(np.uint16)), f"step {step}: tiled lighting differs"
            elif op == "submit":                                          # one call: tile pass over a 'cleared' target + lighting
                cur = oracle.GBufferHost(w, h); ot.render(v, cur, rp_k)
                frame_call.submit(v, hdr, [views[int(rng.integers(len(views)))]])
                for k in names:
                    rt_ref.upload(k, getattr(cur, k))
                dl.Render(v, rt_ref, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr_ref)
                assert np.array_equal(hdr.download().view(np.uint16), hdr_ref.download().view(np.uint16)), f"step {step}: vr_frame_submit's HdrColor differs"
            elif op == "lit":                                             # the fused variant: depth + HdrColor only, the other planes stay
                want = oracle.GBufferHost(w, h); ot.render(v, want, rp_k)
                tp.RenderLit(v, rt, vr.default_render_params(400.0, assume_cleared=1), sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr)
                for k in names:
                    rt_ref.upload(k, getattr(want, k))
                dl.Render(v, rt_ref, sun, AMBIENT_TOP, AMBIENT_BOTTOM, hdr_ref)
                assert np.array_equal(hdr.download().view(np.uint16), hdr_ref.download().view(np.uint16)), f"step {step}: fused HdrColor differs"
                cur.depth[...] = want.depth
            if op == "check" or step % 7 == 6:
                _assert_gbuffer_equal(cur, {k: rt.download(k) for k in names}, f"step {step} ({op})")
    finally:
        for o in (hdr, hdr_ref, rt, rt_ref):
            o.close()


@pytest.mark.parametrize("seed", _soak_seeds(424242))
def test_api_sequence_fuzz(oracle, gpu_ctx, seed):
    """A random sequence of the frame-loop calls (Render, Prepare for the same or other views - up to two frames ahead -,
    prepared frames consumed later, lock-view renders, a rank's partition with and without a prepared set, SetHeight toggles,
    stand-alone NodeSelect, a shadow-map render in between): every frame must still equal the oracle's.  Exercises the three
    rotating geometry sets and their streams, the prepared-geometry matching, the per-partition tables and the
    cross-stream dependencies.  The very first call on a fresh terrain is a Prepare with a partition."""
    rng = np.random.default_rng(seed)
    size = 256
    hmap = oracle.synth_heightmap(size)
    alb = oracle.synth_albedo(size, hmap)
    ot = oracle.OracleTerrain(params(size), hmap, alb)
    tp = vr.TerrainPass(gpu_ctx, params(size)).Init(hmap, alb)          # fresh: nothing cached for any partition yet
    w, h = 320, 180
    views = [vr.make_view(*scaled_camera(c, 256), w, h) for c in CAMERAS]
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    sm = vr.CascadedShadowMap(gpu_ctx, vr.default_shadow_params(256.0, resolution=256))
    heights = False
    can_lock = False
    ops = ["render", "render", "prepare_render", "prepare_other", "prepare_two", "lock", "part", "part_prepared", "height", "select", "shadow"]
    try:
        first = True
        for step in range(90):
            op = "part_prepared" if first else rng.choice(ops)
            first = False
            v = views[int(rng.integers(len(views)))]
            if op == "height":
                heights = not heights
                ot.set_height(heights); tp.SetHeight(heights)
                can_lock = False
                continue
            if op == "select":
                n_o, ids_o, inst_o = ot.select(v, 400.0)
                n_g, ids_g, inst_g = tp.NodeSelect(v, 400.0)
                assert n_g == n_o and np.array_equal(ids_g, ids_o) and np.array_equal(inst_g, inst_o), (step, op)
                can_lock = False                         # a stand-alone NodeSelect is not part of the reference's Render state
                continue
            if op == "shadow":                           # a depth-only render of another size between two frames (Renderer.cpp:333-372)
                lv = sm.SetupForPlanarViewStable(vr.reference_sun(), v)
                sm.Clear(); sm.RenderTerrain(tp)
                gb_l = oracle.GBufferHost(256, 256)
                ot.render(lv, gb_l, vr.default_render_params(400.0, depth_only=1))
                assert np.array_equal(sm.download_depth().view(np.uint32), gb_l.depth.view(np.uint32)), (step, op)
                can_lock = False
                continue
            part = vr.Partition(int(rng.integers(0, 3)), 3) if op in ("part", "part_prepared") else None
            lock = op == "lock" and can_lock
            # (depth_ranges: the tile pass variant that also leaves the light tiles' depth ranges - never consumed here, so its
            # clean / valid / dirty bookkeeping runs through every transition; the G-buffer must not change)
            rp = vr.default_render_params(400.0, assume_cleared=1, lock_view=int(lock), depth_ranges=int(step % 3 == 1))
            if op in ("prepare_render", "part_prepared"):
                tp.Prepare(v, rt, rp, part)
            elif op == "prepare_other":
                tp.Prepare(views[int(rng.integers(len(views)))], rt, rp, part)
            elif op == "prepare_two":                    # two frames ahead, in either order, one of them this frame's view
                other = views[int(rng.integers(len(views)))]
                for pv in ((v, other) if rng.integers(2) else (other, v)):
                    tp.Prepare(pv, rt, rp, part)
                tp.Prepare(v, rt, rp, part)              # naming a prepared frame again is a no-op
            gb_o = oracle.GBufferHost(w, h)
            n_o = ot.render(v, gb_o, rp, part)
            if part is not None:
                rt.Clear()                               # a rank leaves the tiles it does not own alone
            tp.Render(v, v, rt, rp, part)
            planes = {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}
            assert tp.num_chunks() == n_o, (step, op)
            _assert_gbuffer_equal(gb_o, planes, f"step {step} ({op}, heights {heights}, lock {lock})")
            can_lock = True
    finally:
        sm.close(); rt.close(); tp.close(); ot.close()


def test_shadow_fuzz(scene256, oracle, gpu_ctx):
    """Random sun directions, cameras, map resolutions and biases: the light view (host code vs restatement), the terrain
    shadow map (bit-exact) and the shadowed lighting pass (no flipped PCF comparison: max error at rounding level)."""
    import ctypes as C
    rng = np.random.default_rng(777)
    w, h = 320, 180
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    hdr = vr.HdrImage(gpu_ctx, w, h)
    dl = vr.DeferredLightingPass(gpu_ctx)
    for it in range(8):
        cam = vr.make_view(*scaled_camera(CAMERAS[int(rng.integers(len(CAMERAS)))], 256), w, h)
        d = rng.normal(size=3); d[1] = -abs(d[1]) - 0.15                      # the sun is above the horizon
        sun = vr.directional_light(tuple(d), 1.0, 0.53)
        sun.out_of_bounds_shadow = float(rng.choice([0.0, 1.0]))
        res = int(rng.choice([256, 384, 512]))
        bias = float(rng.choice([0.0, 0.001, 0.004]))
        sm = vr.CascadedShadowMap(gpu_ctx, vr.default_shadow_params(256.0, resolution=res, depth_bias=bias,
                                                                    max_shadow_distance=float(rng.choice([64.0, 256.0]))))
        lv = sm.SetupForPlanarViewStable(sun, cam)
        assert bytes(lv) == bytes(oracle.shadow_view(sun, cam, sm.params)), it
        sm.Clear(); sm.RenderTerrain(scene256["tp"])
        gb_l = oracle.GBufferHost(res, res)
        scene256["ot"].render(lv, gb_l, vr.default_render_params(400.0, depth_only=1))
        assert np.array_equal(sm.download_depth().view(np.uint32), gb_l.depth.view(np.uint32)), f"case {it}: shadow map"
        scene256["tp"].Render(cam, cam, rt, vr.default_render_params(400.0, assume_cleared=1))
        dl.Render(cam, rt, [sun], AMBIENT_TOP, AMBIENT_BOTTOM, hdr, shadow_map=sm)
        gb = oracle.GBufferHost(w, h)
        for name, arr in (("depth", gb.depth), ("diffuse", gb.diffuse), ("specular", gb.specular), ("normals", gb.normals), ("emissive", gb.emissive)):
            arr[...] = rt.download(name)
        want = oracle.deferred(cam, gb, [sun], AMBIENT_TOP, AMBIENT_BOTTOM, f32=True, shadow=(lv, gb_l.depth, 0, bias))
        err = np.abs(oracle.half_to_float(hdr.download()).astype(np.float64)[..., :3] - want[..., :3])
        assert err.max() < 2e-3 and np.sqrt((err ** 2).mean()) <= 1e-4, (it, err.max())
        sm.close()
    hdr.close(); rt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size,part", [((1920, 1080), None), ((7680, 4320), None), ((7680, 4320), (3, 8)), ((1000, 700), (1, 2))])
def test_tile_pass_launch_order_covers_every_tile_once_longest_bins_first(scene2048, gpu_ctx, size, part):
    """k_scan (workgroup-local scans, one atomic per workgroup and bin-length class): the tile pass's launch order must name
    every raster tile of the frame - or of this rank's share - exactly once, in classes of falling bin length, and the
    tiles' slices of the entry array must be disjoint and add up to the frame's entry count."""
    from vrenderer_amd.scene import flythrough_camera
    tp = scene2048["tp"]
    w, h = size
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    rp = vr.default_render_params(400.0, assume_cleared=1)
    v = vr.make_view(*flythrough_camera(40), w, h)
    p = vr.Partition(*part) if part else None
    tp.Render(v, v, rt, rp, p)
    tiles, lens = tp.tile_order()
    st = tp.render_stats()
    assert st["flags"] == 0
    world = part[1] if part else 1
    # the raster tile edge the library picks (vr_internal.h: vr_raster_tile_shift)
    tiles64 = ((w + 63) // 64) * ((h + 63) // 64)
    edge = 32                          # (round 4: at every size; tiles64 kept for the record)
    rtx, rty = (w + edge - 1) // edge, (h + edge - 1) // edge
    if part:
        sub = 128 // edge
        ty, tx = np.divmod(np.arange(rtx * rty), rtx)
        expect = np.nonzero(((tx // sub + ty // sub) % world) == part[0])[0]
    else:
        expect = np.arange(rtx * rty)
    assert len(tiles) == len(expect)
    assert np.array_equal(np.sort(tiles), expect), "launch order is not a permutation of the tiles to draw"
    cls = np.select([lens >= 256, lens >= 128, lens >= 64, lens >= 32, lens >= 16, lens >= 4, lens >= 1], [0, 1, 2, 3, 4, 5, 6], 7)
    assert (np.diff(cls) >= 0).all(), "bin-length classes out of order"
    assert int(lens.sum()) == st["bin_entries"]
    if part is None:                   # (render_stats walks every tile of the target; a rank's chain updates its own tiles' slices only)
        assert int(lens.max()) == st["max_bin"]
    rt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size,part", [((1920, 1080), None), ((7680, 4320), None), ((7680, 4320), (2, 3)), ((7680, 4320), (5, 8))])
def test_tiled_lighting_with_the_tile_pass_depth_ranges_equals_the_depth_re_read(scene2048, gpu_ctx, size, part):
    """vr_render_params::depth_ranges: the tile pass leaves every 32x32 light tile's depth range with the G-buffer and the
    culling stage of the tiled lighting pass takes it instead of reading the depth plane again.  Same minima and maxima ->
    same light lists -> the lit frame must be identical byte for byte (32- and 64-pixel raster tiles, whole frame and a
    rank's share); a second lighting call finds the ranges consumed and falls back to the depth plane; a clear in
    between drops them."""
    from vrenderer_amd.scene import flythrough_camera
    from vrenderer_amd.passes import partition_info
    tp = scene2048["tp"]
    w, h = size
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    v = vr.make_view(*flythrough_camera(55), w, h)
    lights = [vr.reference_sun()] + vr.synthetic_point_lights(511, 2048.0, scene2048["h"], 400.0, seed=9001)
    p = vr.Partition(*part) if part else None
    if part:
        info = partition_info(w, h, part[0], part[1])
        mk = lambda: vr.HdrImage(gpu_ctx, 128, info["max_owned"] * 128)
        dl = lambda im: im.download(info["packed_bytes"])
    else:
        mk = lambda: vr.HdrImage(gpu_ctx, w, h)
        dl = lambda im: im.download()
    tiled = vr.TiledDeferredLightingPass(gpu_ctx)
    out = {}
    for name, flag in (("plain", 0), ("ranges", 1)):
        img = mk()
        tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1, depth_ranges=flag), p)
        tiled.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, img, p)
        out[name] = dl(img)
        if flag:                       # the ranges are consumed: this call reads the depth plane
            tiled.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, img, p)
            out["again"] = dl(img)
        img.close()
    tiled.Status()
    assert np.array_equal(out["plain"], out["ranges"])
    assert np.array_equal(out["plain"], out["again"])
    # stale ranges are never used: render with ranges, clear, render another view without, light
    img = mk()
    tp.Render(v, v, rt, vr.default_render_params(400.0, assume_cleared=1, depth_ranges=1), p)
    rt.Clear()
    v2 = vr.make_view(*flythrough_camera(90), w, h)
    tp.Render(v2, v2, rt, vr.default_render_params(400.0), p)
    tiled.Render(v2, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, img, p)
    a = dl(img)
    tp.Render(v2, v2, rt, vr.default_render_params(400.0, assume_cleared=1, depth_ranges=1), p)      # (and fresh ones after stale ones)
    tiled.Render(v2, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, img, p)
    b = dl(img)
    assert np.array_equal(a, b)
    img.close(); rt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("edge", [64, 32])
def test_split_of_the_8k_frame_reassembles_to_the_unsplit_frame_on_either_tile_size(scene2048, gpu_ctx, edge):
    """The raster tile edge follows the frame size and the split (32 pixels up to ~9.6K x 5.4K, vr_internal.h); either can be
    pinned (VR_OPT_RASTER_TILE).  With each pinned in turn: the packed lit tiles of a 3-way split of the 8K frame, de-tiled,
    must equal the unsplit frame (rendered on the OTHER tile size) byte for byte, and so must every rank's G-buffer planes on
    its own tiles - candidate lists in the scan, owner-tile launch order, packed lighting of both variants."""
    from vrenderer_amd.scene import flythrough_camera
    from vrenderer_amd.passes import frame_detile, partition_info
    tp = scene2048["tp"]
    W, H, world = 7680, 4320, 3
    v = vr.make_view(*flythrough_camera(17), W, H)
    rt = vr.RenderTargets(gpu_ctx).Init(W, H)
    rp = vr.default_render_params(400.0, assume_cleared=1)
    lights = [vr.reference_sun()]
    dl = vr.DeferredLightingPass(gpu_ctx)
    try:
        gpu_ctx.set_raster_tile(96 - edge)                                    # the unsplit reference on the other tile size
        full = vr.HdrImage(gpu_ctx, W, H)
        tp.Render(v, v, rt, rp)
        tiles, _ = tp.tile_order()
        assert len(tiles) == ((W + 95 - edge) // (96 - edge)) * ((H + 95 - edge) // (96 - edge))
        dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, full)
        ref = full.download()
        ref_depth = rt.download("depth").view(np.uint32)
        ref_nrm = rt.download("normals")
        full.close()
        gpu_ctx.set_raster_tile(edge)
        info = partition_info(W, H, 0, world)
        gathered = np.zeros(world * info["packed_bytes"] // 2, np.uint16)
        oh, ow = (H + 127) // 128, W // 128
        ty, tx = np.divmod(np.arange(oh * ow), ow)
        pad = oh * 128 - H
        def owner_tiles(a):
            a = np.pad(a, ((0, pad),) + ((0, 0),) * (a.ndim - 1))
            return a.reshape((oh, 128, ow, 128) + a.shape[2:]).swapaxes(1, 2)
        for r in range(world):
            part = vr.Partition(r, world)
            packed = vr.HdrImage(gpu_ctx, 128, info["max_owned"] * 128)
            rt.Clear()
            tp.Render(v, v, rt, rp, part)
            t_r, _ = tp.tile_order()
            assert 0 < len(t_r) < len(tiles) * (96 - edge) ** 2 // edge ** 2          # a share of the tiles of this size
            dl.Render(v, rt, lights, AMBIENT_TOP, AMBIENT_BOTTOM, packed, part)
            gathered[r * info["packed_bytes"] // 2:(r + 1) * info["packed_bytes"] // 2] = packed.download(info["packed_bytes"])
            packed.close()
            own = ((tx + ty) % world == r).reshape(oh, ow)
            assert np.array_equal(owner_tiles(rt.download("depth").view(np.uint32))[own], owner_tiles(ref_depth)[own])
            assert np.array_equal(owner_tiles(rt.download("normals"))[own], owner_tiles(ref_nrm)[own])
        big = vr.HdrImage(gpu_ctx, 128, world * info["max_owned"] * 128)
        big.upload(gathered)
        out = vr.HdrImage(gpu_ctx, W, H)
        frame_detile(gpu_ctx, big.device_ptr, world, out)
        assert np.array_equal(out.download(), ref)
        big.close(); out.close()
    finally:
        gpu_ctx.set_raster_tile(0)
        rt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("edge", [32, 64])
@pytest.mark.parametrize("size,part", [((1280, 720), None), ((1000, 700), None), ((1000, 700), (1, 3))])
def test_both_tile_sizes_bit_exact_against_the_oracle(scene2048, oracle, gpu_ctx, edge, size, part):
    """Every G-buffer plane against the oracle with the raster tile edge pinned to 32 and to 64 pixels (VR_OPT_RASTER_TILE), whole
    frame and a rank's share, at sizes the oracle renders in a fraction of a second - the frame size picks only one of the two
    variants, and since round 3's re-measured rule the 64-pixel one only beyond ~9.6K x 5.4K."""
    from vrenderer_amd.scene import flythrough_camera
    w, h = size
    v = vr.make_view(*flythrough_camera(83), w, h)
    rp = vr.default_render_params(400.0, assume_cleared=1)
    p = vr.Partition(*part) if part else None
    rt = vr.RenderTargets(gpu_ctx).Init(w, h)
    try:
        gpu_ctx.set_raster_tile(96 - edge)
        scene2048["tp"].Prepare(v, rt, rp, p)          # geometry prepared for the OTHER tile size must not be taken for this frame
        gpu_ctx.set_raster_tile(edge)
        gb_o = oracle.GBufferHost(w, h)
        n_o = scene2048["ot"].render(v, gb_o, rp, p)
        rt.Clear()
        scene2048["tp"].Render(v, v, rt, rp, p)
        assert scene2048["tp"].num_chunks() == n_o
        tiles, _ = scene2048["tp"].tile_order()
        if part is None:
            assert len(tiles) == ((w + edge - 1) // edge) * ((h + edge - 1) // edge)
        planes = {k: rt.download(k) for k in ("depth", "diffuse", "specular", "normals", "emissive")}
        _assert_gbuffer_equal(gb_o, planes, f"{w}x{h}, {edge}-pixel tiles, partition {part}")
    finally:
        gpu_ctx.set_raster_tile(0)
        rt.close()
