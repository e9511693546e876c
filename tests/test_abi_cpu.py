"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, its pure-host entry points behave, and it fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import vrenderer_amd as vr
from vrenderer_amd import capi
from vrenderer_amd import partition as pt
from tests.common import DEFAULT_EYE, DEFAULT_TARGET

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu(lib):
    h = C.c_void_p()
    rc = lib.vr_context_create(0, C.byref(h))
    if rc == capi.VR_OK:
        lib.vr_context_destroy(h)
        return True
    return False


def test_library_exports_every_declared_symbol(product_lib):
    header = open(os.path.join(ROOT, "include", "vrterrain.h")).read()
    declared = re.findall(r"VR_API\s+[\w\s\*]+?\b(vr_\w+)\s*\(", header)
    assert len(declared) >= 30
    assert sorted(set(declared)) == sorted(set(capi.EXPORTS)), "capi.EXPORTS and include/vrterrain.h disagree"
    for name in declared:
        assert hasattr(product_lib, name), name


def test_struct_layouts_match_header():
    assert C.sizeof(capi.Instance) == 112 and C.sizeof(capi.Light) == 64
    assert C.sizeof(capi.TerrainParams) == 40 and C.sizeof(capi.RenderParams) == 32
    assert C.sizeof(capi.View) == 16 * 4 * 4 + 16 + 96 + 8 * 4
    assert C.sizeof(capi.Partition) == 8


def test_defaults_are_the_reference_settings(product_lib):
    p = vr.default_terrain_params()
    # TerrainSettings (TerrainPass.h:23-30), minLodDistance (QuadTree.cpp:236), morph start (terrain_vs.hlsl:20)
    assert (p.max_instances, p.surface_size, p.world_size, p.grid_size) == (4096, 2048.0, 2048.0, 32)
    assert p.min_lod_distance == 4.0 and abs(p.morph_start - 0.85) < 1e-7
    rp = vr.default_render_params()
    assert rp.max_height == 400.0 and not (rp.wireframe or rp.lock_view or rp.depth_only)   # Renderer.h:37-40


def test_shadow_and_tonemap_defaults_and_argument_checks(product_lib, oracle):
    assert C.sizeof(capi.ShadowParams) == 32 and C.sizeof(capi.TonemapParams) == 40 and C.sizeof(capi.ShadowBinding) == 24
    sp = vr.default_shadow_params(2048.0)                      # CascadedShadowMap(device, 2048, 1, 0, fmt), WORLD_SIZE x3 (Renderer.cpp:83,349-352)
    assert (sp.resolution, sp.max_shadow_distance, sp.light_space_z_up, sp.light_space_z_down, sp.depth_bias) == (2048, 2048.0, 2048.0, 2048.0, 0.0)
    tm = vr.default_tonemap_params()                           # ToneMappingParameters() defaults
    assert [round(getattr(tm, n), 4) for n, _ in capi.TonemapParams._fields_] == [0.8, 0.95, 1.0, 0.5, 0.02, 0.5, -0.5, 3.0, -10.0, 4.0]
    cam = vr.make_view(DEFAULT_EYE, DEFAULT_TARGET, 1920, 1080)
    sun, out = vr.reference_sun(), vr.View()
    assert product_lib.vr_shadow_view_setup(C.byref(sun), C.byref(cam), C.byref(sp), C.byref(out)) == capi.VR_OK
    assert bytes(out) == bytes(oracle.shadow_view(sun, cam, sp))                     # host code == restatement, every field
    assert out.viewport_w == out.viewport_h == 2048 and out.view_to_clip[15] == 1.0 and out.view_to_clip[11] == 0.0   # orthographic
    lamp = vr.point_light((0, 10, 0), 100.0, 50.0)
    assert product_lib.vr_shadow_view_setup(C.byref(lamp), C.byref(cam), C.byref(sp), C.byref(out)) == capi.VR_ERR_INVALID_ARGUMENT
    assert b"directional" in product_lib.vr_last_error()
    bad = vr.default_shadow_params(2048.0, resolution=0)
    assert product_lib.vr_shadow_view_setup(C.byref(sun), C.byref(cam), C.byref(bad), C.byref(out)) == capi.VR_ERR_INVALID_ARGUMENT
    assert product_lib.vr_shadow_view_setup(None, C.byref(cam), C.byref(sp), C.byref(out)) == capi.VR_ERR_INVALID_ARGUMENT
    assert product_lib.vr_partition_packed_bytes_ldr(7680, 4320, 8) * 2 == product_lib.vr_partition_packed_bytes(7680, 4320, 8)


def test_view_helper_matches_oracle_bit_for_bit(product_lib, oracle):
    for (w, h) in ((1920, 1080), (7680, 4320), (333, 777)):
        for eye, tgt in ((DEFAULT_EYE, DEFAULT_TARGET), ((600.0, 250.0, 0.0), (0.0, 0.0, 0.0)), ((3.0, 9.0, -4.0), (100.0, 0.0, 50.0))):
            assert bytes(vr.make_view(eye, tgt, w, h)) == bytes(oracle.view_from_camera(eye, tgt, w, h))


def test_view_helper_rejects_bad_arguments(product_lib):
    v = vr.View()
    f3 = (C.c_float * 3)(0, 0, 0)
    assert product_lib.vr_view_from_camera(f3, f3, f3, 1.0, 0.1, 100.0, 0, 10, C.byref(v)) == capi.VR_ERR_INVALID_ARGUMENT
    assert b"projection" in product_lib.vr_last_error()
    assert product_lib.vr_view_from_camera(f3, f3, f3, 1.0, 1.0, 0.5, 10, 10, C.byref(v)) == capi.VR_ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("w,h,world", [(7680, 4320, 8), (7680, 4320, 1), (3840, 2160, 4), (1920, 1080, 2), (300, 260, 3), (64, 64, 8)])
def test_partition_tables_agree_with_python_mirror(product_lib, w, h, world):
    tx, ty = pt.owner_grid(w, h)
    counts = []
    for r in range(world):
        info = vr.passes.partition_info(w, h, r, world)
        assert (info["tiles_x"], info["tiles_y"]) == (tx, ty)
        assert info["owned"] == len(pt.owned_tiles(w, h, r, world))
        assert info["max_owned"] == pt.max_owned(w, h, world)
        assert info["packed_bytes"] == pt.max_owned(w, h, world) * 128 * 128 * 6
        counts.append(info["owned"])
    assert sum(counts) == tx * ty
    if (w, h, world) == (7680, 4320, 8):
        # 60 x 34 = 2,040 owner tiles, ~255 per rank (the diagonal interleave is balanced to within 2 tiles)
        assert (tx, ty) == (60, 34) and max(counts) - min(counts) <= 2 and max(counts) == 256
    slots = pt.tile_slots(w, h, world)
    assert len(set(slots.tolist())) == tx * ty


def test_fails_loudly_without_a_gpu(product_lib):
    if _has_gpu(product_lib):
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert product_lib.vr_context_create(0, C.byref(h)) == capi.VR_ERR_NO_DEVICE
    assert b"no HIP device" in product_lib.vr_last_error()
    with pytest.raises(vr.VrError):
        vr.Context(0)


def test_missing_extension_raises(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load_library()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under vrenderer_amd/ may reference it."""
    pkg = os.path.join(ROOT, "vrenderer_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "vr_oracle" not in text and "pyoracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)


def test_product_build_has_no_experiment_switch():
    """Timing experiments that render wrong images on purpose (VR_EXP_*) and instrumented kernels live behind
    csrc/vr_experiments.h and -DVR_EXPERIMENT_BUILD: the library the tests and the bench load was built with none of them,
    the kernel sources test constexpr flags instead of the preprocessor, and the product's compiler flags define nothing."""
    lib = capi.load_library()
    assert lib.vr_build_experiments() == 0
    csrc = os.path.join(ROOT, "vrenderer_amd", "csrc")
    for f in os.listdir(csrc):
        text = open(os.path.join(csrc, f), errors="ignore").read()
        if f == "vr_experiments.h":
            assert "#error" in text and "VR_EXPERIMENT_BUILD" in text
            continue
        assert "VR_EXP_" not in text and "VR_LDS_PAD" not in text, f"{f}: experiment switches belong in vr_experiments.h"
        if f.endswith(".hip") and ("VR_RASTER_PROFILE" in text or "VR_SELECT_PROFILE" in text):
            assert '#include "vr_experiments.h"' in text, f"{f}: profiling switches are guarded by vr_experiments.h"
    from vrenderer_amd import build as b
    flags = list(b.FLAGS) + [x for v in b.PER_FILE_FLAGS.values() for x in v]
    assert not any(x.startswith("-DVR_") for x in flags), flags


def test_pack_detile_roundtrip_numpy():
    rng = np.random.default_rng(3)
    for (w, h, world) in ((300, 260, 3), (512, 256, 2), (130, 70, 8)):
        frame = rng.integers(0, 65535, (h, w, 4), dtype=np.uint16)
        frame[..., 3] = 0                                  # HdrColor's alpha is 0 on this path and is not exchanged
        gathered = np.concatenate([pt.pack(frame, r, world) for r in range(world)], 0)
        assert np.array_equal(pt.detile(gathered, w, h, world), frame)


def _build_host_example(tmpdir):
    import subprocess
    exe = os.path.join(str(tmpdir), "frame_example")
    lib_dir = os.path.join(ROOT, "vrenderer_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "host", "frame_example.cpp"), "-o", exe,
           "-L", lib_dir, "-lvrterrain", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_cpp_host_example_compiles_links_and_fails_soft_without_gpu(product_lib, tmp_path):
    """The header is usable from C++ and the library links; without a device the example reports it."""
    import subprocess
    exe = _build_host_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if _has_gpu(product_lib):
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 0 and "no device" in r.stdout and "no HIP device" in r.stderr, r.stdout + r.stderr


def _build_allgather_example(tmpdir):
    import subprocess
    exe = os.path.join(str(tmpdir), "frame_allgather_example")
    lib_dir = os.path.join(ROOT, "vrenderer_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "host", "frame_allgather_example.cpp"), "-o", exe,
           "-L", lib_dir, "-lvrterrain", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_cpp_allgather_example_compiles_and_fails_soft_without_gpu(product_lib, tmp_path):
    """The N-rank C++ host (RCCL communicator + vr_frame_allgather_ldr + vr_tonemap_allreduce_histogram) builds against
    rccl.h and the C ABI; without a device it says so."""
    import subprocess
    exe = _build_allgather_example(tmp_path)
    if _has_gpu(product_lib):
        pytest.skip("run by the GPU suite")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "no device" in r.stdout, r.stdout + r.stderr


def test_library_does_not_link_rccl(product_lib):
    """RCCL is resolved at first use from the copy in the process (vr_comm.hip): no DT_NEEDED entry for it."""
    import subprocess
    out = subprocess.run(["readelf", "-d", os.path.join(ROOT, "vrenderer_amd", "lib", "libvrterrain.so")], capture_output=True, text=True).stdout
    assert "rccl" not in out.lower(), out


def test_header_is_valid_c(tmp_path):
    import subprocess
    src = tmp_path / "c_check.c"
    src.write_text('#include <vrterrain.h>\nint main(void) { vr_view v; vr_light l; (void)v; (void)l; return sizeof(vr_instance) == 112 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                    str(tmp_path / "c_check")], check=True, capture_output=True, text=True)
    assert subprocess.run([str(tmp_path / "c_check")]).returncode == 0


def test_ctypes_mirrors_have_the_header_s_struct_sizes(tmp_path):
    """The ctypes structures of vrenderer_amd/capi.py against sizeof() of the C declarations they mirror."""
    import subprocess
    from vrenderer_amd import capi
    pairs = [("vr_view", capi.View), ("vr_light", capi.Light), ("vr_instance", capi.Instance), ("vr_terrain_params", capi.TerrainParams),
             ("vr_render_params", capi.RenderParams), ("vr_partition", capi.Partition), ("vr_shadow_params", capi.ShadowParams),
             ("vr_shadow_binding", capi.ShadowBinding), ("vr_tonemap_params", capi.TonemapParams), ("vr_frame_desc", capi.FrameDesc),
             ("vr_gbuffer_desc", capi.GBufferDesc)]
    src = tmp_path / "sizes.c"
    src.write_text("#include <stdio.h>\n#include <vrterrain.h>\nint main(void) {\n"
                   + "".join(f'    printf("{n} %zu\\n", sizeof({n}));\n' for n, _ in pairs) + "    return 0;\n}\n")
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "sizes")], check=True,
                   capture_output=True, text=True)
    out = dict(l.split() for l in subprocess.run([str(tmp_path / "sizes")], capture_output=True, text=True, check=True).stdout.splitlines())
    import ctypes as C
    for n, t in pairs:
        assert int(out[n]) == C.sizeof(t), (n, out[n], C.sizeof(t))


def test_png_ingest_round_trip(tmp_path, oracle):
    """Row f4: PNG heightmap / albedo ingest gives back exactly the bytes TerrainPass.Init takes."""
    from vrenderer_amd import io as vio
    h = oracle.synth_heightmap(64)
    a = oracle.synth_albedo(64, h)
    vio.save_png(str(tmp_path / "h.png"), h)
    vio.save_png(str(tmp_path / "a.png"), a)
    assert np.array_equal(vio.load_heightmap_png(str(tmp_path / "h.png")), h)
    assert np.array_equal(vio.load_albedo_png(str(tmp_path / "a.png")), a)
    vio.save_png(str(tmp_path / "rgb.png"), a[..., :3])
    assert np.array_equal(vio.load_heightmap_png(str(tmp_path / "rgb.png")), a[..., 0])
