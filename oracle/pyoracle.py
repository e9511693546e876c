"""ctypes binding of the CPU oracle (oracle/vr_oracle.c).

TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (see oracle/vr_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (vrenderer_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from vrenderer_amd.capi import (Instance, Light, Partition, RenderParams, ShadowParams, TerrainParams, TonemapParams, View, VR_MAX_LODS,
                                VR_TONEMAP_BINS)

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "_build", "libvroracle.so")
_lib = None


def _host_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            return any(" fma " in (line + " ") for line in f if line.startswith("flags"))
    except OSError:
        return True


def build(force=False):
    """Builds the oracle library (with -mfma: the model's fmaf() calls become one instruction).  On a host without FMA
    hardware the same source is built without the flag into its own file - libm's fmaf gives the identical values."""
    global LIB_PATH
    if os.environ.get("VR_ORACLE_LIB"):           # an instrumented build of the same source (oracle/Makefile: asan), already built
        LIB_PATH = os.environ["VR_ORACLE_LIB"]
        return LIB_PATH
    src = os.path.join(_DIR, "vr_oracle.c")
    nofma = not _host_has_fma()
    if nofma:
        LIB_PATH = os.path.join(_DIR, "_build", "libvroracle_nofma.so")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _DIR, "-B" if force else "-s"] + (["NOFMA=1"] if nofma else []), check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) or not _host_has_fma() or os.environ.get("VR_ORACLE_LIB"):
        build()
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp = C.c_void_p
    L.orc_terrain_create.restype = vp
    L.orc_terrain_create.argtypes = [P(TerrainParams), vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.orc_terrain_destroy.argtypes = [vp]
    L.orc_terrain_num_lods.argtypes = [vp]
    L.orc_terrain_lod_ranges.argtypes = [vp, P(C.c_float)]
    L.orc_terrain_num_nodes.restype = C.c_long
    L.orc_terrain_num_nodes.argtypes = [vp]
    L.orc_terrain_height_levels.argtypes = [vp]
    L.orc_terrain_albedo_levels.argtypes = [vp]
    L.orc_terrain_height_mip.restype = vp
    L.orc_terrain_height_mip.argtypes = [vp, C.c_int, P(C.c_int), P(C.c_int)]
    L.orc_terrain_albedo_mip.restype = vp
    L.orc_terrain_albedo_mip.argtypes = [vp, C.c_int, P(C.c_int), P(C.c_int)]
    L.orc_select.argtypes = [vp, P(View), C.c_float, C.c_int, vp, vp, C.c_int]
    L.orc_set_height.argtypes = [vp]
    L.orc_node_height.argtypes = [vp, C.c_uint32, P(C.c_float), P(C.c_float)]
    L.orc_set_height_loaded.argtypes = [vp, C.c_int]
    L.orc_reset_height.argtypes = [vp]
    L.orc_node_heights.argtypes = [vp, vp, C.c_long]
    L.orc_view_from_camera.argtypes = [P(C.c_float), P(C.c_float), P(C.c_float), C.c_float, C.c_float, C.c_float,
                                       C.c_int, C.c_int, P(View)]
    L.orc_vertex.argtypes = [vp, P(View), C.c_float, P(Instance), C.c_int, C.c_int, P(C.c_float), P(C.c_float)]
    L.orc_gbuffer_clear.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.orc_render.argtypes = [vp, P(View), P(RenderParams), P(Partition), C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.orc_deferred.argtypes = [P(View), C.c_int, C.c_int, vp, vp, vp, vp, vp, P(Light), C.c_int,
                               P(C.c_float), P(C.c_float), vp]
    L.orc_deferred_f32.argtypes = L.orc_deferred.argtypes
    L.orc_debug_set_pixel_plane.argtypes = [vp]
    L.orc_debug_set_pixel_plane.restype = None
    L.orc_model_revision.restype = C.c_int
    L.orc_synth_heightmap.argtypes = [C.c_int, C.c_uint32, vp]
    L.orc_synth_albedo.argtypes = [C.c_int, C.c_uint32, vp, vp]
    L.orc_half_to_float.restype = C.c_float
    L.orc_half_to_float.argtypes = [C.c_uint16]
    L.orc_float_to_half.restype = C.c_uint16
    L.orc_float_to_half.argtypes = [C.c_float]
    L.orc_srgb8_to_linear.restype = C.c_float
    L.orc_srgb8_to_linear.argtypes = [C.c_uint8]
    L.orc_linear_to_srgb8.restype = C.c_uint8
    L.orc_linear_to_srgb8.argtypes = [C.c_float]
    L.orc_linear_to_srgb8_batch.argtypes = [vp, C.c_size_t, vp]
    L.orc_shadow_view.argtypes = [P(Light), P(View), P(ShadowParams), P(View)]
    L.orc_deferred_set_shadow.argtypes = [P(View), vp, C.c_int, C.c_int, C.c_float]
    L.orc_tonemap_histogram.argtypes = [P(TonemapParams), vp, C.c_int, C.c_int, P(Partition), vp]
    L.orc_tonemap_exposure.restype = C.c_float
    L.orc_tonemap_exposure.argtypes = [P(TonemapParams), vp, C.c_float, C.c_float]
    L.orc_tonemap_apply.argtypes = [P(TonemapParams), C.c_float, vp, C.c_int, C.c_int, vp]
    L.orc_log2_pinned.restype = C.c_float
    L.orc_log2_pinned.argtypes = [C.c_float]
    L.orc_exp2_pinned.restype = C.c_float
    L.orc_exp2_pinned.argtypes = [C.c_float]
    L.orc_time_tree_build.restype = C.c_double
    L.orc_time_tree_build.argtypes = [P(TerrainParams), vp, C.c_int, C.c_int]
    L.orc_time_select.restype = C.c_double
    L.orc_time_select.argtypes = [vp, P(View), C.c_int, C.c_float, C.c_int, P(C.c_int)]
    L.orc_time_set_height.restype = C.c_double
    L.orc_time_set_height.argtypes = [vp]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def synth_heightmap(size, seed=1337):
    out = np.empty((size, size), np.uint8)
    lib().orc_synth_heightmap(size, seed, _ptr(out))
    return out


def synth_albedo(size, height, seed=4242):
    out = np.empty((size, size, 4), np.uint8)
    h = np.ascontiguousarray(height, np.uint8)
    lib().orc_synth_albedo(size, seed, _ptr(h), _ptr(out))
    return out


def view_from_camera(eye, target, w, h, vfov_deg=60.0, z_near=0.1, z_far=10000.0, up=(0, 1, 0)):
    v = View()
    lib().orc_view_from_camera(_f3(eye), _f3(target), _f3(up), np.float32(np.radians(np.float32(vfov_deg))),
                               z_near, z_far, w, h, C.byref(v))
    return v


class GBufferHost:
    """Host G-buffer planes in the layout of vr_gbuffer_desc."""

    def __init__(self, w, h):
        self.w, self.h = w, h
        self.depth = np.empty((h, w), np.float32)
        self.diffuse = np.empty((h, w), np.uint32)
        self.specular = np.empty((h, w), np.uint32)
        self.normals = np.empty((h, w, 4), np.uint16)
        self.emissive = np.empty((h, w, 4), np.uint16)
        self.clear()

    def clear(self):
        lib().orc_gbuffer_clear(self.w, self.h, _ptr(self.depth), _ptr(self.diffuse), _ptr(self.specular),
                                _ptr(self.normals), _ptr(self.emissive))

    def planes(self):
        return [self.depth, self.diffuse, self.specular, self.normals, self.emissive]


class OracleTerrain:
    def __init__(self, params, height, albedo):
        self.params = params
        self._h = np.ascontiguousarray(height, np.uint8)
        self._a = np.ascontiguousarray(albedo, np.uint8)
        self.handle = lib().orc_terrain_create(C.byref(params), _ptr(self._h), self._h.shape[1], self._h.shape[0],
                                               _ptr(self._a), self._a.shape[1], self._a.shape[0])

    def close(self):
        if self.handle:
            lib().orc_terrain_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    @property
    def num_lods(self):
        return lib().orc_terrain_num_lods(self.handle)

    @property
    def num_nodes(self):
        return lib().orc_terrain_num_nodes(self.handle)

    def lod_ranges(self):
        out = (C.c_float * VR_MAX_LODS)()
        lib().orc_terrain_lod_ranges(self.handle, out)
        return np.array(out[:], np.float32)

    def height_mip(self, level):
        w, h = C.c_int(), C.c_int()
        p = lib().orc_terrain_height_mip(self.handle, level, C.byref(w), C.byref(h))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value)).copy()

    def albedo_mip(self, level):
        w, h = C.c_int(), C.c_int()
        p = lib().orc_terrain_albedo_mip(self.handle, level, C.byref(w), C.byref(h))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value, 4)).copy()

    def set_height(self, loaded=True):
        """QuadTree::SetHeight over the whole tree + m_HeightLoaded (QuadTree.cpp:46-51,191-208)."""
        if loaded:
            lib().orc_set_height(self.handle)
            lib().orc_set_height_loaded(self.handle, 1)
        else:
            lib().orc_reset_height(self.handle)       # what vr_terrain_update_heights(t, 0) stands for

    def node_heights(self):
        out = np.zeros((self.num_nodes, 2), np.float32)
        lib().orc_node_heights(self.handle, _ptr(out), self.num_nodes)
        return out

    def height_levels(self):
        return lib().orc_terrain_height_levels(self.handle)

    def albedo_levels(self):
        return lib().orc_terrain_albedo_levels(self.handle)

    def select(self, view, max_height=400.0, stub_frustum=False, capacity=None):
        cap = capacity or self.params.max_instances
        ids = np.zeros(cap, np.uint32)
        inst = (Instance * cap)()
        n = lib().orc_select(self.handle, C.byref(view), max_height, int(stub_frustum), _ptr(ids), inst, cap)
        m = min(n, cap)
        return n, ids[:m].copy(), np.frombuffer(inst, dtype=np.uint8).reshape(cap, 112)[:m].copy()

    def vertex(self, view, max_height, inst_bytes, vx, vz):
        inst = Instance.from_buffer_copy(bytes(inst_bytes))
        clip = (C.c_float * 4)()
        world = (C.c_float * 3)()
        lib().orc_vertex(self.handle, C.byref(view), max_height, C.byref(inst), vx, vz, clip, world)
        return np.array(clip[:], np.float32), np.array(world[:], np.float32)

    def render(self, view, gb, rp, part=None, debug_plane=None):
        """debug_plane: optional (h, w, 3) float32 array that receives, per shaded pixel, main_ps's implicit LOD (before the
        sampler's clamp) and the interpolated world x, z (orc_debug_set_pixel_plane; not part of the model)."""
        if debug_plane is not None:
            assert debug_plane.dtype == np.float32 and debug_plane.shape == (gb.h, gb.w, 3) and debug_plane.flags.c_contiguous
            lib().orc_debug_set_pixel_plane(_ptr(debug_plane))
        try:
            return lib().orc_render(self.handle, C.byref(view), C.byref(rp), C.byref(part) if part is not None else None,
                                    gb.w, gb.h, _ptr(gb.depth), _ptr(gb.diffuse), _ptr(gb.specular),
                                    _ptr(gb.normals), _ptr(gb.emissive))
        finally:
            if debug_plane is not None:
                lib().orc_debug_set_pixel_plane(None)


def model_revision():
    """Revision of the oracle's raster / sampler model (frozen from round 4 on; tests/golden/MODEL_REVISIONS.txt)."""
    return lib().orc_model_revision()


def shadow_view(light, camera_view, params):
    v = View()
    lib().orc_shadow_view(C.byref(light), C.byref(camera_view), C.byref(params), C.byref(v))
    return v


def deferred(view, gb, lights, amb_top, amb_bottom, f32=False, shadow=None):
    """shadow = (light_view, depth array res x res float32, light_index, depth_bias) or None."""
    if shadow is not None:
        lv, sd, li, bias = shadow
        sd = np.ascontiguousarray(sd, np.float32)
        lib().orc_deferred_set_shadow(C.byref(lv), _ptr(sd), sd.shape[0], li, bias)
    try:
        return _deferred(view, gb, lights, amb_top, amb_bottom, f32)
    finally:
        lib().orc_deferred_set_shadow(None, None, 0, 0, 0.0)


def _deferred(view, gb, lights, amb_top, amb_bottom, f32=False):
    n = len(lights)
    arr = (Light * max(n, 1))(*lights)
    if f32:
        out = np.empty((gb.h, gb.w, 4), np.float32)
        fn = lib().orc_deferred_f32
    else:
        out = np.empty((gb.h, gb.w, 4), np.uint16)
        fn = lib().orc_deferred
    fn(C.byref(view), gb.w, gb.h, _ptr(gb.depth), _ptr(gb.diffuse), _ptr(gb.specular), _ptr(gb.normals),
       _ptr(gb.emissive), arr, n, _f3(amb_top), _f3(amb_bottom), _ptr(out))
    return out


class ToneMapper:
    """ToneMappingPass state (histogram + adapted luminance) on the host; same call sequence as the product's."""

    def __init__(self):
        self.hist = np.zeros(VR_TONEMAP_BINS, np.uint32)
        self.adapted = 0.0
        self.frame_time = 0.0

    def AdvanceFrame(self, seconds):
        self.frame_time = float(seconds)

    def ResetHistogram(self):
        self.hist[:] = 0

    def AddFrameToHistogram(self, params, hdr_u16, part=None):
        a = np.ascontiguousarray(hdr_u16, np.uint16)
        lib().orc_tonemap_histogram(C.byref(params), _ptr(a), a.shape[1], a.shape[0],
                                    C.byref(part) if part is not None else None, _ptr(self.hist))

    def ComputeExposure(self, params):
        self.adapted = float(lib().orc_tonemap_exposure(C.byref(params), _ptr(self.hist), self.frame_time, self.adapted))

    def Render(self, params, hdr_u16):
        a = np.ascontiguousarray(hdr_u16, np.uint16)
        out = np.empty((a.shape[0], a.shape[1], 4), np.uint8)
        lib().orc_tonemap_apply(C.byref(params), self.adapted, _ptr(a), a.shape[1], a.shape[0], _ptr(out))
        return out

    def SimpleRender(self, params, hdr_u16):
        self.ResetHistogram()
        self.AddFrameToHistogram(params, hdr_u16)
        self.ComputeExposure(params)
        return self.Render(params, hdr_u16)


def linear_to_srgb8(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.uint8)
    lib().orc_linear_to_srgb8_batch(_ptr(x), x.size, _ptr(out))
    return out


def half_to_float(a):
    return np.ascontiguousarray(a, np.uint16).view(np.float16).astype(np.float32)
