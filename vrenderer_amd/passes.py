"""Host-side mirror of the reference's operator interface for the hot path.

Names follow the reference (source/terrain/TerrainPass.h:133-159, QuadTree.h:94-118,
Renderer.h:50-110, and DeferredLightingPass as called at Renderer.cpp:417-428); every
method forwards to one C-ABI entry point of libvrterrain.so.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import (GBufferDesc, Instance, Light, Partition, RenderParams, TerrainParams, View, check)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Context:
    """nvrhi device + the frame's single command list (main.cpp:57-61, Renderer.cpp:48)."""

    def __init__(self, device=0):
        self.lib = capi.load_library()
        h = C.c_void_p()
        check(self.lib.vr_context_create(device, C.byref(h)), "vr_context_create")
        self.handle = h
        self.device = device

    def set_stream(self, hip_stream):
        check(self.lib.vr_context_set_stream(self.handle, C.c_void_p(hip_stream)), "vr_context_set_stream")

    def synchronize(self):
        check(self.lib.vr_context_synchronize(self.handle), "vr_context_synchronize")

    def set_async_geometry(self, enable):
        """Geometry stages of TerrainPass.Render on a second stream (overlaps the previous frame's lighting)."""
        check(self.lib.vr_context_set_option(self.handle, capi.VR_OPT_ASYNC_GEOMETRY, int(enable)), "vr_context_set_option")

    def set_raster_tile(self, edge):
        """Edge of the tile pass's raster tiles: 0 = by frame size and split (default), 32 or 64 = pinned."""
        check(self.lib.vr_context_set_option(self.handle, capi.VR_OPT_RASTER_TILE, int(edge)), "vr_context_set_option")

    def set_plane_tracking(self, enable):
        """The tile pass does not rewrite a G-buffer plane the library knows to be all zero (the emissive plane; default on)."""
        check(self.lib.vr_context_set_option(self.handle, capi.VR_OPT_PLANE_TRACKING, int(enable)), "vr_context_set_option")

    def set_scratch_worst_case(self, enable):
        """Terrains created from now on size their per-frame scratch for max_instances up front instead of by high-water mark."""
        check(self.lib.vr_context_set_option(self.handle, capi.VR_OPT_SCRATCH_WORST_CASE, int(enable)), "vr_context_set_option")

    def set_dispatch_events(self, enable):
        """Tile pass / lighting pass launched with dispatch-stamped events that double as cross-stream dependencies (default on)."""
        check(self.lib.vr_context_set_option(self.handle, capi.VR_OPT_DISPATCH_EVENTS, int(enable)), "vr_context_set_option")

    def timing_enable(self, enable=True):
        """0 / False: off; 1 / True: events around every kernel; 2: only the tile pass and the lighting passes (no host cost)."""
        check(self.lib.vr_timing_enable(self.handle, int(enable)), "vr_timing_enable")

    def timing_collect(self):
        """{kernel name: (total ms, launches)} since the last enable/collect (HIP events on the stream)."""
        ms = (C.c_float * capi.VR_K_COUNT)()
        n = (C.c_int32 * capi.VR_K_COUNT)()
        check(self.lib.vr_timing_collect(self.handle, ms, n), "vr_timing_collect")
        return {self.lib.vr_kernel_name(i).decode(): (float(ms[i]), int(n[i])) for i in range(capi.VR_K_COUNT) if n[i]}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vr_context_destroy(self.handle)
            self.handle = None


def default_terrain_params(size=2048.0):
    """TerrainSettings (TerrainPass.h:23-30) with SURFACE_SIZE = WORLD_SIZE = size."""
    p = TerrainParams()
    capi.load_library().vr_terrain_default_params(C.byref(p))
    p.surface_size = float(size)
    p.world_size = float(size)
    return p


def default_render_params(max_height=400.0, **kw):
    rp = RenderParams()
    capi.load_library().vr_render_default_params(C.byref(rp))
    rp.max_height = max_height
    for k, v in kw.items():
        setattr(rp, k, v)
    return rp


def make_view(eye, target, width, height, vfov_deg=60.0, z_near=0.1, z_far=10000.0, up=(0.0, 1.0, 0.0)):
    """FirstPersonCamera::LookAt + perspProjD3DStyle(60 deg, aspect, 0.1, 10000) (Renderer.cpp:97,315)."""
    v = View()
    check(capi.load_library().vr_view_from_camera(_f3(eye), _f3(target), _f3(up),
                                                  np.float32(np.radians(np.float32(vfov_deg))), z_near, z_far,
                                                  width, height, C.byref(v)), "vr_view_from_camera")
    return v


def directional_light(direction, irradiance=1.0, angular_size_deg=0.53, color=(1.0, 1.0, 1.0)):
    """DirectionalLight::FillLightConstants: normalised direction, angular size in radians."""
    l = Light()
    d = np.asarray(direction, np.float32)
    d = d / np.float32(np.sqrt(np.float32((d * d).sum())))
    l.direction[:] = [float(x) for x in d]
    l.type = capi.VR_LIGHT_DIRECTIONAL
    l.color[:] = [float(c) for c in color]
    l.intensity = float(irradiance)
    l.angular_size_or_inv_range = float(np.float32(np.radians(np.float32(angular_size_deg))))
    return l


def point_light(position, intensity, light_range, color=(1.0, 1.0, 1.0), radius=0.0):
    """PointLight::FillLightConstants: position, radius (0 = punctual), intensity, 1/range."""
    l = Light()
    l.type = capi.VR_LIGHT_POINT
    l.position[:] = [float(x) for x in position]
    l.color[:] = [float(c) for c in color]
    l.intensity = float(intensity)
    l.radius = float(radius)
    l.angular_size_or_inv_range = float(np.float32(1.0) / np.float32(light_range)) if light_range > 0 else 0.0
    return l


def spot_light(position, direction, intensity, light_range, inner_angle_deg, outer_angle_deg, color=(1.0, 1.0, 1.0), radius=0.0):
    """SpotLight::FillLightConstants: a point light with a cone (half angles, here in degrees) around `direction`."""
    l = point_light(position, intensity, light_range, color, radius)
    l.type = capi.VR_LIGHT_SPOT
    d = np.asarray(direction, np.float32)
    d = d / np.float32(np.sqrt(np.float32((d * d).sum())))
    l.direction[:] = [float(x) for x in d]
    l.inner_angle = float(np.float32(np.radians(np.float32(inner_angle_deg))))
    l.outer_angle = float(np.float32(np.radians(np.float32(outer_angle_deg))))
    return l


def reference_sun():
    """The "Sun" Renderer::SceneLoaded creates (Renderer.cpp:133-146)."""
    l = directional_light((-0.9, -0.25, 0.35), irradiance=1.0, angular_size_deg=0.53)
    l.out_of_bounds_shadow = 1.0        # LightConstants::outOfBoundsShadow: lit outside the shadow map
    return l


class RenderTargets:
    """RenderTargets : GBufferRenderTargets (Renderer.h:50-110): Init / Clear."""

    PLANES = {"depth": (0, np.float32, ()), "diffuse": (1, np.uint32, ()), "specular": (2, np.uint32, ()),
              "normals": (3, np.uint16, (4,)), "emissive": (4, np.uint16, (4,))}

    def __init__(self, ctx):
        self.ctx = ctx
        self.handle = None
        self.width = self.height = 0

    def Init(self, width, height):
        self.close()
        h = C.c_void_p()
        check(self.ctx.lib.vr_gbuffer_create(self.ctx.handle, width, height, C.byref(h)), "vr_gbuffer_create")
        self.handle, self.width, self.height = h, width, height
        return self

    def Clear(self):
        check(self.ctx.lib.vr_gbuffer_clear(self.handle), "vr_gbuffer_clear")

    def plane_known_zero(self, plane="emissive"):
        """True when the next tile pass will not rewrite `plane` because the library knows it holds only zeros."""
        return bool(self.ctx.lib.vr_gbuffer_plane_known_zero(self.handle, self.PLANES[plane][0]))

    def region_census(self):
        """Plane-state tracking per 8x32-pixel region: dict(unknown, specular_constant, clear, total); synchronises."""
        c = (C.c_uint32 * 4)()
        check(self.ctx.lib.vr_gbuffer_region_census(self.handle, c), "vr_gbuffer_region_census")
        return dict(unknown=c[0], specular_constant=c[1], clear=c[2], total=c[3])

    def describe(self):
        d = GBufferDesc()
        check(self.ctx.lib.vr_gbuffer_describe(self.handle, C.byref(d)), "vr_gbuffer_describe")
        return d

    def download(self, plane):
        idx, dt, tail = self.PLANES[plane]
        out = np.empty((self.height, self.width) + tail, dt)
        check(self.ctx.lib.vr_gbuffer_download(self.handle, idx, _vp(out), out.nbytes), "vr_gbuffer_download")
        return out

    def upload(self, plane, arr):
        idx, dt, tail = self.PLANES[plane]
        a = np.ascontiguousarray(arr, dt)
        assert a.shape == (self.height, self.width) + tail
        check(self.ctx.lib.vr_gbuffer_upload(self.handle, idx, _vp(a), a.nbytes), "vr_gbuffer_upload")

    def close(self):
        if self.handle:
            self.ctx.lib.vr_gbuffer_destroy(self.handle)
            self.handle = None


class HdrImage:
    """HdrColor, RGBA16_FLOAT (Renderer.h:69-79); optionally over caller-owned device memory."""

    def __init__(self, ctx, width, height, external_ptr=None):
        self.ctx, self.width, self.height = ctx, width, height
        h = C.c_void_p()
        check(ctx.lib.vr_image_create(ctx.handle, width, height, C.c_void_p(external_ptr) if external_ptr else None,
                                      C.byref(h)), "vr_image_create")
        self.handle = h

    @property
    def device_ptr(self):
        return self.ctx.lib.vr_image_device_ptr(self.handle)

    def download(self, nbytes=None):
        n = nbytes if nbytes is not None else self.width * self.height * 8
        out = np.empty(n // 2, np.uint16)
        check(self.ctx.lib.vr_image_download(self.handle, _vp(out), n), "vr_image_download")
        if nbytes is None:
            return out.reshape(self.height, self.width, 4)
        return out

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        check(self.ctx.lib.vr_image_upload(self.handle, _vp(a), a.nbytes), "vr_image_upload")

    def close(self):
        if self.handle:
            self.ctx.lib.vr_image_destroy(self.handle)
            self.handle = None


class LdrImage:
    """LdrColor, SRGBA8_UNORM (Renderer.h:81-92): a vr_ldr_image of width*height*4 bytes of device memory, or the
    packed RGB8 tiles of one rank when used as a multi-GPU send buffer (capacity_bytes)."""

    def __init__(self, ctx, width, height, external_ptr=None, capacity_bytes=None):
        self.ctx, self.width, self.height = ctx, width, height
        self.capacity = capacity_bytes if capacity_bytes is not None else width * height * 4
        self.owned = not external_ptr
        h = C.c_void_p()
        check(ctx.lib.vr_ldr_image_create(ctx.handle, width, height, self.capacity, C.c_void_p(external_ptr or 0), C.byref(h)),
              "vr_ldr_image_create")
        self.handle = h

    @property
    def device_ptr(self):
        return self.ctx.lib.vr_ldr_image_device_ptr(self.handle)

    def download(self, nbytes=None):
        n = nbytes if nbytes is not None else self.width * self.height * 4
        raw = np.empty(n, np.uint8)
        check(self.ctx.lib.vr_ldr_image_download(self.handle, _vp(raw), n), "vr_ldr_image_download")
        return raw.reshape(self.height, self.width, 4) if nbytes is None else raw

    def upload(self, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        check(self.ctx.lib.vr_ldr_image_upload(self.handle, _vp(a), a.nbytes), "vr_ldr_image_upload")

    def close(self):
        if self.handle:
            self.ctx.lib.vr_ldr_image_destroy(self.handle)
            self.handle = None


def default_tonemap_params(**kw):
    """ToneMappingParameters() + the log-luminance range of ToneMappingPass::CreateParameters (Renderer.cpp:256,431)."""
    p = capi.TonemapParams()
    capi.load_library().vr_tonemap_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class ToneMappingPass:
    """donut::render::ToneMappingPass as the reference drives it (Renderer.cpp:188-189,256-257,430-431)."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        check(ctx.lib.vr_tonemap_create(ctx.handle, C.byref(h)), "vr_tonemap_create")
        self.handle = h
        self.frame_time = 0.0

    def AdvanceFrame(self, seconds):
        self.frame_time = float(seconds)

    def ResetExposure(self, adapted_luminance=0.0):
        check(self.ctx.lib.vr_tonemap_reset_exposure(self.handle, adapted_luminance), "vr_tonemap_reset_exposure")

    def ResetHistogram(self):
        check(self.ctx.lib.vr_tonemap_reset_histogram(self.handle), "vr_tonemap_reset_histogram")

    def AddFrameToHistogram(self, params, hdr, width=None, height=None, partition=None):
        check(self.ctx.lib.vr_tonemap_add_frame_to_histogram(
            self.handle, C.byref(params), hdr.handle, width or hdr.width, height or hdr.height,
            C.byref(partition) if partition is not None else None), "vr_tonemap_add_frame_to_histogram")

    @property
    def histogram_device_ptr(self):
        return self.ctx.lib.vr_tonemap_histogram_device_ptr(self.handle)

    def ComputeExposure(self, params):
        check(self.ctx.lib.vr_tonemap_compute_exposure(self.handle, C.byref(params), self.frame_time), "vr_tonemap_compute_exposure")

    def Render(self, params, hdr, ldr, width=None, height=None, partition=None):
        check(self.ctx.lib.vr_tonemap_render(
            self.handle, C.byref(params), hdr.handle, width or hdr.width, height or hdr.height, C.c_void_p(ldr.device_ptr),
            ldr.capacity, C.byref(partition) if partition is not None else None), "vr_tonemap_render")

    def SimpleRender(self, params, hdr, ldr):
        """ResetHistogram + AddFrameToHistogram + ComputeExposure + Render (one GPU)."""
        check(self.ctx.lib.vr_tonemap_simple_render(self.handle, C.byref(params), self.frame_time, hdr.handle,
                                                    C.c_void_p(ldr.device_ptr), ldr.capacity), "vr_tonemap_simple_render")

    def AllReduceHistogram(self, comm):
        """The tone mapper's one exchange step between AddFrameToHistogram(partition) and ComputeExposure: the 256 bins summed
        over the ranks (ncclAllReduce on this pass's context stream; comm: vrenderer_amd.rccl.Communicator or an ncclComm_t)."""
        check(self.ctx.lib.vr_tonemap_allreduce_histogram(self.handle, getattr(comm, "handle", comm)), "vr_tonemap_allreduce_histogram")

    def download(self):
        hist = np.zeros(capi.VR_TONEMAP_BINS, np.uint32)
        lum = C.c_float()
        check(self.ctx.lib.vr_tonemap_download(self.handle, hist.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(lum)),
              "vr_tonemap_download")
        return hist, float(lum.value)

    def close(self):
        if self.handle:
            self.ctx.lib.vr_tonemap_destroy(self.handle)
            self.handle = None


class TerrainPass:
    """vRenderer::TerrainPass + its QuadTree (TerrainPass.h:32-159, QuadTree.h:64-127)."""

    def __init__(self, ctx, params=None):
        self.ctx = ctx
        self.params = params or default_terrain_params()
        self.handle = None

    def Init(self, heightmap_r8, albedo_srgba8):
        """TerrainPass::Init + QuadTree::Init (TerrainPass.cpp:34-141, QuadTree.cpp:19-52)."""
        h = np.ascontiguousarray(heightmap_r8, np.uint8)
        a = np.ascontiguousarray(albedo_srgba8, np.uint8)
        assert h.ndim == 2 and a.ndim == 3 and a.shape[2] == 4
        out = C.c_void_p()
        check(self.ctx.lib.vr_terrain_create(self.ctx.handle, C.byref(self.params), _vp(h), h.shape[1], h.shape[0],
                                             _vp(a), a.shape[1], a.shape[0], C.byref(out)), "vr_terrain_create")
        self.handle = out
        return self

    def GetNumLods(self):
        return self.ctx.lib.vr_terrain_num_lods(self.handle)

    def GetLodRanges(self):
        out = (C.c_float * capi.VR_MAX_LODS)()
        check(self.ctx.lib.vr_terrain_lod_ranges(self.handle, out), "vr_terrain_lod_ranges")
        return np.array(out[:], np.float32)

    def download_mip(self, which, level):
        """Read back one level of the device mip chain ('height' -> (h,w) u8, 'albedo' -> (h,w,4) u8)."""
        idx = 0 if which == "height" else 1
        w, h, lv = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.ctx.lib.vr_terrain_download_mip(self.handle, idx, level, None, 0, C.byref(w), C.byref(h), C.byref(lv)),
              "vr_terrain_download_mip")
        out = np.empty((h.value, w.value) if idx == 0 else (h.value, w.value, 4), np.uint8)
        check(self.ctx.lib.vr_terrain_download_mip(self.handle, idx, level, _vp(out), out.nbytes, None, None, None),
              "vr_terrain_download_mip")
        return out

    def mip_levels(self, which):
        lv = C.c_int32()
        check(self.ctx.lib.vr_terrain_download_mip(self.handle, 0 if which == "height" else 1, 0, None, 0, None, None,
                                                   C.byref(lv)), "vr_terrain_download_mip")
        return lv.value

    def SetHeight(self, enable=True):
        """QuadTree::SetHeight for every node on the device + m_HeightLoaded (QuadTree.cpp:46-51,164-208)."""
        check(self.ctx.lib.vr_terrain_update_heights(self.handle, int(enable)), "vr_terrain_update_heights")

    def node_heights(self, first, count):
        out = np.zeros((count, 2), np.float32)
        check(self.ctx.lib.vr_terrain_download_node_heights(self.handle, first, count, _vp(out)), "vr_terrain_download_node_heights")
        return out

    def NodeSelect(self, view, max_height=400.0):
        """ClearSelectedNodes + NodeSelect + UpdateTransforms (TerrainPass.cpp:173-190).

        Returns (count, node_ids, instance bytes[count,112]) in m_SelectedNodes order."""
        cap = self.params.max_instances
        ids = np.zeros(cap, np.uint32)
        inst = (Instance * cap)()
        n = C.c_uint32()
        check(self.ctx.lib.vr_terrain_select(self.handle, C.byref(view), max_height, _vp(ids), inst, C.byref(n)),
              "vr_terrain_select")
        k = n.value
        return k, ids[:k].copy(), np.frombuffer(inst, dtype=np.uint8).reshape(cap, 112)[:k].copy()

    def Render(self, view, view_prev, render_targets, render_params, partition=None):
        """TerrainPass::Render (TerrainPass.cpp:143-232); asynchronous on the context's stream."""
        check(self.ctx.lib.vr_terrain_render(self.handle, C.byref(view), C.byref(view_prev if view_prev is not None else view),
                                             render_targets.handle, C.byref(render_params),
                                             C.byref(partition) if partition is not None else None), "vr_terrain_render")

    def RenderLit(self, view, render_targets, render_params, lights, ambient_top, ambient_bottom, output, partition=None):
        """vr_terrain_render_lit: TerrainPass::Render + DeferredLightingPass::Render fused into the tile pass (opt-in): writes the
        depth plane and `output` only; same bits as Render(assume_cleared=1) + DeferredLightingPass.Render."""
        arr = light_array(lights)
        check(self.ctx.lib.vr_terrain_render_lit(self.handle, C.byref(view), render_targets.handle, C.byref(render_params),
                                                 C.byref(partition) if partition is not None else None, arr, len(lights),
                                                 _f3(ambient_top), _f3(ambient_bottom), output.handle), "vr_terrain_render_lit")

    def render_stats(self):
        out = (C.c_uint32 * 8)()
        check(self.ctx.lib.vr_debug_render_stats(self.handle, out), "vr_debug_render_stats")
        keys = ("nodes", "flags", "clip_subtris", "clip_verts", "clipped_tris", "bin_entries", "max_bin", "nonempty_bins")
        return dict(zip(keys, [int(v) for v in out]))

    def tile_order(self, max_tiles=1 << 18):
        """(tiles, bin_lengths) of the last Render's tile pass in launch order (k_scan: longest bins first, eight classes)."""
        tiles = np.zeros(max_tiles, np.int32)
        lens = np.zeros(max_tiles, np.uint32)
        n = C.c_int32()
        check(self.ctx.lib.vr_debug_tile_order(self.handle, _vp(tiles), _vp(lens), max_tiles, C.byref(n)), "vr_debug_tile_order")
        return tiles[:n.value].copy(), lens[:n.value].copy()

    def memory_bytes(self):
        """Device memory this terrain holds: dict(textures, scratch, node_heights, total) in bytes."""
        out = (C.c_uint64 * 4)()
        check(self.ctx.lib.vr_terrain_memory_bytes(self.handle, out), "vr_terrain_memory_bytes")
        return dict(zip(("textures", "scratch", "node_heights", "total"), [int(v) for v in out]))

    def download_vertices(self, first_instance, num_instances=1):
        """main_vs outputs of the last Render for whole instances: (n, 1089, 6) floats = clip xyzw + world xz (test helper)."""
        import numpy as np
        n = int(num_instances) * 1089
        out = np.empty((n, 6), np.float32)
        check(self.ctx.lib.vr_debug_download_vertices(self.handle, int(first_instance) * 1089, n, out.ctypes.data_as(C.c_void_p)),
              "vr_debug_download_vertices")
        return out.reshape(int(num_instances), 1089, 6)

    def Prepare(self, view, render_targets, render_params, partition=None):
        """Build the next frame's geometry ahead of time (overlaps the current frame's tile pass)."""
        check(self.ctx.lib.vr_terrain_prepare(self.handle, C.byref(view), render_targets.handle, C.byref(render_params),
                                              C.byref(partition) if partition is not None else None), "vr_terrain_prepare")

    def num_chunks(self):
        """EditorParams::m_NumChunks (TerrainPass.cpp:198); synchronises."""
        n = C.c_uint32()
        check(self.ctx.lib.vr_terrain_num_chunks(self.handle, C.byref(n)), "vr_terrain_num_chunks")
        return n.value

    def close(self):
        if self.handle:
            self.ctx.lib.vr_terrain_destroy(self.handle)
            self.handle = None


class DeferredLightingPass:
    """donut::render::DeferredLightingPass as used at Renderer.cpp:239-240,417-428."""

    def __init__(self, ctx):
        self.ctx = ctx

    def Render(self, view, render_targets, lights, ambient_top, ambient_bottom, output, partition=None, shadow_map=None,
               shadow_light_index=0):
        """shadow_map: a CascadedShadowMap whose light view was set up and rendered this frame
        (DirectionalLight::shadowMap, Renderer.cpp:336)."""
        n = len(lights)
        arr = (Light * max(n, 1))(*lights)
        part = C.byref(partition) if partition is not None else None
        if shadow_map is None:
            check(self.ctx.lib.vr_deferred_light(self.ctx.handle, C.byref(view), render_targets.handle, arr, n,
                                                 _f3(ambient_top), _f3(ambient_bottom), output.handle, part), "vr_deferred_light")
            return
        sb = capi.ShadowBinding(C.cast(C.pointer(shadow_map.view), C.c_void_p), shadow_map.targets.handle, shadow_light_index,
                                shadow_map.params.depth_bias)
        check(self.ctx.lib.vr_deferred_light_shadowed(self.ctx.handle, C.byref(view), render_targets.handle, arr, n,
                                                      _f3(ambient_top), _f3(ambient_bottom), output.handle, part, C.byref(sb)),
              "vr_deferred_light_shadowed")


def default_shadow_params(world_size=2048.0, **kw):
    """CascadedShadowMap(device, 2048, 1, 0, fmt) + the arguments of SetupForPlanarViewStable (Renderer.cpp:83,345-352)."""
    p = capi.ShadowParams()
    capi.load_library().vr_shadow_default_params(C.byref(p), world_size)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class CascadedShadowMap:
    """donut::render::CascadedShadowMap with one cascade, as the reference uses it (Renderer.cpp:83-87, 333-367):
    SetupForPlanarViewStable -> Clear -> TerrainPass.Render(depthOnly) into it -> lighting."""

    def __init__(self, ctx, params):
        self.ctx, self.params = ctx, params
        self.targets = RenderTargets(ctx).Init(params.resolution, params.resolution)     # m_ShadowFramebuffer (depth plane)
        self.view = View()

    def SetupForPlanarViewStable(self, light, camera_view):
        check(self.ctx.lib.vr_shadow_view_setup(C.byref(light), C.byref(camera_view), C.byref(self.params), C.byref(self.view)),
              "vr_shadow_view_setup")
        return self.view

    def GetView(self):
        return self.view

    def Clear(self):
        self.targets.Clear()

    def RenderTerrain(self, terrain_pass, max_height=400.0, lock_view=0):
        """The "Terrain Shadow" scope (Renderer.cpp:356-372): depthOnly render from the light's view."""
        rp = default_render_params(max_height, depth_only=1, assume_cleared=1, lock_view=lock_view)
        terrain_pass.Render(self.view, self.view, self.targets, rp)

    def PrepareTerrain(self, terrain_pass, light, camera_view, max_height=400.0):
        """vr_terrain_prepare for the shadow pass of an upcoming frame: the light view that frame's
        SetupForPlanarViewStable(light, camera_view) will produce, built ahead on a geometry stream."""
        lv = View()
        check(self.ctx.lib.vr_shadow_view_setup(C.byref(light), C.byref(camera_view), C.byref(self.params), C.byref(lv)),
              "vr_shadow_view_setup")
        terrain_pass.Prepare(lv, self.targets, default_render_params(max_height, depth_only=1, assume_cleared=1))

    def download_depth(self):
        return self.targets.download("depth")

    def close(self):
        self.targets.close()


class Frame:
    """One frame for vr_frame_submit - the terrain part of Renderer::RecordCommand (Renderer.cpp:321-446) as one call: Render
    [Clear fused] -> Prepare for up to two upcoming frames -> lighting -> optional tone-map stage [-> exchange].  Everything
    that does not change from frame to frame (lights, ambient terms, render parameters, partition, tone mapper, buffers) is
    set once; submit() fills in the views and the output image and makes the one call."""

    def __init__(self, terrain_pass, render_targets, render_params, lights, ambient_top, ambient_bottom, partition=None, tiled=False,
                 tonemap=None, tonemap_params=None, ldr=None, comm=None, gathered_ptr=None, ldr_frame=None):
        self.tp, self.rt = terrain_pass, render_targets
        self._keep = (render_params, light_array(lights), partition, tonemap, tonemap_params, ldr, comm, ldr_frame)
        d = capi.FrameDesc()
        d.render = C.addressof(render_params)
        d.part = C.addressof(partition) if partition is not None else None
        d.lights = C.addressof(self._keep[1])
        d.num_lights = len(lights)
        d.tiled = 1 if tiled else 0
        d.ambient_top[:] = [float(x) for x in ambient_top]
        d.ambient_bottom[:] = [float(x) for x in ambient_bottom]
        if tonemap is not None:
            d.tonemap = tonemap.handle
            d.tonemap_params = C.addressof(tonemap_params)
            d.frame_time_seconds = tonemap.frame_time
            d.ldr_out = ldr.device_ptr
            d.ldr_capacity = ldr.capacity
            if comm is not None:
                d.nccl_comm = getattr(comm, "handle", comm)
                d.gathered = gathered_ptr
                d.ldr_frame = ldr_frame.device_ptr
        self.desc = d
        self._submit = terrain_pass.ctx.lib.vr_frame_submit

    def submit(self, view, hdr_out, prepare=(), ldr=None):
        d = self.desc
        d.view = C.addressof(view)
        d.prepare_views[0] = C.addressof(prepare[0]) if len(prepare) > 0 else None
        d.prepare_views[1] = C.addressof(prepare[1]) if len(prepare) > 1 else None
        d.hdr_out = hdr_out.handle
        if ldr is not None:
            d.ldr_out, d.ldr_capacity = ldr.device_ptr, ldr.capacity
        rc = self._submit(self.tp.handle, self.rt.handle, C.byref(d))
        if rc:
            check(rc, "vr_frame_submit")


def light_array(lights):
    """The contiguous vr_light[] the C ABI takes, built once: a host that keeps its lights in such an array (as the
    reference keeps them in its scene graph) does not convert a Python list per frame.  Render() accepts either."""
    return lights if isinstance(lights, C.Array) else (Light * max(len(lights), 1))(*lights)


class TiledDeferredLightingPass(DeferredLightingPass):
    """DeferredLightingPass for many lights (BASELINE config 5): light lists per 32x32 screen tile."""

    def Render(self, view, render_targets, lights, ambient_top, ambient_bottom, output, partition=None, num_lights=None):
        n = len(lights) if num_lights is None else num_lights
        arr = light_array(lights)
        check(self.ctx.lib.vr_deferred_light_tiled(self.ctx.handle, C.byref(view), render_targets.handle, arr, n,
                                                   _f3(ambient_top), _f3(ambient_bottom), output.handle,
                                                   C.byref(partition) if partition is not None else None),
              "vr_deferred_light_tiled")

    def Status(self):
        """Waits for the passes queued so far; raises VrError(VR_ERR_OVERFLOW) if a tile kept more than
        VR_TILE_LIGHT_CAP lights since the last call (the launch itself stays asynchronous)."""
        check(self.ctx.lib.vr_deferred_tiled_status(self.ctx.handle), "vr_deferred_tiled_status")


def synthetic_point_lights(n, world_size, heightmap, max_height=400.0, seed=9001, intensity=50.0):
    """positions uniform in the world xz-square at terrain height + U(2,30), range U(20,80),
    colour uniform; numpy PCG64 with a fixed seed.  The survey leaves the intensity open: 50 keeps the lit terrain
    of the 2048 world within [0, 1] (peak ~0.9 with all 1023 lights), the range the 1e-4 RMS tolerance is meant for -
    at 400 the frame peaks near 8, where one half-precision ulp of the RGBA16F output is already 4e-3."""
    rng = np.random.default_rng(seed)
    size = heightmap.shape[0]
    xz = rng.uniform(-0.5 * world_size, 0.5 * world_size, (n, 2))
    tx = np.clip(((xz[:, 0] / world_size + 0.5) * size).astype(int), 0, size - 1)
    tz = np.clip(((xz[:, 1] / world_size + 0.5) * size).astype(int), 0, size - 1)
    y = heightmap[tz, tx].astype(np.float64) / 255.0 * max_height + rng.uniform(2.0, 30.0, n)
    rad = rng.uniform(20.0, 80.0, n)
    col = rng.uniform(0.0, 1.0, (n, 3))
    return [point_light((xz[i, 0], y[i], xz[i, 1]), intensity, rad[i], col[i]) for i in range(n)]


def synth_heightmap(ctx, size, seed=1337):
    out = np.empty((size, size), np.uint8)
    check(ctx.lib.vr_synth_heightmap(ctx.handle, size, seed, _vp(out)), "vr_synth_heightmap")
    return out


def synth_albedo(ctx, size, height, seed=4242):
    h = np.ascontiguousarray(height, np.uint8)
    out = np.empty((size, size, 4), np.uint8)
    check(ctx.lib.vr_synth_albedo(ctx.handle, size, seed, _vp(h), _vp(out)), "vr_synth_albedo")
    return out


def partition_info(width, height, rank, world):
    lib = capi.load_library()
    p = Partition(rank, world)
    tx, ty, owned, mo = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    check(lib.vr_partition_num_tiles(width, height, C.byref(p), C.byref(tx), C.byref(ty), C.byref(owned), C.byref(mo)),
          "vr_partition_num_tiles")
    return dict(tiles_x=tx.value, tiles_y=ty.value, owned=owned.value, max_owned=mo.value,
                packed_bytes=lib.vr_partition_packed_bytes(width, height, world),
                packed_bytes_ldr=lib.vr_partition_packed_bytes_ldr(width, height, world))


def partition_prepare(ctx, width, height, partition):
    check(ctx.lib.vr_partition_prepare(ctx.handle, width, height, C.byref(partition)), "vr_partition_prepare")


def frame_detile_ldr(ctx, gathered_ptr, world, width, height, ldr):
    check(ctx.lib.vr_frame_detile_ldr(ctx.handle, C.c_void_p(gathered_ptr), world, width, height, C.c_void_p(ldr.device_ptr)),
          "vr_frame_detile_ldr")


def frame_allgather_ldr(ctx, comm, packed_ptr, gathered_ptr, world, width, height, ldr):
    """vr_frame_allgather_ldr: ncclAllGather of this rank's packed RGB8 tiles on the context's stream + the de-tile into the
    row-major SRGBA8 frame `ldr` (comm: vrenderer_amd.rccl.Communicator or an ncclComm_t)."""
    check(ctx.lib.vr_frame_allgather_ldr(ctx.handle, getattr(comm, "handle", comm), C.c_void_p(packed_ptr), C.c_void_p(gathered_ptr), world,
                                         width, height, C.c_void_p(ldr.device_ptr)), "vr_frame_allgather_ldr")


def frame_allgather(ctx, comm, packed_ptr, gathered_ptr, world, frame):
    """vr_frame_allgather: the same for packed RGB16F tiles into the row-major RGBA16F `frame` (an HdrImage)."""
    check(ctx.lib.vr_frame_allgather(ctx.handle, getattr(comm, "handle", comm), C.c_void_p(packed_ptr), C.c_void_p(gathered_ptr), world,
                                     frame.handle), "vr_frame_allgather")


def frame_allgather_tiles(ctx, comm, packed_ptr, gathered_ptr, world, bytes_per_rank):
    """vr_frame_allgather_tiles: the all-gather alone; the host de-tiles where it likes (frame_detile[_ldr] on another context)."""
    check(ctx.lib.vr_frame_allgather_tiles(ctx.handle, getattr(comm, "handle", comm), C.c_void_p(packed_ptr), C.c_void_p(gathered_ptr), world,
                                           bytes_per_rank), "vr_frame_allgather_tiles")


def frame_detile(ctx, gathered_ptr, world, frame):
    check(ctx.lib.vr_frame_detile(ctx.handle, C.c_void_p(gathered_ptr), world, frame.handle), "vr_frame_detile")
