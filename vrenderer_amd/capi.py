"""ctypes view of include/vrterrain.h: POD structs and the libvrterrain.so loader.

The structs mirror the C header field for field; `load_library()` fails loudly when
the HIP extension has not been built — there is no CPU fallback in the product.
"""
import ctypes as C
import os

VR_MAX_LODS = 12
VR_OWNER_TILE = 128

VR_OK = 0
VR_ERR_INVALID_ARGUMENT = 1
VR_ERR_NO_DEVICE = 2
VR_ERR_OUT_OF_MEMORY = 3
VR_ERR_TOO_MANY_INSTANCES = 4
VR_ERR_OVERFLOW = 5
VR_ERR_HIP = 6

VR_LIGHT_DIRECTIONAL = 1
VR_LIGHT_SPOT = 2
VR_LIGHT_POINT = 3
VR_K_COUNT = 19
VR_TONEMAP_BINS = 256
VR_OPT_ASYNC_GEOMETRY = 1
VR_OPT_DISPATCH_EVENTS = 2
VR_OPT_RASTER_TILE = 3
VR_OPT_PLANE_TRACKING = 4
VR_OPT_SCRATCH_WORST_CASE = 5


class TerrainParams(C.Structure):
    _fields_ = [
        ("max_instances", C.c_int32),
        ("surface_size", C.c_float),
        ("world_size", C.c_float),
        ("grid_size", C.c_int32),
        ("min_lod_distance", C.c_float),
        ("morph_start", C.c_float),
        ("location", C.c_float * 3),
        ("reserved", C.c_int32),
    ]


class View(C.Structure):
    _fields_ = [
        ("world_to_view", C.c_float * 16),
        ("view_to_clip", C.c_float * 16),
        ("world_to_clip", C.c_float * 16),
        ("clip_to_world", C.c_float * 16),
        ("camera_pos", C.c_float * 4),
        ("planes", (C.c_float * 4) * 6),
        ("viewport_x", C.c_int32),
        ("viewport_y", C.c_int32),
        ("viewport_w", C.c_int32),
        ("viewport_h", C.c_int32),
        ("mirrored", C.c_int32),
        ("reverse_depth", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


class Instance(C.Structure):
    _fields_ = [
        ("padding", C.c_uint32),
        ("first_geometry_instance_index", C.c_uint32),
        ("first_geometry_index", C.c_uint32),
        ("num_geometries", C.c_uint32),
        ("transform", C.c_float * 12),
        ("prev_transform", C.c_float * 12),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("wireframe", C.c_int32),
        ("lock_view", C.c_int32),
        ("depth_only", C.c_int32),
        ("assume_cleared", C.c_int32),
        ("max_height", C.c_float),
        ("depth_ranges", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


class Light(C.Structure):
    _fields_ = [
        ("direction", C.c_float * 3), ("type", C.c_int32),
        ("position", C.c_float * 3), ("radius", C.c_float),
        ("color", C.c_float * 3), ("intensity", C.c_float),
        ("angular_size_or_inv_range", C.c_float),
        ("inner_angle", C.c_float), ("outer_angle", C.c_float),
        ("out_of_bounds_shadow", C.c_float),
    ]


class GBufferDesc(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("depth", C.c_void_p), ("diffuse", C.c_void_p), ("specular", C.c_void_p),
        ("normals", C.c_void_p), ("emissive", C.c_void_p),
    ]


class Partition(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world_size", C.c_int32)]


class ShadowParams(C.Structure):
    _fields_ = [("resolution", C.c_int32), ("max_shadow_distance", C.c_float), ("light_space_z_up", C.c_float),
                ("light_space_z_down", C.c_float), ("depth_bias", C.c_float), ("reserved", C.c_int32 * 3)]


class ShadowBinding(C.Structure):
    _fields_ = [("light_view", C.c_void_p), ("shadow_map", C.c_void_p), ("light_index", C.c_int32), ("depth_bias", C.c_float)]


class TonemapParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "histogram_low_percentile", "histogram_high_percentile", "eye_adaptation_speed_up", "eye_adaptation_speed_down",
        "min_adapted_luminance", "max_adapted_luminance", "exposure_bias", "white_point", "min_log_luminance",
        "max_log_luminance")]


class FrameDesc(C.Structure):
    """vr_frame_desc: one frame for vr_frame_submit (pointers as c_void_p; the Python side keeps the objects alive)."""
    _fields_ = [("view", C.c_void_p), ("prepare_views", C.c_void_p * 2), ("render", C.c_void_p), ("part", C.c_void_p),
                ("lights", C.c_void_p), ("num_lights", C.c_int32), ("tiled", C.c_int32),
                ("ambient_top", C.c_float * 3), ("ambient_bottom", C.c_float * 3), ("shadow", C.c_void_p), ("hdr_out", C.c_void_p),
                ("tonemap", C.c_void_p), ("tonemap_params", C.c_void_p), ("frame_time_seconds", C.c_float), ("reserved0", C.c_int32),
                ("ldr_out", C.c_void_p), ("ldr_capacity", C.c_size_t), ("nccl_comm", C.c_void_p), ("gathered", C.c_void_p),
                ("ldr_frame", C.c_void_p)]


assert C.sizeof(Instance) == 112
assert C.sizeof(Light) == 64
assert C.sizeof(FrameDesc) == 160

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "lib", "libvrterrain.so")
if os.environ.get("VRTERRAIN_LIB"):          # development: A/B another build of the same library (tools/build_variant.py)
    LIB_PATH = os.environ["VRTERRAIN_LIB"]

# every symbol include/vrterrain.h declares
EXPORTS = [
    "vr_context_create", "vr_context_destroy", "vr_context_set_stream", "vr_context_set_option", "vr_context_synchronize",
    "vr_last_error", "vr_version", "vr_build_experiments", "vr_timing_enable", "vr_timing_collect", "vr_timing_kernel_count", "vr_kernel_name", "vr_view_from_camera", "vr_terrain_default_params",
    "vr_render_default_params", "vr_terrain_create", "vr_terrain_destroy", "vr_terrain_num_lods",
    "vr_terrain_lod_ranges", "vr_terrain_download_mip", "vr_terrain_update_heights", "vr_terrain_download_node_heights", "vr_terrain_select", "vr_terrain_render", "vr_terrain_render_lit", "vr_terrain_prepare", "vr_terrain_num_chunks",
    "vr_gbuffer_create", "vr_gbuffer_destroy", "vr_gbuffer_clear", "vr_gbuffer_describe", "vr_gbuffer_plane_known_zero", "vr_gbuffer_region_census",
    "vr_gbuffer_download", "vr_gbuffer_upload", "vr_image_create", "vr_image_destroy",
    "vr_image_device_ptr", "vr_image_download", "vr_image_upload", "vr_ldr_image_create", "vr_ldr_image_destroy", "vr_ldr_image_device_ptr", "vr_ldr_image_capacity", "vr_ldr_image_download", "vr_ldr_image_upload", "vr_deferred_light", "vr_deferred_light_tiled", "vr_deferred_tiled_status", "vr_partition_num_tiles",
    "vr_partition_packed_bytes", "vr_partition_prepare", "vr_frame_detile",
    "vr_shadow_default_params", "vr_shadow_view_setup", "vr_deferred_light_shadowed",
    "vr_tonemap_default_params", "vr_tonemap_create", "vr_tonemap_destroy", "vr_tonemap_reset_exposure", "vr_tonemap_reset_histogram",
    "vr_tonemap_add_frame_to_histogram", "vr_tonemap_histogram_device_ptr", "vr_tonemap_compute_exposure", "vr_tonemap_render",
    "vr_tonemap_simple_render", "vr_tonemap_download", "vr_partition_packed_bytes_ldr", "vr_frame_detile_ldr", "vr_frame_submit", "vr_frame_allgather", "vr_frame_allgather_tiles", "vr_frame_allgather_ldr", "vr_tonemap_allreduce_histogram", "vr_synth_heightmap", "vr_synth_albedo", "vr_debug_srgb_encode", "vr_debug_fastmath_check", "vr_debug_render_stats", "vr_debug_tile_order", "vr_debug_download_vertices", "vr_terrain_memory_bytes",
]

_lib = None


def load_library():
    """Load libvrterrain.so (built by __graft_entry__.build()).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "vrenderer_amd has no CPU fallback.")
    # torch ships its own libamdhip64.so.7; import it first (when present) so that this
    # process holds ONE HIP runtime: our library's NEEDED libamdhip64.so.7 then binds
    # to the copy torch already loaded instead of pulling in /opt/rocm's.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not a requirement
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    P = C.POINTER
    vp = C.c_void_p
    sig = {
        "vr_context_create": (C.c_int, [C.c_int, P(vp)]),
        "vr_context_destroy": (None, [vp]),
        "vr_context_set_stream": (C.c_int, [vp, vp]),
        "vr_context_set_option": (C.c_int, [vp, C.c_int, C.c_int]),
        "vr_context_synchronize": (C.c_int, [vp]),
        "vr_last_error": (C.c_char_p, []),
        "vr_version": (C.c_char_p, []),
        "vr_build_experiments": (C.c_uint32, []),
        "vr_timing_enable": (C.c_int, [vp, C.c_int]),
        "vr_timing_collect": (C.c_int, [vp, P(C.c_float), P(C.c_int32)]),
        "vr_kernel_name": (C.c_char_p, [C.c_int]),
        "vr_view_from_camera": (C.c_int, [P(C.c_float), P(C.c_float), P(C.c_float), C.c_float, C.c_float,
                                          C.c_float, C.c_int32, C.c_int32, P(View)]),
        "vr_terrain_default_params": (None, [P(TerrainParams)]),
        "vr_render_default_params": (None, [P(RenderParams)]),
        "vr_terrain_create": (C.c_int, [vp, P(TerrainParams), vp, C.c_int32, C.c_int32, vp, C.c_int32,
                                        C.c_int32, P(vp)]),
        "vr_terrain_destroy": (None, [vp]),
        "vr_terrain_num_lods": (C.c_int, [vp]),
        "vr_terrain_lod_ranges": (C.c_int, [vp, P(C.c_float)]),
        "vr_terrain_download_mip": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_size_t, P(C.c_int32), P(C.c_int32),
                                              P(C.c_int32)]),
        "vr_terrain_update_heights": (C.c_int, [vp, C.c_int]),
        "vr_terrain_download_node_heights": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
        "vr_terrain_select": (C.c_int, [vp, P(View), C.c_float, vp, vp, P(C.c_uint32)]),
        "vr_terrain_render": (C.c_int, [vp, P(View), P(View), vp, P(RenderParams), P(Partition)]),
        "vr_terrain_prepare": (C.c_int, [vp, P(View), vp, P(RenderParams), P(Partition)]),
        "vr_terrain_num_chunks": (C.c_int, [vp, P(C.c_uint32)]),
        "vr_gbuffer_create": (C.c_int, [vp, C.c_int32, C.c_int32, P(vp)]),
        "vr_gbuffer_destroy": (None, [vp]),
        "vr_gbuffer_clear": (C.c_int, [vp]),
        "vr_gbuffer_describe": (C.c_int, [vp, P(GBufferDesc)]),
        "vr_gbuffer_plane_known_zero": (C.c_int, [vp, C.c_int]),
        "vr_gbuffer_region_census": (C.c_int, [vp, vp]),
        "vr_gbuffer_download": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
        "vr_gbuffer_upload": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
        "vr_image_create": (C.c_int, [vp, C.c_int32, C.c_int32, vp, P(vp)]),
        "vr_image_destroy": (None, [vp]),
        "vr_image_device_ptr": (vp, [vp]),
        "vr_image_download": (C.c_int, [vp, vp, C.c_size_t]),
        "vr_image_upload": (C.c_int, [vp, vp, C.c_size_t]),
        "vr_ldr_image_create": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_size_t, vp, P(vp)]),
        "vr_ldr_image_destroy": (None, [vp]),
        "vr_ldr_image_device_ptr": (vp, [vp]),
        "vr_ldr_image_capacity": (C.c_size_t, [vp]),
        "vr_ldr_image_download": (C.c_int, [vp, vp, C.c_size_t]),
        "vr_ldr_image_upload": (C.c_int, [vp, vp, C.c_size_t]),
        "vr_deferred_light": (C.c_int, [vp, P(View), vp, P(Light), C.c_int32, P(C.c_float), P(C.c_float),
                                        vp, P(Partition)]),
        "vr_deferred_light_tiled": (C.c_int, [vp, P(View), vp, P(Light), C.c_int32, P(C.c_float), P(C.c_float),
                                              vp, P(Partition)]),
        "vr_deferred_tiled_status": (C.c_int, [vp]),
        "vr_partition_num_tiles": (C.c_int, [C.c_int32, C.c_int32, P(Partition), P(C.c_int32), P(C.c_int32),
                                             P(C.c_int32), P(C.c_int32)]),
        "vr_partition_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
        "vr_partition_prepare": (C.c_int, [vp, C.c_int32, C.c_int32, P(Partition)]),
        "vr_frame_detile": (C.c_int, [vp, vp, C.c_int32, vp]),
        "vr_frame_allgather": (C.c_int, [vp, vp, vp, vp, C.c_int32, vp]),
        "vr_frame_allgather_ldr": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]),
        "vr_frame_allgather_tiles": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_size_t]),
        "vr_frame_submit": (C.c_int, [vp, vp, P(FrameDesc)]),
        "vr_timing_kernel_count": (C.c_int, []),
        "vr_terrain_render_lit": (C.c_int, [vp, P(View), vp, P(RenderParams), P(Partition), P(Light), C.c_int32, P(C.c_float), P(C.c_float), vp]),
        "vr_tonemap_allreduce_histogram": (C.c_int, [vp, vp]),
        "vr_shadow_default_params": (None, [P(ShadowParams), C.c_float]),
        "vr_shadow_view_setup": (C.c_int, [P(Light), P(View), P(ShadowParams), P(View)]),
        "vr_deferred_light_shadowed": (C.c_int, [vp, P(View), vp, P(Light), C.c_int32, P(C.c_float), P(C.c_float),
                                                 vp, P(Partition), P(ShadowBinding)]),
        "vr_tonemap_default_params": (None, [P(TonemapParams)]),
        "vr_tonemap_create": (C.c_int, [vp, P(vp)]),
        "vr_tonemap_destroy": (None, [vp]),
        "vr_tonemap_reset_exposure": (C.c_int, [vp, C.c_float]),
        "vr_tonemap_reset_histogram": (C.c_int, [vp]),
        "vr_tonemap_add_frame_to_histogram": (C.c_int, [vp, P(TonemapParams), vp, C.c_int32, C.c_int32, P(Partition)]),
        "vr_tonemap_histogram_device_ptr": (vp, [vp]),
        "vr_tonemap_compute_exposure": (C.c_int, [vp, P(TonemapParams), C.c_float]),
        "vr_tonemap_render": (C.c_int, [vp, P(TonemapParams), vp, C.c_int32, C.c_int32, vp, C.c_size_t, P(Partition)]),
        "vr_tonemap_simple_render": (C.c_int, [vp, P(TonemapParams), C.c_float, vp, vp, C.c_size_t]),
        "vr_tonemap_download": (C.c_int, [vp, P(C.c_uint32), P(C.c_float)]),
        "vr_partition_packed_bytes_ldr": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
        "vr_frame_detile_ldr": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]),
        "vr_synth_heightmap": (C.c_int, [vp, C.c_int32, C.c_uint32, vp]),
        "vr_synth_albedo": (C.c_int, [vp, C.c_int32, C.c_uint32, vp, vp]),
        "vr_debug_srgb_encode": (C.c_int, [vp, vp, C.c_size_t, vp]),
        "vr_debug_fastmath_check": (C.c_int, [vp, vp]),
        "vr_debug_render_stats": (C.c_int, [vp, P(C.c_uint32)]),
        "vr_debug_tile_order": (C.c_int, [vp, vp, vp, C.c_int32, P(C.c_int32)]),
        "vr_debug_download_vertices": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
        "vr_terrain_memory_bytes": (C.c_int, [vp, P(C.c_uint64)]),
    }
    for name in EXPORTS:
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = sig[name]
    _lib = lib
    return lib


class VrError(RuntimeError):
    def __init__(self, code, where):
        msg = ""
        try:
            msg = load_library().vr_last_error().decode()
        except Exception:
            pass
        super().__init__(f"{where} failed with status {code}: {msg}")
        self.code = code


def check(code, where):
    if code != VR_OK:
        raise VrError(code, where)
