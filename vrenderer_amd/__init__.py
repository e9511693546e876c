"""vrenderer_amd — MI355X-native terrain + deferred-shading hot path of Viictor/vrenderer.

The package is a thin host-side mirror of the reference's interface for this path
(`TerrainPass`, `QuadTree` selection, `RenderTargets`, `DeferredLightingPass`) over the
C ABI of `libvrterrain.so` (include/vrterrain.h).  All compute is hand-written HIP for
gfx950; there is no CPU fallback.
"""
from .capi import (Instance, Light, Partition, RenderParams, ShadowParams, TerrainParams, TonemapParams, View, VrError,  # noqa: F401
                   VR_LIGHT_DIRECTIONAL, VR_LIGHT_POINT, VR_LIGHT_SPOT, VR_MAX_LODS, VR_OWNER_TILE, load_library)
from .passes import (CascadedShadowMap, Context, DeferredLightingPass, Frame, HdrImage, LdrImage, default_shadow_params, RenderTargets, TerrainPass, TiledDeferredLightingPass,  # noqa: F401
                     ToneMappingPass, default_render_params, default_terrain_params, default_tonemap_params, directional_light, make_view,
                     light_array, point_light, reference_sun, spot_light, synth_albedo, synth_heightmap, synthetic_point_lights)
