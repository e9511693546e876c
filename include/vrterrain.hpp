// vrterrain.hpp — C++ host mirror of the reference's interface for the hot path, over the C ABI
// of vrterrain.h.  Header-only, no dependencies beyond the C++17 standard library.
//
// Names, argument meaning and error behaviour follow the reference so that a caller written
// against vRenderer::TerrainPass / QuadTree / RenderTargets / DeferredLightingPass reads the same:
//   vRenderer::TerrainPass      source/terrain/TerrainPass.h:32-159   (Init, Render, GetQuadTrees)
//   QuadTree                    source/terrain/QuadTree.h:64-127      (NodeSelect, GetSelectedNodes,
//                                                                      ClearSelectedNodes, GetNumLods, GetLodRanges)
//   vRenderer::RenderTargets    source/Renderer.h:50-110              (Init, Clear, IsUpdateRequired)
//   DeferredLightingPass        as called at source/Renderer.cpp:239-240,417-428 (Render(view, Inputs))
//   CascadedShadowMap           as called at source/Renderer.cpp:83-87,333-372 (SetupForPlanarViewStable, Clear, GetView)
//   ToneMappingPass             as called at source/Renderer.cpp:188-189,256-257,430-431 (AdvanceFrame, SimpleRender)
// Like the reference, nothing here throws: methods return bool / log through a callback
// (donut::log in the reference, TerrainPass.cpp:224-227, QuadTree.cpp:39).
#pragma once

#include "vrterrain.h"

#include <array>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <vector>

namespace vRenderer
{
    using LogFn = std::function<void(const char*)>;
    inline LogFn& Log() { static LogFn fn = [](const char* m) { std::fprintf(stderr, "[vrterrain] %s\n", m); }; return fn; }
    inline bool Check(int rc, const char* what)
    {
        if (rc == VR_OK) return true;
        char buf[640];
        std::snprintf(buf, sizeof(buf), "%s failed (%d): %s", what, rc, vr_last_error());
        Log()(buf);
        return false;
    }

    // nvrhi::IDevice + the frame's command list (main.cpp:57-61, Renderer.cpp:48)
    class Device
    {
        vr_context* m_Ctx = nullptr;
    public:
        explicit Device(int ordinal = 0) { Check(vr_context_create(ordinal, &m_Ctx), "vr_context_create"); }
        ~Device() { vr_context_destroy(m_Ctx); }
        Device(const Device&) = delete;
        Device& operator=(const Device&) = delete;
        explicit operator bool() const { return m_Ctx != nullptr; }
        vr_context* Get() const { return m_Ctx; }
        void SetStream(void* hipStream) { vr_context_set_stream(m_Ctx, hipStream); }
        void WaitForIdle() { vr_context_synchronize(m_Ctx); }            // nvrhi::IDevice::waitForIdle
    };

    // EditorParams (Renderer.h:34-48): what the path reads and writes
    struct EditorParams
    {
        bool m_RenderTerrain = true;
        bool m_Wireframe = false;
        bool m_LockView = false;
        float m_MaxHeight = 400.0f;
        uint32_t m_NumChunks = 0;
        float m_AmbientIntensity = 0.01f;
    };

    // RenderTargets : GBufferRenderTargets (Renderer.h:50-110)
    class RenderTargets
    {
        vr_gbuffer* m_GBuffer = nullptr;
        vr_image* m_Hdr = nullptr;
        vr_ldr_image* m_Ldr = nullptr;             // LdrColor SRGBA8 (Renderer.h:81-92)
        int m_Width = 0, m_Height = 0;
        void Release() { vr_ldr_image_destroy(m_Ldr); vr_image_destroy(m_Hdr); vr_gbuffer_destroy(m_GBuffer); m_Ldr = nullptr; m_Hdr = nullptr; m_GBuffer = nullptr; }
    public:
        ~RenderTargets() { Release(); }
        bool Init(Device& device, int width, int height)
        {
            Release();
            m_Width = width; m_Height = height;
            return Check(vr_gbuffer_create(device.Get(), width, height, &m_GBuffer), "vr_gbuffer_create")
                && Check(vr_image_create(device.Get(), width, height, nullptr, &m_Hdr), "vr_image_create")
                && Check(vr_ldr_image_create(device.Get(), width, height, 0, nullptr, &m_Ldr), "vr_ldr_image_create");
        }
        void* LdrColor() const { return vr_ldr_image_device_ptr(m_Ldr); }                    // Renderer.h:81-92
        size_t LdrColorBytes() const { return vr_ldr_image_capacity(m_Ldr); }
        bool DownloadLdrColor(std::vector<uint8_t>& out) const
        {
            out.resize(LdrColorBytes());
            return Check(vr_ldr_image_download(m_Ldr, out.data(), out.size()), "vr_ldr_image_download");
        }
        [[nodiscard]] bool IsUpdateRequired(int width, int height) const { return width != m_Width || height != m_Height; }
        void Clear() { Check(vr_gbuffer_clear(m_GBuffer), "vr_gbuffer_clear"); }            // Renderer.cpp:382
        vr_gbuffer* GBufferFramebuffer() const { return m_GBuffer; }
        vr_image* HdrColor() const { return m_Hdr; }
        int Width() const { return m_Width; }
        int Height() const { return m_Height; }
    };

    class TerrainPass;

    // QuadTree (QuadTree.h:64-127).  The tree itself is implicit on the device; this object is the
    // reference's view of it: selection results of the last NodeSelect, LOD ranges, LOD count.
    class QuadTree
    {
        friend class TerrainPass;
        vr_terrain* m_Terrain = nullptr;
        std::vector<uint32_t> m_SelectedNodes;      // node ids in m_SelectedNodes order
        std::vector<vr_instance> m_Instances;       // what UpdateTransforms wrote for them
    public:
        static constexpr int MAX_LODS = VR_MAX_LODS;
        // NodeSelect(viewOrigin, root, numLods, frustum, maxHeight) + UpdateTransforms (TerrainPass.cpp:181-183)
        bool NodeSelect(const vr_view& view, float maxHeight)
        {
            vr_terrain_params p; vr_terrain_default_params(&p);
            m_SelectedNodes.assign(4096, 0u); m_Instances.assign(4096, vr_instance{});
            uint32_t n = 0;
            const bool ok = Check(vr_terrain_select(m_Terrain, &view, maxHeight, m_SelectedNodes.data(), m_Instances.data(), &n),
                                  "vr_terrain_select");
            m_SelectedNodes.resize(n); m_Instances.resize(n);
            return ok;
        }
        const std::vector<uint32_t>& GetSelectedNodes() const { return m_SelectedNodes; }
        const std::vector<vr_instance>& GetInstanceData() const { return m_Instances; }
        void ClearSelectedNodes() { m_SelectedNodes.clear(); m_Instances.clear(); }
        int GetNumLods() const { return vr_terrain_num_lods(m_Terrain); }
        std::array<float, MAX_LODS> GetLodRanges() const
        {
            std::array<float, MAX_LODS> r{};
            vr_terrain_lod_ranges(m_Terrain, r.data());
            return r;
        }
        // SetHeight over the whole tree + m_HeightLoaded (QuadTree.cpp:46-51,191-208)
        bool SetHeight(bool loaded) { return Check(vr_terrain_update_heights(m_Terrain, loaded ? 1 : 0), "vr_terrain_update_heights"); }
    };

    // vRenderer::TerrainPass (TerrainPass.h:32-159)
    class TerrainPass
    {
    public:
        struct CreateParameters { vr_terrain_params terrain; CreateParameters() { vr_terrain_default_params(&terrain); } };
        struct RenderParams { bool wireframe = false; bool lockView = false; bool depthOnly = false; };   // TerrainPass.h:68-73
    private:
        Device& m_Device;
        vr_terrain* m_Terrain = nullptr;
        std::vector<std::shared_ptr<QuadTree>> m_QuadTrees;
        float m_MaxHeight = 1.0f;
    public:
        explicit TerrainPass(Device& device) : m_Device(device) {}
        ~TerrainPass() { vr_terrain_destroy(m_Terrain); }
        TerrainPass(const TerrainPass&) = delete;
        TerrainPass& operator=(const TerrainPass&) = delete;

        // Init(shaderFactory, params, commandList, heightmapTexture, colorTexture, executor) (TerrainPass.cpp:34-141):
        // the two textures arrive as the decoded bytes Donut's TextureCache holds (R8 and SRGBA8).
        bool Init(const CreateParameters& params, const uint8_t* heightmapR8, int heightmapWidth, int heightmapHeight,
                  const uint8_t* colorSRGBA8, int colorWidth, int colorHeight)
        {
            if (!heightmapR8) { Log()("Heightmap texture data missing for QuadTree generation"); return false; }   // QuadTree.cpp:39
            vr_terrain_destroy(m_Terrain); m_Terrain = nullptr;
            if (!Check(vr_terrain_create(m_Device.Get(), &params.terrain, heightmapR8, heightmapWidth, heightmapHeight,
                                         colorSRGBA8, colorWidth, colorHeight, &m_Terrain), "vr_terrain_create"))
                return false;
            auto qt = std::make_shared<QuadTree>();
            qt->m_Terrain = m_Terrain;
            m_QuadTrees = { qt };          // all surfaces are swept together on the device; one facade covers them
            return true;
        }

        // Render(commandList, compositeView, compositeViewPrev, framebufferFactory, renderParams, editorParams)
        // (TerrainPass.cpp:143-232).  Asynchronous; editorParams.m_NumChunks is refreshed by UpdateNumChunks().
        bool Render(const vr_view& view, const vr_view* viewPrev, RenderTargets& targets, const RenderParams& renderParams,
                    EditorParams& editorParams, const vr_partition* partition = nullptr)
        {
            m_MaxHeight = editorParams.m_MaxHeight;
            vr_render_params rp; vr_render_default_params(&rp);
            rp.wireframe = renderParams.wireframe; rp.lock_view = renderParams.lockView; rp.depth_only = renderParams.depthOnly;
            rp.max_height = m_MaxHeight;
            if (!Check(vr_terrain_render(m_Terrain, &view, viewPrev ? viewPrev : &view, targets.GBufferFramebuffer(), &rp, partition),
                       "vr_terrain_render")) {
                Log()("TerrainPass::Render - Couldn't create PSO");          // TerrainPass.cpp:226, the reference's only failure path
                return false;
            }
            return true;
        }
        // editorParams.m_NumChunks = numNodes (TerrainPass.cpp:198); synchronises with the device
        bool UpdateNumChunks(EditorParams& editorParams)
        {
            uint32_t n = 0;
            const bool ok = Check(vr_terrain_num_chunks(m_Terrain, &n), "vr_terrain_num_chunks");
            editorParams.m_NumChunks = n;
            return ok;
        }
        const std::vector<std::shared_ptr<QuadTree>>& GetQuadTrees() const { return m_QuadTrees; }
        vr_terrain* Get() const { return m_Terrain; }
    };

    // donut::render::CascadedShadowMap with the one cascade the reference creates (Renderer.cpp:83-87)
    class CascadedShadowMap
    {
        Device& m_Device;
        RenderTargets m_Targets;           // m_ShadowFramebuffer: only the depth plane is used
        vr_shadow_params m_Params;
        vr_view m_View{};
    public:
        CascadedShadowMap(Device& device, int resolution, float worldSize) : m_Device(device)
        {
            vr_shadow_default_params(&m_Params, worldSize);
            m_Params.resolution = resolution;
            m_Targets.Init(device, resolution, resolution);
        }
        vr_shadow_params& Params() { return m_Params; }
        // SetupForPlanarViewStable(light, projectionFrustum, inverseViewMatrix, maxShadowDistance, zUp, zDown, exponent, 0, 1)
        bool SetupForPlanarViewStable(const vr_light& light, const vr_view& cameraView)
        {
            return Check(vr_shadow_view_setup(&light, &cameraView, &m_Params, &m_View), "vr_shadow_view_setup");
        }
        void Clear() { m_Targets.Clear(); }                                                   // Renderer.cpp:354
        const vr_view& GetView() const { return m_View; }
        RenderTargets& Framebuffer() { return m_Targets; }
        vr_shadow_binding Binding(int lightIndex) { return vr_shadow_binding{ &m_View, m_Targets.GBufferFramebuffer(), lightIndex, m_Params.depth_bias }; }
    };

    // donut::render::DeferredLightingPass as used at Renderer.cpp:239-240,417-428
    class DeferredLightingPass
    {
        Device& m_Device;
    public:
        struct Inputs
        {
            RenderTargets* gbuffer = nullptr;                       // SetGBuffer(*m_RenderTargets)
            float ambientColorTop[3] = { 0, 0, 0 };
            float ambientColorBottom[3] = { 0, 0, 0 };
            const std::vector<vr_light>* lights = nullptr;
            CascadedShadowMap* shadowMap = nullptr;                 // DirectionalLight::shadowMap (Renderer.cpp:336)
            int shadowLightIndex = 0;
            vr_image* output = nullptr;                             // HdrColor
            void SetGBuffer(RenderTargets& targets) { gbuffer = &targets; output = targets.HdrColor(); }
        };
        explicit DeferredLightingPass(Device& device) : m_Device(device) {}
        void Init() {}
        void ResetBindingCache() {}                                 // Renderer.cpp:216: nothing is cached here
        bool Render(const vr_view& view, const Inputs& inputs, const vr_partition* partition = nullptr)
        {
            static const std::vector<vr_light> none;
            const std::vector<vr_light>& l = inputs.lights ? *inputs.lights : none;
            if (inputs.shadowMap) {
                const vr_shadow_binding sb = inputs.shadowMap->Binding(inputs.shadowLightIndex);
                return Check(vr_deferred_light_shadowed(m_Device.Get(), &view, inputs.gbuffer->GBufferFramebuffer(), l.data(), (int32_t)l.size(),
                                                        inputs.ambientColorTop, inputs.ambientColorBottom, inputs.output, partition, &sb),
                             "vr_deferred_light_shadowed");
            }
            return Check(vr_deferred_light(m_Device.Get(), &view, inputs.gbuffer->GBufferFramebuffer(), l.data(), (int32_t)l.size(),
                                           inputs.ambientColorTop, inputs.ambientColorBottom, inputs.output, partition),
                         "vr_deferred_light");
        }
    };

    // donut::render::ToneMappingPass as used at Renderer.cpp:188-189,256-257,430-431
    class ToneMappingPass
    {
        vr_tonemap* m_Pass = nullptr;
        float m_FrameTime = 0.0f;
    public:
        struct ToneMappingParameters : vr_tonemap_params { ToneMappingParameters() { vr_tonemap_default_params(this); } };
        explicit ToneMappingPass(Device& device) { Check(vr_tonemap_create(device.Get(), &m_Pass), "vr_tonemap_create"); }
        ~ToneMappingPass() { vr_tonemap_destroy(m_Pass); }
        ToneMappingPass(const ToneMappingPass&) = delete;
        ToneMappingPass& operator=(const ToneMappingPass&) = delete;
        void AdvanceFrame(float frameTime) { m_FrameTime = frameTime; }                      // Renderer.cpp:188-189
        bool ResetExposure(float initialExposure = 0.0f) { return Check(vr_tonemap_reset_exposure(m_Pass, initialExposure), "vr_tonemap_reset_exposure"); }
        bool ResetHistogram() { return Check(vr_tonemap_reset_histogram(m_Pass), "vr_tonemap_reset_histogram"); }
        bool AddFrameToHistogram(const ToneMappingParameters& params, RenderTargets& targets)
        {
            return Check(vr_tonemap_add_frame_to_histogram(m_Pass, &params, targets.HdrColor(), targets.Width(), targets.Height(), nullptr),
                         "vr_tonemap_add_frame_to_histogram");
        }
        bool ComputeExposure(const ToneMappingParameters& params) { return Check(vr_tonemap_compute_exposure(m_Pass, &params, m_FrameTime), "vr_tonemap_compute_exposure"); }
        bool Render(const ToneMappingParameters& params, RenderTargets& targets)
        {
            return Check(vr_tonemap_render(m_Pass, &params, targets.HdrColor(), targets.Width(), targets.Height(), targets.LdrColor(),
                                           targets.LdrColorBytes(), nullptr), "vr_tonemap_render");
        }
        // SimpleRender(commandList, params, compositeView, sourceTexture) (Renderer.cpp:431)
        bool SimpleRender(const ToneMappingParameters& params, RenderTargets& targets)
        {
            return Check(vr_tonemap_simple_render(m_Pass, &params, m_FrameTime, targets.HdrColor(), targets.LdrColor(), targets.LdrColorBytes()),
                         "vr_tonemap_simple_render");
        }
        vr_tonemap* Get() const { return m_Pass; }
    };
} // namespace vRenderer
