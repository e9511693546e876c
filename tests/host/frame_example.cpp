// A frame through the C++ host mirror (include/vrterrain.hpp), written the way the reference's
// Renderer drives the path: Renderer::Renderer (Renderer.cpp:51-66,97), RenderScene (:207-224),
// RecordCommand (:333-372, :382, :401-415, :417-428, :430-431).  Compiled with g++ by tests/test_abi_cpu.py (the header
// is valid C++ and the library links) and run on the GPU box by tests/test_gpu_parity.py.
#include <vrterrain.hpp>

#include <cstring>
#include <vector>

using namespace vRenderer;

int main(int argc, char** argv)
{
    const bool requireGpu = argc > 1 && !std::strcmp(argv[1], "--require-gpu");
    bool noDevice = false;
    Log() = [&](const char* m) { if (std::strstr(m, "no HIP device")) noDevice = true; std::fprintf(stderr, "%s\n", m); };

    Device device(0);
    if (!device) { std::printf("no device\n"); return (requireGpu || !noDevice) ? 3 : 0; }

    // media/terrain_heightmap.png and terrain_albedo.png are not in the reference checkout: synthetic stand-ins
    const int size = 256;
    std::vector<uint8_t> heightmap((size_t)size * size), albedo((size_t)size * size * 4);
    if (!Check(vr_synth_heightmap(device.Get(), size, 1337, heightmap.data()), "vr_synth_heightmap")) return 2;
    if (!Check(vr_synth_albedo(device.Get(), size, 4242, heightmap.data(), albedo.data()), "vr_synth_albedo")) return 2;

    // Renderer::Renderer: m_TerrainPass->Init(...) (Renderer.cpp:65-66)
    TerrainPass terrainPass(device);
    TerrainPass::CreateParameters createParams;
    createParams.terrain.surface_size = createParams.terrain.world_size = (float)size;
    if (!terrainPass.Init(createParams, heightmap.data(), size, size, albedo.data(), size, size)) return 2;

    // RenderScene: m_RenderTargets->Init(...) on resize (Renderer.cpp:207-222)
    RenderTargets renderTargets;
    const int width = 320, height = 180;
    if (renderTargets.IsUpdateRequired(width, height) && !renderTargets.Init(device, width, height)) return 2;

    // UpdateView: LookAt((0,205,227.4),(1,1.8,0)), perspProjD3DStyle(60 deg, aspect, 0.1, 10000) (Renderer.cpp:97,312-319)
    const float s = size / 2048.0f;
    const float eye[3] = { 0.0f, 205.0f * s, 227.4f * s }, target[3] = { 1.0f * s, 1.8f * s, 0.0f }, up[3] = { 0, 1, 0 };
    vr_view view;
    if (!Check(vr_view_from_camera(eye, target, up, 1.04719755f, 0.1f, 10000.0f, width, height, &view), "vr_view_from_camera")) return 2;

    // SceneLoaded: the "Sun" (Renderer.cpp:133-146)
    std::vector<vr_light> lights(1);
    std::memset(&lights[0], 0, sizeof(vr_light));
    lights[0].type = VR_LIGHT_DIRECTIONAL;
    lights[0].direction[0] = -0.9188f; lights[0].direction[1] = -0.2552f; lights[0].direction[2] = 0.3573f;
    lights[0].color[0] = lights[0].color[1] = lights[0].color[2] = 1.0f;
    lights[0].intensity = 1.0f; lights[0].angular_size_or_inv_range = 0.00925f;   // 0.53 degrees

    // RecordCommand
    EditorParams editorParams;
    lights[0].out_of_bounds_shadow = 1.0f;
    CascadedShadowMap shadowMap(device, 512, (float)size);                         // Renderer.cpp:83 (2048 there)
    shadowMap.Params().depth_bias = 0.002f;
    {                                                                              // "Cascade ShadowMap", :333-372
        if (!shadowMap.SetupForPlanarViewStable(lights[0], view)) return 2;
        shadowMap.Clear();
        TerrainPass::RenderParams shadowParams;
        shadowParams.depthOnly = true;
        if (!terrainPass.Render(shadowMap.GetView(), &shadowMap.GetView(), shadowMap.Framebuffer(), shadowParams, editorParams)) return 2;
    }
    renderTargets.Clear();                                                         // :382
    if (editorParams.m_RenderTerrain) {                                            // :401-415
        TerrainPass::RenderParams renderParams;
        renderParams.wireframe = editorParams.m_Wireframe;
        renderParams.lockView = editorParams.m_LockView;
        if (!terrainPass.Render(view, &view, renderTargets, renderParams, editorParams)) return 2;
    }
    DeferredLightingPass deferredLightingPass(device);
    deferredLightingPass.Init();
    {                                                                              // :417-428
        DeferredLightingPass::Inputs deferredInputs;
        deferredInputs.SetGBuffer(renderTargets);
        for (int c = 0; c < 3; c++) deferredInputs.ambientColorTop[c] = editorParams.m_AmbientIntensity;
        const float k[3] = { 0.3f, 0.4f, 0.3f };
        for (int c = 0; c < 3; c++) deferredInputs.ambientColorBottom[c] = deferredInputs.ambientColorTop[c] * k[c];
        deferredInputs.lights = &lights;
        deferredInputs.shadowMap = &shadowMap;                                     // m_DirectionalLight->shadowMap, :336
        if (!deferredLightingPass.Render(view, deferredInputs)) return 2;
    }
    ToneMappingPass toneMappingPass(device);                                       // CreateRenderPasses, :256-257
    toneMappingPass.AdvanceFrame(1.0f / 60.0f);                                    // Animate, :188-189
    if (!toneMappingPass.SimpleRender(ToneMappingPass::ToneMappingParameters(), renderTargets)) return 2;   // :430-431
    terrainPass.UpdateNumChunks(editorParams);                                     // ImGui "Num instances" (Renderer.cpp:468)

    // QuadTree facade: the selection the frame used, in m_SelectedNodes order
    auto& quadTree = *terrainPass.GetQuadTrees()[0];
    quadTree.NodeSelect(view, editorParams.m_MaxHeight);
    const auto ranges = quadTree.GetLodRanges();

    std::vector<uint16_t> hdr((size_t)width * height * 4);
    std::vector<float> depth((size_t)width * height);
    if (!Check(vr_image_download(renderTargets.HdrColor(), hdr.data(), hdr.size() * 2), "vr_image_download")) return 2;
    if (!Check(vr_gbuffer_download(renderTargets.GBufferFramebuffer(), 0, depth.data(), depth.size() * 4), "vr_gbuffer_download")) return 2;
    std::vector<uint8_t> ldr;
    if (!renderTargets.DownloadLdrColor(ldr)) return 2;
    size_t covered = 0, lit = 0, shown = 0;
    for (size_t i = 0; i < depth.size(); i++) { covered += depth[i] < 1.0f; lit += hdr[i * 4] != 0; shown += ldr[i * 4 + 1] > 0 && ldr[i * 4 + 3] == 255; }
    std::printf("chunks=%u selected=%zu lods=%d range0=%g covered=%zu lit=%zu shown=%zu\n", editorParams.m_NumChunks,
                quadTree.GetSelectedNodes().size(), quadTree.GetNumLods(), ranges[0], covered, lit, shown);
    const bool ok = editorParams.m_NumChunks > 0 && quadTree.GetSelectedNodes().size() == editorParams.m_NumChunks
                 && ranges[0] == 4.0f && covered > 0 && lit > 0 && shown > 0;
    return ok ? 0 : 4;
}
