// C++ caller of the C ABI, written the way Renderer::RecordCommand drives the path
// (Renderer.cpp:382 -> 401-415 -> 417-428).  Compiled by tests/test_abi_cpu.py with g++
// (proves include/vrterrain.h is a valid C/C++ header and the library links); run on the GPU
// box by tests/test_gpu_parity.py::test_cpp_host_example.
#include <vrterrain.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(call) do { int rc_ = (call); if (rc_ != VR_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, vr_last_error()); return 2; } } while (0)

int main(int argc, char** argv)
{
    const int size = 256, w = 320, h = 180;
    vr_context* ctx = nullptr;
    int rc = vr_context_create(0, &ctx);
    if (rc == VR_ERR_NO_DEVICE) { std::printf("no device: %s\n", vr_last_error()); return argc > 1 && !std::strcmp(argv[1], "--require-gpu") ? 3 : 0; }
    if (rc != VR_OK) return 2;

    std::vector<uint8_t> hm((size_t)size * size), al((size_t)size * size * 4);
    CHECK(vr_synth_heightmap(ctx, size, 1337, hm.data()));
    CHECK(vr_synth_albedo(ctx, size, 4242, hm.data(), al.data()));

    vr_terrain_params tp; vr_terrain_default_params(&tp);
    tp.surface_size = tp.world_size = (float)size;
    vr_terrain* terrain = nullptr;
    CHECK(vr_terrain_create(ctx, &tp, hm.data(), size, size, al.data(), size, size, &terrain));

    vr_gbuffer* gb = nullptr; vr_image* hdr = nullptr;
    CHECK(vr_gbuffer_create(ctx, w, h, &gb));
    CHECK(vr_image_create(ctx, w, h, nullptr, &hdr));

    const float s = size / 2048.0f;
    const float eye[3] = { 0.0f, 205.0f * s, 227.4f * s }, target[3] = { 1.0f * s, 1.8f * s, 0.0f }, up[3] = { 0, 1, 0 };
    vr_view view;
    CHECK(vr_view_from_camera(eye, target, up, 1.04719755f, 0.1f, 10000.0f, w, h, &view));

    vr_render_params rp; vr_render_default_params(&rp);
    CHECK(vr_gbuffer_clear(gb));                                            // m_RenderTargets->Clear
    CHECK(vr_terrain_render(terrain, &view, &view, gb, &rp, nullptr));      // m_TerrainPass->Render
    vr_light sun; std::memset(&sun, 0, sizeof(sun));
    sun.type = VR_LIGHT_DIRECTIONAL; sun.direction[0] = -0.9188f; sun.direction[1] = -0.2552f; sun.direction[2] = 0.3573f;
    sun.color[0] = sun.color[1] = sun.color[2] = 1.0f; sun.intensity = 1.0f; sun.angular_size_or_inv_range = 0.00925f;
    const float top[3] = { 0.01f, 0.01f, 0.01f }, bot[3] = { 0.003f, 0.004f, 0.003f };
    CHECK(vr_deferred_light(ctx, &view, gb, &sun, 1, top, bot, hdr, nullptr));   // m_DeferredLightingPass->Render

    uint32_t chunks = 0;
    CHECK(vr_terrain_num_chunks(terrain, &chunks));                         // EditorParams::m_NumChunks
    std::vector<uint16_t> px((size_t)w * h * 4);
    CHECK(vr_image_download(hdr, px.data(), px.size() * 2));
    std::vector<float> depth((size_t)w * h);
    CHECK(vr_gbuffer_download(gb, 0, depth.data(), depth.size() * 4));
    size_t covered = 0, lit = 0;
    for (size_t i = 0; i < depth.size(); i++) { covered += depth[i] < 1.0f; lit += px[i * 4] != 0; }
    std::printf("chunks=%u covered=%zu lit=%zu\n", chunks, covered, lit);
    vr_image_destroy(hdr); vr_gbuffer_destroy(gb); vr_terrain_destroy(terrain); vr_context_destroy(ctx);
    return (chunks > 0 && covered > 0 && lit > 0) ? 0 : 4;
}
