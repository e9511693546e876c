// An N-rank frame through the C ABI with the exchange the library exports (include/vrterrain.h: vr_frame_allgather_ldr,
// vr_tonemap_allreduce_histogram): what a C++ Renderer host adds for a node of GPUs.  Ranks are threads of one process
// here (one per visible device, or --ranks N capped by the device count) so that the example needs no launcher; a
// production host runs one process per GPU and creates the communicator with ncclGetUniqueId / ncclCommInitRank.
// Per rank and frame: shadow-less RecordCommand (Renderer.cpp:382, 401-415, 417-428, 430-431) on the rank's screen tiles,
// histogram all-reduce, tone map to packed RGB8 tiles, all-gather + de-tile.  Rank 0 then checks the assembled frame
// against its own unsplit render, byte for byte.
// Compiled by tests/test_abi_cpu.py (g++ against rccl.h, hip_runtime_api.h and vrterrain.h), run by the GPU tests.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <vrterrain.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define VR(call) do { int rc_ = (call); if (rc_ != VR_OK) { std::fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #call, rc_, vr_last_error()); return 2; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s -> %s\n", rank, #call, hipGetErrorString(e_)); return 2; } } while (0)
#define NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { std::fprintf(stderr, "rank %d: %s -> %s\n", rank, #call, ncclGetErrorString(r_)); return 2; } } while (0)

static const int kSize = 256, kW = 640, kH = 360, kFrames = 3;

struct Shared { std::vector<uint8_t> heightmap, albedo; ncclUniqueId id; int world; };

static void sun(vr_light* l)
{
    std::memset(l, 0, sizeof(*l));
    l->type = VR_LIGHT_DIRECTIONAL;
    l->direction[0] = -0.9188f; l->direction[1] = -0.2552f; l->direction[2] = 0.3573f;
    l->color[0] = l->color[1] = l->color[2] = 1.0f; l->intensity = 1.0f; l->angular_size_or_inv_range = 0.00925f;
}

static int rank_main(int rank, const Shared* sh, int* verdict)
{
    const int world = sh->world;
    HIP(hipSetDevice(rank));
    ncclComm_t comm;
    NCCL(ncclCommInitRank(&comm, world, sh->id, rank));
    hipStream_t stream;
    HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    vr_context* ctx; VR(vr_context_create(rank, &ctx));
    VR(vr_context_set_stream(ctx, stream));
    vr_terrain_params tp; vr_terrain_default_params(&tp);
    tp.surface_size = tp.world_size = (float)kSize;
    vr_terrain* terrain;
    VR(vr_terrain_create(ctx, &tp, sh->heightmap.data(), kSize, kSize, sh->albedo.data(), kSize, kSize, &terrain));
    vr_gbuffer* gb; VR(vr_gbuffer_create(ctx, kW, kH, &gb));
    vr_tonemap* tm; VR(vr_tonemap_create(ctx, &tm));
    vr_tonemap_params tmp; vr_tonemap_default_params(&tmp);
    const vr_partition part = { rank, world };
    const size_t packed_hdr = vr_partition_packed_bytes(kW, kH, world), packed_ldr = vr_partition_packed_bytes_ldr(kW, kH, world);
    void *d_packed_ldr, *d_gathered, *d_frame;
    HIP(hipMalloc(&d_packed_ldr, packed_ldr));
    HIP(hipMalloc(&d_gathered, packed_ldr * world)); HIP(hipMalloc(&d_frame, (size_t)kW * kH * 4));
    // the packed tile buffer as an image: 128 pixels wide, as many rows as the bytes need at 8 B per pixel
    vr_image* hdr_tiles;
    VR(vr_image_create(ctx, VR_OWNER_TILE, (int32_t)((packed_hdr + VR_OWNER_TILE * 8 - 1) / (VR_OWNER_TILE * 8)), nullptr, &hdr_tiles));

    vr_light light; sun(&light);
    const float amb_top[3] = { 0.01f, 0.01f, 0.01f }, amb_bot[3] = { 0.003f, 0.004f, 0.003f }, up[3] = { 0, 1, 0 };
    vr_render_params rp; vr_render_default_params(&rp);
    vr_view view;
    for (int f = 0; f < kFrames; f++) {
        const float a = 6.2831853f * f / 120.0f, s = kSize / 2048.0f;
        const float eye[3] = { 600.0f * s * cosf(a), 250.0f * s, 600.0f * s * sinf(a) }, target[3] = { 0, 0, 0 };
        VR(vr_view_from_camera(eye, target, up, 1.04719755f, 0.1f, 10000.0f, kW, kH, &view));
        VR(vr_gbuffer_clear(gb));                                                            // Renderer.cpp:382
        VR(vr_terrain_render(terrain, &view, &view, gb, &rp, &part));                        // :401-415, owned tiles only
        VR(vr_deferred_light(ctx, &view, gb, &light, 1, amb_top, amb_bot, hdr_tiles, &part)); // :417-428 -> packed RGB16F tiles
        VR(vr_tonemap_reset_histogram(tm));                                                  // :430-431, split over the ranks
        VR(vr_tonemap_add_frame_to_histogram(tm, &tmp, hdr_tiles, kW, kH, &part));
        VR(vr_tonemap_allreduce_histogram(tm, comm));
        VR(vr_tonemap_compute_exposure(tm, &tmp, 1.0f / 60.0f));
        VR(vr_tonemap_render(tm, &tmp, hdr_tiles, kW, kH, d_packed_ldr, packed_ldr, &part));
        VR(vr_frame_allgather_ldr(ctx, comm, d_packed_ldr, d_gathered, world, kW, kH, d_frame));
    }
    VR(vr_context_synchronize(ctx));

    if (rank == 0) {        // the same last frame unsplit, on this rank's device
        std::vector<uint8_t> got((size_t)kW * kH * 4), want((size_t)kW * kH * 4);
        HIP(hipMemcpy(got.data(), d_frame, got.size(), hipMemcpyDeviceToHost));
        vr_tonemap* tm1; VR(vr_tonemap_create(ctx, &tm1));
        vr_image* hdr; VR(vr_image_create(ctx, kW, kH, nullptr, &hdr));
        void* d_ldr; HIP(hipMalloc(&d_ldr, want.size()));
        for (int f = 0; f < kFrames; f++) {      // the adapted luminance has a history: replay all frames
            const float a = 6.2831853f * f / 120.0f, s = kSize / 2048.0f;
            const float eye[3] = { 600.0f * s * cosf(a), 250.0f * s, 600.0f * s * sinf(a) }, target[3] = { 0, 0, 0 };
            VR(vr_view_from_camera(eye, target, up, 1.04719755f, 0.1f, 10000.0f, kW, kH, &view));
            VR(vr_gbuffer_clear(gb));
            VR(vr_terrain_render(terrain, &view, &view, gb, &rp, nullptr));
            VR(vr_deferred_light(ctx, &view, gb, &light, 1, amb_top, amb_bot, hdr, nullptr));
            VR(vr_tonemap_simple_render(tm1, &tmp, 1.0f / 60.0f, hdr, d_ldr, want.size()));
        }
        VR(vr_context_synchronize(ctx));
        HIP(hipMemcpy(want.data(), d_ldr, want.size(), hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t i = 0; i < got.size(); i++) diff += got[i] != want[i];
        std::printf("ranks=%d frame=%dx%d assembled-vs-unsplit differing bytes=%zu\n", world, kW, kH, diff);
        *verdict = diff == 0 ? 0 : 1;
        (void)hipFree(d_ldr); vr_image_destroy(hdr); vr_tonemap_destroy(tm1);
    }
    (void)hipFree(d_packed_ldr); (void)hipFree(d_gathered); (void)hipFree(d_frame);
    vr_image_destroy(hdr_tiles); vr_tonemap_destroy(tm); vr_gbuffer_destroy(gb); vr_terrain_destroy(terrain);
    vr_context_destroy(ctx);
    (void)hipStreamDestroy(stream);
    ncclCommDestroy(comm);
    return 0;
}

int main(int argc, char** argv)
{
    bool require_gpu = false; int want_ranks = 0;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--require-gpu")) require_gpu = true;
        else if (!std::strcmp(argv[i], "--ranks") && i + 1 < argc) want_ranks = std::atoi(argv[++i]);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { std::printf("no device\n"); return require_gpu ? 3 : 0; }
    Shared sh;
    sh.world = want_ranks > 0 && want_ranks < ndev ? want_ranks : ndev;
    if (sh.world > 8) sh.world = 8;
    {   // media/terrain_heightmap.png and terrain_albedo.png are not in the reference checkout: synthetic stand-ins
        vr_context* c; int rank = 0;
        VR(vr_context_create(0, &c));
        sh.heightmap.resize((size_t)kSize * kSize); sh.albedo.resize((size_t)kSize * kSize * 4);
        VR(vr_synth_heightmap(c, kSize, 1337, sh.heightmap.data()));
        VR(vr_synth_albedo(c, kSize, 4242, sh.heightmap.data(), sh.albedo.data()));
        vr_context_destroy(c);
    }
    if (ncclGetUniqueId(&sh.id) != ncclSuccess) { std::fprintf(stderr, "ncclGetUniqueId failed\n"); return 2; }
    int verdict = 0;
    std::vector<int> rcs(sh.world, 0);
    std::vector<std::thread> threads;
    for (int r = 0; r < sh.world; r++) threads.emplace_back([&, r] { rcs[r] = rank_main(r, &sh, &verdict); });
    for (auto& t : threads) t.join();
    for (int r = 0; r < sh.world; r++) if (rcs[r]) return rcs[r];
    return verdict;
}
