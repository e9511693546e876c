// Multi-GPU frame assembly through the C ABI: the RCCL all-gather of a rank's packed tiles on the
// context's stream followed by the de-tile kernel, and the one real exchange step of the tone mapper (all-reduce of
// the 256 histogram bins).  One process per GPU; the communicator is the host's (ncclCommInitRank, or the one its
// framework created).
//
// RCCL is not a link-time dependency of libvrterrain.so: the three entry points are resolved at first use from the
// RCCL that is already in the process - the one the caller's ncclComm_t came from - and only then from librccl.so on
// the loader path.  A host that never goes multi-GPU never loads RCCL, and a host that brings its own build (PyTorch
// ships one) does not end up with two.
#include "vr_internal.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <mutex>

namespace {
// rccl.h: ncclDataType_t / ncclRedOp_t values used here
constexpr int kNcclUint8 = 1, kNcclUint32 = 3, kNcclSum = 0;
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*ErrStrFn)(int);

// Resolved exactly once (std::call_once): a host may drive its ranks as threads of one process (tests/host/
// frame_allgather_example.cpp), and every one of them comes through rccl_load() on its first frame.  The pointers are
// written inside the once-block only and read after it has returned, so no thread can see a half-filled table.
struct Rccl {
    std::once_flag once;
    AllGatherFn all_gather = nullptr;
    AllReduceFn all_reduce = nullptr;
    ErrStrFn err_str = nullptr;
};
Rccl g_rccl;

void* find_symbol(const char* name)
{
    if (void* p = dlsym(RTLD_DEFAULT, name)) return p;
    const char* env = getenv("VRTERRAIN_RCCL");
    const char* names[] = { env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (int pass = 0; pass < 2; pass++)                       // first a copy that is already loaded, then a fresh load
        for (const char* n : names) {
            if (!n || !*n) continue;
            void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (!h) continue;
            if (void* p = dlsym(h, name)) return p;
        }
    return nullptr;
}

int rccl_load()
{
    std::call_once(g_rccl.once, [] {
        g_rccl.all_gather = (AllGatherFn)find_symbol("ncclAllGather");
        g_rccl.all_reduce = (AllReduceFn)find_symbol("ncclAllReduce");
        g_rccl.err_str = (ErrStrFn)find_symbol("ncclGetErrorString");
    });
    if (!g_rccl.all_gather || !g_rccl.all_reduce) {
        vr_set_error("RCCL not found: ncclAllGather / ncclAllReduce are neither in the process nor in librccl.so (set VRTERRAIN_RCCL to its path)");
        return VR_ERR_INVALID_ARGUMENT;
    }
    return VR_OK;
}

// The library's own events carry hipEventDisableSystemFence (device-to-device ordering on one GPU).  A collective hands this
// GPU's buffers to kernels that move them to peer GPUs: in front of it the stream gets ONE event record with the default flags,
// i.e. a system-scope release of everything the stream has written (a few microseconds on the exchange stream, per collective).
int system_release(vr_context* ctx)
{
    if (!ctx->ev_sysfence) VR_HIP(hipEventCreateWithFlags(&ctx->ev_sysfence, hipEventDisableTiming));
    VR_HIP(hipEventRecord(ctx->ev_sysfence, ctx->stream));
    return VR_OK;
}

int rccl_check(int rc, const char* what)
{
    if (rc == 0) return VR_OK;
    vr_set_error("%s failed: %s (ncclResult_t %d)", what, g_rccl.err_str ? g_rccl.err_str(rc) : "?", rc);
    return VR_ERR_HIP;
}
} // namespace

extern "C" VR_API int vr_frame_allgather(vr_context* ctx, void* nccl_comm, const void* packed, void* gathered, int32_t world,
                                          vr_image* frame_out)
{
    VR_REQUIRE(ctx && nccl_comm && packed && gathered && frame_out && world >= 1, "bad arguments");
    int rc = rccl_load(); if (rc) return rc;
    VR_HIP(hipSetDevice(ctx->device));
    const size_t bytes = vr_partition_packed_bytes(frame_out->w, frame_out->h, world);
    if ((rc = system_release(ctx))) return rc;
    if ((rc = rccl_check(g_rccl.all_gather(packed, gathered, bytes, kNcclUint8, nccl_comm, ctx->stream), "ncclAllGather"))) return rc;
    return vr_frame_detile(ctx, gathered, world, frame_out);
}

extern "C" VR_API int vr_frame_allgather_tiles(vr_context* ctx, void* nccl_comm, const void* packed, void* gathered, int32_t world, size_t bytes_per_rank)
{
    VR_REQUIRE(ctx && nccl_comm && packed && gathered && world >= 1 && bytes_per_rank > 0, "bad arguments");
    int rc = rccl_load(); if (rc) return rc;
    VR_HIP(hipSetDevice(ctx->device));
    if ((rc = system_release(ctx))) return rc;
    return rccl_check(g_rccl.all_gather(packed, gathered, bytes_per_rank, kNcclUint8, nccl_comm, ctx->stream), "ncclAllGather");
}

extern "C" VR_API int vr_frame_allgather_ldr(vr_context* ctx, void* nccl_comm, const void* packed_ldr, void* gathered, int32_t world,
                                              int32_t w, int32_t h, void* ldr_frame)
{
    VR_REQUIRE(ctx && nccl_comm && packed_ldr && gathered && ldr_frame && world >= 1 && w > 0 && h > 0, "bad arguments");
    int rc = rccl_load(); if (rc) return rc;
    VR_HIP(hipSetDevice(ctx->device));
    const size_t bytes = vr_partition_packed_bytes_ldr(w, h, world);
    if ((rc = system_release(ctx))) return rc;
    if ((rc = rccl_check(g_rccl.all_gather(packed_ldr, gathered, bytes, kNcclUint8, nccl_comm, ctx->stream), "ncclAllGather"))) return rc;
    return vr_frame_detile_ldr(ctx, gathered, world, w, h, ldr_frame);
}

extern "C" VR_API int vr_tonemap_allreduce_histogram(vr_tonemap* tm, void* nccl_comm)
{
    VR_REQUIRE(tm && nccl_comm, "bad arguments");
    int rc = rccl_load(); if (rc) return rc;
    vr_context* ctx = vr_tonemap_context(tm);
    VR_HIP(hipSetDevice(ctx->device));
    void* hist = vr_tonemap_histogram_device_ptr(tm);
    if ((rc = system_release(ctx))) return rc;
    return rccl_check(g_rccl.all_reduce(hist, hist, VR_TONEMAP_BINS, kNcclUint32, kNcclSum, nccl_comm, ctx->stream), "ncclAllReduce");
}
