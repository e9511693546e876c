"""A thin ncclComm_t for a host that drives libvrterrain.so from Python (bench.py's N-rank loop, the tests): ctypes over the
RCCL that is already in the process (PyTorch ships one) or, failing that, the system's.  Plumbing only - the exchange itself
is the C ABI's (vr_frame_allgather[_ldr], vr_tonemap_allreduce_histogram, vrenderer_amd/csrc/vr_comm.hip), which resolves the
same RCCL at first use.  One process per GPU; rank 0 makes the unique id and the host hands it to the other ranks (bench.py:
a broadcast over its torch.distributed process group)."""
import ctypes as C
import os

NCCL_UNIQUE_ID_BYTES = 128


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def _loaded_rccl_path():
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                p = line.split()[-1]
                if "librccl" in os.path.basename(p):
                    return p
    except OSError:
        pass
    return None


def load():
    """The RCCL of this process: the copy already mapped (torch's), else VRTERRAIN_RCCL, else librccl.so(.1) on the loader path."""
    global _lib
    if _lib is not None:
        return _lib
    tried = []
    for cand in (_loaded_rccl_path(), os.environ.get("VRTERRAIN_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        if not cand:
            continue
        try:
            lib = C.CDLL(cand, mode=C.RTLD_GLOBAL)      # global: libvrterrain.so's dlsym(RTLD_DEFAULT, ...) finds the same copy
        except OSError as e:
            tried.append(f"{cand}: {e}")
            continue
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        _lib = lib
        return lib
    raise RuntimeError("RCCL not found: " + "; ".join(tried))


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {load().ncclGetErrorString(rc).decode()} (ncclResult_t {rc})")


def get_unique_id():
    """ncclGetUniqueId -> 128 bytes (rank 0 calls this and sends the bytes to every rank)."""
    uid = _UniqueId()
    _check(load().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return C.string_at(C.addressof(uid), NCCL_UNIQUE_ID_BYTES)


class Communicator:
    """ncclComm_t of `rank` in a group of `world` ranks (the current HIP device is the rank's).  `.handle` is what the C ABI takes."""

    def __init__(self, world, rank, unique_id_bytes):
        assert len(unique_id_bytes) == NCCL_UNIQUE_ID_BYTES
        uid = _UniqueId()
        C.memmove(C.addressof(uid), unique_id_bytes, NCCL_UNIQUE_ID_BYTES)
        h = C.c_void_p()
        _check(load().ncclCommInitRank(C.byref(h), int(world), uid, int(rank)), "ncclCommInitRank")
        self.handle, self.world, self.rank = h, int(world), int(rank)

    def close(self):
        if getattr(self, "handle", None):
            load().ncclCommDestroy(self.handle)
            self.handle = None
