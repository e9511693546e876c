"""Builds vrenderer_amd/lib/libvrterrain.so from csrc/*.hip with hipcc for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical
contract: kernels evaluate fp32 expressions exactly as written.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_DIR, "csrc")
LIB_DIR = os.path.join(_DIR, "lib")
OBJ_DIR = os.path.join(_DIR, "build")
LIB = os.path.join(LIB_DIR, "libvrterrain.so")
SOURCES = ["vr_host.hip", "vr_tex.hip", "vr_select.hip", "vr_raster.hip", "vr_deferred.hip", "vr_tonemap.hip", "vr_comm.hip", "vr_frame.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


# The SLP vectoriser pairs the tile pass's scalar fp32 operations into v_pk_mul/add_f32, which issue no faster than two
# scalar instructions on this part (profiles/r01_valu_issue_rates.txt) and cost register-pair moves: 4.5 % slower (A/B on one device).
# The same in the lighting kernels: the tiled pass (issue-bound) 388 -> 368 us at 8K / 1024 lights without it; k_deferred
# (HBM-bound) unchanged, 98 -> 95 VGPRs.
PER_FILE_FLAGS = {"vr_raster.hip": ["-fno-slp-vectorize"], "vr_deferred.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if c == "hipcc" or os.path.exists(c):
            return c


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(_DIR, "..", "include", "vrterrain.h"))
    headers.append(os.path.abspath(__file__))          # the flags live here
    hdr_time = _newest(headers)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            jobs.append([hipcc, *FLAGS, *PER_FILE_FLAGS.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
