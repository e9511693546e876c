"""Scratch: build libvrterrain.so with extra hipcc flags into vrenderer_amd/lib/variants/NAME/."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vrenderer_amd import build as B  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
# switches that change the image or instrument a kernel only compile in an experiment build (csrc/vr_experiments.h)
if any(f.split("=", 1)[-1].startswith(("-DVR_EXP_", "-DVR_RASTER_PROFILE", "-DVR_SELECT_PROFILE")) for f in extra) and "-DVR_EXPERIMENT_BUILD" not in extra:
    extra = extra + ["-DVR_EXPERIMENT_BUILD"]
out = os.path.join(B.LIB_DIR, "variants", name)
os.makedirs(out, exist_ok=True)
objs = []
for src in B.SOURCES:
    o = os.path.join(out, src.replace(".hip", ".o"))
    per_file = [f.split("=", 1)[1] for f in extra if f.startswith(src + "=")]
    glob = [f for f in extra if "=" not in f or not f.split("=", 1)[0].endswith(".hip")]
    subprocess.run([B._hipcc(), *B.FLAGS, *B.PER_FILE_FLAGS.get(src, []), *glob, *per_file, "-c", os.path.join(B.CSRC, src), "-o", o], check=True)
    objs.append(o)
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libvrterrain.so"), *objs, "-ldl"], check=True)
print(os.path.join(out, "libvrterrain.so"))
