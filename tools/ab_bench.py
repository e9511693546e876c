"""Scratch A/B helper: run bench.py against an alternative build of the library.

    python tools/build_variant.py NAME [extra hipcc flags...]      -> vrenderer_amd/lib/variants/NAME/libvrterrain.so
    python tools/ab_bench.py NAME [bench.py flags...]
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1]
from vrenderer_amd import capi  # noqa: E402

if name != "default":
    capi.LIB_PATH = os.path.join(ROOT, "vrenderer_amd", "lib", "variants", name, "libvrterrain.so")
    assert os.path.exists(capi.LIB_PATH), capi.LIB_PATH
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
