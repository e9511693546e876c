"""The synthetic scene of SURVEY.md §8d / BASELINE.json: reference camera, ambient terms, terrain
parameters and the flythrough.  Shared by bench.py, __graft_entry__.smoke() and the tests."""
import math

import numpy as np

from . import capi as _capi

DEFAULT_EYE = (0.0, 205.0, 227.4)      # Renderer.cpp:97
DEFAULT_TARGET = (1.0, 1.8, 0.0)
AMBIENT_TOP = (0.01, 0.01, 0.01)        # Renderer.cpp:422
AMBIENT_BOTTOM = tuple(float(np.float32(0.01) * np.float32(c)) for c in (0.3, 0.4, 0.3))   # Renderer.cpp:423


def params(size, max_instances=4096):
    """TerrainPass.h:23-30 with WORLD_SIZE = SURFACE_SIZE = size."""
    p = _capi.TerrainParams()
    p.max_instances = max_instances
    p.surface_size = float(size)
    p.world_size = float(size)
    p.grid_size = 32
    p.min_lod_distance = 4.0
    p.morph_start = 0.85
    p.location[:] = [0.0, 0.0, 0.0]
    return p


def scaled_camera(cam, size):
    """Cameras are authored for the 2048 world; scale them for smaller surfaces."""
    s = size / 2048.0
    eye, tgt = cam
    return tuple(c * s for c in eye), tuple(c * s for c in tgt)


def flythrough_camera(i, n=120, radius=600.0, height=250.0):
    """Frame i of the 120-frame circle of radius 600 at y = 250 looking at the origin."""
    a = 2.0 * math.pi * (i % n) / n
    return (radius * math.cos(a), height, radius * math.sin(a)), (0.0, 0.0, 0.0)
