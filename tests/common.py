"""Shared scene definitions for the parity tests."""
import numpy as np

import vrenderer_amd as vr

DEFAULT_EYE = (0.0, 205.0, 227.4)      # Renderer.cpp:97
DEFAULT_TARGET = (1.0, 1.8, 0.0)
AMBIENT_TOP = (0.01, 0.01, 0.01)        # Renderer.cpp:422
AMBIENT_BOTTOM = tuple(float(np.float32(0.01) * np.float32(c)) for c in (0.3, 0.4, 0.3))   # Renderer.cpp:423

# a handful of cameras: the reference default, the flythrough circle, low/grazing, inside terrain
CAMERAS = [
    (DEFAULT_EYE, DEFAULT_TARGET),
    ((600.0, 250.0, 0.0), (0.0, 0.0, 0.0)),
    ((0.0, 250.0, -600.0), (0.0, 0.0, 0.0)),
    ((-424.26, 250.0, 424.26), (0.0, 0.0, 0.0)),
    ((10.0, 900.0, 10.0), (0.0, 0.0, 1.0)),
    ((-300.0, 120.0, 80.0), (200.0, 60.0, -40.0)),
    ((50.0, 40.0, -20.0), (300.0, 80.0, 200.0)),
    ((900.0, 500.0, 900.0), (0.0, 0.0, 0.0)),
]


def params(size, max_instances=4096):
    p = vr.TerrainParams()
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
