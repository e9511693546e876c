"""Shared scene definitions for the parity tests."""
import numpy as np

import vrenderer_amd as vr

from vrenderer_amd.scene import (AMBIENT_BOTTOM, AMBIENT_TOP, DEFAULT_EYE, DEFAULT_TARGET, flythrough_camera,  # noqa: F401
                                 params, scaled_camera)

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
