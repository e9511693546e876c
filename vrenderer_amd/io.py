"""Asset ingest for the path's two textures (Renderer.cpp:51-55).

The reference loads `/media/terrain_heightmap.png` (sRGB = false) and `/media/terrain_albedo.png`
(sRGB = true) through Donut's TextureCache; QuadTree::GetHeightValue then reads the heightmap as one
byte per texel (QuadTree.cpp:153-162), i.e. an 8-bit single-channel image.  These helpers decode PNG
files into exactly the byte arrays `TerrainPass.Init` / `vr_terrain_create` take.
"""
import numpy as np


def load_heightmap_png(path):
    """8-bit single-channel heightmap -> (H, W) uint8.  Multi-channel images use their first channel;
    16-bit images are reduced to their high byte."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("I;16", "I;16B", "I"):
        a = np.asarray(im).astype(np.uint32)
        return np.ascontiguousarray((a >> 8).clip(0, 255).astype(np.uint8))
    if im.mode != "L":
        im = im.split()[0]
    return np.ascontiguousarray(np.asarray(im, np.uint8))


def load_albedo_png(path):
    """Colour texture -> (H, W, 4) uint8 sRGB-encoded RGBA (alpha 255 when absent)."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGBA"), np.uint8))


def save_png(path, array):
    from PIL import Image
    a = np.asarray(array)
    Image.fromarray(a, "L" if a.ndim == 2 else ("RGBA" if a.shape[2] == 4 else "RGB")).save(path)
