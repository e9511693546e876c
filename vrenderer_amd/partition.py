"""Screen-tile partition of a frame over the GPUs of one node.

Pure index math, mirroring vr_ensure_partition / k_detile in csrc/ (the product path
runs those on the device; this module is what host code — bench.py, tests — uses to
reason about the same layout):

  * owner tiles are 128x128 pixels, owner(tx, ty) = (tx + ty) mod world_size;
  * a rank's packed buffer is [local tile][128 rows][128 px] RGB16F (alpha is always 0 and is
    not exchanged) or, tone-mapped, RGB8 (alpha always 255), local tiles in
    row-major order of the rank's owned tiles, padded to `max_owned` tiles so that every
    rank's all-gather send count is equal;
  * after the all-gather, tile t of the frame lives at slot owner*max_owned + local.
"""
import numpy as np

TILE = 128


def owner_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def owned_tiles(width, height, rank, world):
    tx, ty = owner_grid(width, height)
    return [y * tx + x for y in range(ty) for x in range(tx) if (x + y) % world == rank]


def max_owned(width, height, world):
    return max(len(owned_tiles(width, height, r, world)) for r in range(world))


def tile_slots(width, height, world):
    """slot[tile] = owner * max_owned + local index, for the gathered buffer."""
    tx, ty = owner_grid(width, height)
    mo = max_owned(width, height, world)
    nxt = [0] * world
    slot = np.zeros(tx * ty, np.int64)
    for y in range(ty):
        for x in range(tx):
            o = (x + y) % world
            slot[y * tx + x] = o * mo + nxt[o]
            nxt[o] += 1
    return slot


def packed_shape(width, height, world):
    return (max_owned(width, height, world), TILE, TILE, 3)


def pack(frame, rank, world):
    """frame: (H, W, 4) uint16 -> this rank's packed tiles (max_owned, 128, 128, 3), zero padded."""
    h, w = frame.shape[:2]
    tx, _ = owner_grid(w, h)
    out = np.zeros(packed_shape(w, h, world), frame.dtype)
    for i, t in enumerate(owned_tiles(w, h, rank, world)):
        y0, x0 = (t // tx) * TILE, (t % tx) * TILE
        blk = frame[y0:y0 + TILE, x0:x0 + TILE, :3]
        out[i, :blk.shape[0], :blk.shape[1]] = blk
    return out


def detile(gathered, width, height, world, alpha=0):
    """gathered: (world * max_owned, 128, 128, 3) -> (H, W, 4) frame; the alpha that was not exchanged is filled in
    (0 for HdrColor, 255 for the tone-mapped LdrColor)."""
    tx, ty = owner_grid(width, height)
    slot = tile_slots(width, height, world)
    g = gathered.reshape(-1, TILE, TILE, 3)
    out = np.zeros((height, width, 4), gathered.dtype)
    out[..., 3] = alpha
    for t in range(tx * ty):
        y0, x0 = (t // tx) * TILE, (t % tx) * TILE
        hh, ww = min(TILE, height - y0), min(TILE, width - x0)
        out[y0:y0 + hh, x0:x0 + ww, :3] = g[slot[t], :hh, :ww]
    return out
