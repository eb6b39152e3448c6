"""Screen-space tile bookkeeping for the multi-GPU partition (host side, numpy).

The frame is cut into tile_w x tile_h tiles numbered row-major from the
bottom-left corner (row 0 of the frame buffer is the bottom scanline,
raytracer.c:81).  Owner `r` of `n` renders tiles r, r+n, r+2n, ... - a cyclic
assignment, because the geometry sits mid-frame and contiguous slabs would be
unbalanced.  The packed per-owner buffer is [owned_tiles, tile_h, tile_w, 4]
float32; edge tiles are padded.  The HIP pack/unpack kernels use the same
mapping (rtamd_tiles_pack / rtamd_tiles_unpack).
"""
import numpy as np


def grid(width, height, tile_w, tile_h):
    return (width + tile_w - 1) // tile_w, (height + tile_h - 1) // tile_h


def owned_tiles(width, height, tile_w, tile_h, first, stride):
    tx, ty = grid(width, height, tile_w, tile_h)
    return list(range(first, tx * ty, stride))


def owned_count(width, height, tile_w, tile_h, first, stride):
    tx, ty = grid(width, height, tile_w, tile_h)
    total = tx * ty
    return (total - first + stride - 1) // stride if first < total else 0


def max_owned_count(width, height, tile_w, tile_h, stride):
    """Owner 0 always holds the most tiles; gathers are padded to this."""
    return owned_count(width, height, tile_w, tile_h, 0, stride)


def pack(frame, tile_w, tile_h, first, stride):
    """frame [h, w, 4] -> [owned, tile_h, tile_w, 4] (numpy reference of rtamd_tiles_pack)."""
    h, w, _ = frame.shape
    tx, _ = grid(w, h, tile_w, tile_h)
    tiles = owned_tiles(w, h, tile_w, tile_h, first, stride)
    out = np.zeros((len(tiles), tile_h, tile_w, 4), frame.dtype)
    for k, t in enumerate(tiles):
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        x1, y1 = min(x0 + tile_w, w), min(y0 + tile_h, h)
        out[k, : y1 - y0, : x1 - x0] = frame[y0:y1, x0:x1]
    return out


def unpack(packed, frame, tile_w, tile_h, first, stride):
    """Inverse of pack: writes the owner's tiles into frame [h, w, 4] in place."""
    h, w, _ = frame.shape
    tx, _ = grid(w, h, tile_w, tile_h)
    for k, t in enumerate(owned_tiles(w, h, tile_w, tile_h, first, stride)):
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        x1, y1 = min(x0 + tile_w, w), min(y0 + tile_h, h)
        frame[y0:y1, x0:x1] = packed[k, : y1 - y0, : x1 - x0]
    return frame
