"""Multi-GPU frame rendering: one process per GPU, screen-space tiles dealt out
cyclically, one gather of the packed tile buffers to rank 0 per frame.

The only data-path exchange is that gather (RCCL over xGMI when the process
group is "nccl"; the same code runs over "gloo" with CPU tensors in the CPU
tests).  The scene is tiny and fully replicated, and the keyed sample generator
depends only on (seed, pixel, sample, path), so the image does not depend on the
number of ranks.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _abi, tiles


def _pack(frame, params):
    """Owner's tiles of `frame` -> dense [max_owned, th, tw, 4] tensor."""
    h, w, _ = frame.shape
    tw, th = params.tile_w, params.tile_h
    nmax = tiles.max_owned_count(w, h, tw, th, params.tile_stride)
    if frame.is_cuda:
        packed = torch.zeros((nmax, th, tw, 4), dtype=torch.float32, device=frame.device)
        s = torch.cuda.current_stream(frame.device)
        _abi.check(_abi.lib().rtamd_tiles_pack(w, h, C.byref(params), frame.data_ptr(),
                                               packed.data_ptr(), s.cuda_stream), "rtamd_tiles_pack")
        return packed
    own = tiles.pack(frame.numpy(), tw, th, params.tile_first, params.tile_stride)
    packed = torch.zeros((nmax, th, tw, 4), dtype=torch.float32)
    packed[: own.shape[0]] = torch.from_numpy(own)
    return packed


def _unpack(packed, frame, params):
    h, w, _ = frame.shape
    if frame.is_cuda:
        s = torch.cuda.current_stream(frame.device)
        _abi.check(_abi.lib().rtamd_tiles_unpack(w, h, C.byref(params), packed.data_ptr(),
                                                 frame.data_ptr(), s.cuda_stream), "rtamd_tiles_unpack")
    else:
        tiles.unpack(packed.numpy(), frame.numpy(), params.tile_w, params.tile_h, params.tile_first,
                     params.tile_stride)


def owner_params(base, rank, world):
    """Copy of a parameter block with the tile assignment of `rank`."""
    p = _abi.RenderParams.from_buffer_copy(base)
    p.tile_first = rank
    p.tile_stride = world
    return p


def gather_frame(frame, base_params, rank, world, group=None):
    """Collects every rank's tiles into rank 0's `frame` (in place).  `frame` holds
    this rank's own tiles already.  Returns the frame (complete on rank 0 only)."""
    if world == 1:
        return frame
    mine = _pack(frame, owner_params(base_params, rank, world))
    # gloo moves host memory: device buffers are staged through the host there
    # (only used to rehearse the multi-rank path on one GPU; RCCL takes them as they are)
    staged = mine.is_cuda and dist.get_backend(group) == "gloo"
    if staged:
        mine = mine.cpu()
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, bufs, dst=0, group=group)
    if rank == 0:
        for r in range(1, world):
            buf = bufs[r].to(frame.device) if staged else bufs[r]
            _unpack(buf, frame, owner_params(base_params, r, world))
    return frame


def render_frame(scene, base_params, rank, world, frame=None, want_stats=True, group=None):
    """Renders this rank's tiles and gathers the frame on rank 0.
    Returns (frame, stats of this rank)."""
    p = owner_params(base_params, rank, world)
    frame, _, stats = scene.render(p, frame=frame, want_stats=want_stats)
    gather_frame(frame, base_params, rank, world, group=group)
    return frame, stats
