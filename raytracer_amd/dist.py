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


class FrameGather:
    """The per-frame exchange with its buffers allocated ONCE: this rank's packed
    tiles, and on rank 0 one receive buffer per rank.  Every rank pads to the
    tile count of owner 0 (the largest), so all buffers have one shape."""

    def __init__(self, frame, base_params, rank, world, group=None):
        h, w, _ = frame.shape
        self.rank, self.world, self.group = rank, world, group
        self.base = base_params
        self.mine_params = owner_params(base_params, rank, world)
        tw, th = self.mine_params.tile_w or 32, self.mine_params.tile_h or 32
        self.tile = (tw, th)
        nmax = tiles.max_owned_count(w, h, tw, th, world)
        self.cuda = frame.is_cuda
        # gloo moves host memory: device buffers are staged through the host there (only
        # used to rehearse the multi-rank path on one GPU; RCCL takes them as they are)
        self.staged = self.cuda and world > 1 and dist.get_backend(group) == "gloo"
        shape = (nmax, th, tw, 4)
        self.packed = torch.zeros(shape, dtype=torch.float32, device=frame.device)
        self.send = torch.zeros(shape, dtype=torch.float32, pin_memory=True) if self.staged else self.packed
        self.recv = None
        self.recv_all = None
        self.recv_dev = None
        if rank == 0 and world > 1:
            # ONE buffer for all ranks (the gather's outputs are views of it), so that the
            # tiles of every other rank are unpacked by a single launch
            where = "cpu" if self.staged else frame.device
            self.recv_all = torch.zeros((world,) + shape, dtype=torch.float32, device=where,
                                        pin_memory=bool(self.staged))
            self.recv = [self.recv_all[r] for r in range(world)]
            if self.staged:
                self.recv_dev = torch.zeros((world,) + shape, dtype=torch.float32, device=frame.device)
        self.owner = [owner_params(base_params, r, world) for r in range(world)]
        self.all_params = owner_params(base_params, 0, world)

    def __call__(self, frame):
        """Collects every rank's tiles into rank 0's `frame` (in place); `frame` holds
        this rank's own tiles already."""
        if self.world == 1:
            return frame
        h, w, _ = frame.shape
        tw, th = self.tile
        if self.cuda:
            s = torch.cuda.current_stream(frame.device)
            _abi.check(_abi.lib().rtamd_tiles_pack(w, h, C.byref(self.mine_params), frame.data_ptr(),
                                                   self.packed.data_ptr(), s.cuda_stream), "rtamd_tiles_pack")
            if self.staged:
                self.send.copy_(self.packed)
        else:
            own = tiles.pack(frame.numpy(), tw, th, self.rank, self.world)
            self.packed[: own.shape[0]] = torch.from_numpy(own)
        dist.gather(self.send, self.recv, dst=0, group=self.group)
        if self.rank == 0:
            if self.cuda:
                buf = self.recv_all
                if self.staged:
                    self.recv_dev.copy_(self.recv_all, non_blocking=True)
                    buf = self.recv_dev
                s = torch.cuda.current_stream(frame.device)
                stride = buf[0].numel() * 4
                _abi.check(_abi.lib().rtamd_tiles_unpack_all(w, h, C.byref(self.all_params), 1, buf.data_ptr(),
                                                             stride, frame.data_ptr(), s.cuda_stream),
                           "rtamd_tiles_unpack_all")
            else:
                for r in range(1, self.world):
                    _unpack(self.recv[r], frame, self.owner[r])
        return frame


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


def gather_frame(frame, base_params, rank, world, group=None, gather=None):
    """Collects every rank's tiles into rank 0's `frame` (in place).  `frame` holds
    this rank's own tiles already.  Returns the frame (complete on rank 0 only).
    Pass a FrameGather to reuse its buffers from frame to frame."""
    if world == 1:
        return frame
    if gather is None:
        gather = FrameGather(frame, base_params, rank, world, group=group)
    return gather(frame)


def render_frame(scene, base_params, rank, world, frame=None, want_stats=True, group=None, gather=None):
    """Renders this rank's tiles and gathers the frame on rank 0.
    Returns (frame, stats of this rank)."""
    p = owner_params(base_params, rank, world)
    frame, _, stats = scene.render(p, frame=frame, want_stats=want_stats)
    gather_frame(frame, base_params, rank, world, group=group, gather=gather)
    return frame, stats
