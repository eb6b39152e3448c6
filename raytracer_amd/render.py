"""Device-side plumbing above the C ABI: torch tensors hold device memory and
streams; every computation is done by librtamd.so (HIP kernels).  There is no
CPU fallback: without a GPU `DeviceScene` raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from .scene import SceneBlob

DEFAULT_SEED = 0x20071208   # raytracer.c:74

_initialised_device = None


def init(device=0):
    """rtamd_init on `device` (index of the visible HIP devices)."""
    global _initialised_device
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the render path has no CPU fallback")
    if _initialised_device != device:
        torch.cuda.set_device(device)
        _abi.check(_abi.lib().rtamd_init(device), "rtamd_init")
        _initialised_device = device
    return device


def make_params(seed=DEFAULT_SEED, rng_mode=_abi.RNG_KEYED, tile=(32, 32), tile_first=0,
                tile_stride=1, window=None, flags=0, sample_first=0, sample_stride=0, sample_count=0):
    p = _abi.RenderParams()
    p.seed = seed
    p.rng_mode = rng_mode
    p.tile_w, p.tile_h = tile
    p.tile_first = tile_first
    p.tile_stride = tile_stride
    if window is not None:
        p.x0, p.y0, p.x1, p.y1 = window
    p.flags = flags
    p.sample_first, p.sample_stride, p.sample_count = sample_first, sample_stride, sample_count
    return p


class DeviceScene:
    """A scene flattened into HBM (rtamd_scene_create)."""

    def __init__(self, blob: SceneBlob, device=0):
        self.device_index = init(device)
        self.device = torch.device("cuda", self.device_index)
        self.blob = blob
        handle = C.c_void_p()
        _abi.check(_abi.lib().rtamd_scene_create(blob.address, blob.size, C.byref(handle)),
                   "rtamd_scene_create")
        self._handle = handle
        self.width, self.height, self.spp = blob.width, blob.height, blob.spp

    def close(self):
        if self._handle:
            _abi.lib().rtamd_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def default_rng_mode(self):
        return _abi.RNG_GLIBC if self.blob.fixed_rng_consumption() else _abi.RNG_KEYED

    def new_frame(self):
        return torch.zeros((self.height, self.width, 4), dtype=torch.float32, device=self.device)

    def render(self, params=None, frame=None, want_hits=False, want_stats=True, stream=None):
        """Runs rtamd_render.  Returns (frame, hits, stats): frame is a float32
        [height, width, 4] CUDA tensor in the reference's bottom-up layout, hits a
        structured numpy array (or None), stats an _abi.RenderStats (or None)."""
        if params is None:
            params = make_params(rng_mode=self.default_rng_mode())
        if frame is None:
            frame = self.new_frame()
        hits_t = None
        if want_hits:
            if params.x1 or params.y1:
                npix = (params.x1 - params.x0) * (params.y1 - params.y0)
            else:
                npix = self.width * self.height
            hits_t = torch.zeros((npix * self.spp, C.sizeof(_abi.HitRecord)), dtype=torch.uint8,
                                 device=self.device)
        stats = _abi.RenderStats() if want_stats else None
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        _abi.check(_abi.lib().rtamd_render(
            self._handle, C.byref(params), frame.data_ptr(),
            hits_t.data_ptr() if hits_t is not None else None,
            C.byref(stats) if stats is not None else None, s.cuda_stream), "rtamd_render")
        hits = None
        if hits_t is not None:
            hits = hits_from_bytes(hits_t.cpu().numpy())
        return frame, hits, stats

    def take_kernel_times(self):
        """(ms, launches) of the primary-kernel launches recorded by renders with
        flags=_abi.RENDER_TIME_KERNELS since the last call; waits for the last one."""
        ms, n = C.c_float(), C.c_int()
        _abi.check(_abi.lib().rtamd_scene_take_kernel_times(self._handle, C.byref(ms), C.byref(n)),
                   "rtamd_scene_take_kernel_times")
        return ms.value, n.value

    def trace(self, rays):
        """rtamd_trace: rays float64 [n, 8] (origin, direction, near, far)."""
        rays_t = torch.as_tensor(np.ascontiguousarray(rays, np.float64), device=self.device)
        n = rays_t.shape[0]
        out = torch.zeros((n, C.sizeof(_abi.HitRecord)), dtype=torch.uint8, device=self.device)
        s = torch.cuda.current_stream(self.device)
        _abi.check(_abi.lib().rtamd_trace(self._handle, rays_t.data_ptr(), n, out.data_ptr(),
                                          s.cuda_stream), "rtamd_trace")
        torch.cuda.synchronize(self.device)
        return hits_from_bytes(out.cpu().numpy())

    def camera_rays(self, params=None):
        """rtamd_camera_rays: float64 [height, width, spp, 8]."""
        if params is None:
            params = make_params(rng_mode=self.default_rng_mode())
        out = torch.zeros((self.height, self.width, self.spp, 8), dtype=torch.float64,
                          device=self.device)
        s = torch.cuda.current_stream(self.device)
        _abi.check(_abi.lib().rtamd_camera_rays(self._handle, C.byref(params), out.data_ptr(),
                                                s.cuda_stream), "rtamd_camera_rays")
        torch.cuda.synchronize(self.device)
        return out.cpu().numpy()


HIT_DTYPE = np.dtype([("surface", "<i4"), ("triangle", "<i4"), ("t", "<f4"), ("pad", "<f4"),
                      ("position", "<f8", 3), ("normal", "<f8", 3)])


def hits_from_bytes(raw):
    """uint8 [n, 64] or flat bytes -> structured array of rtamd_hit_record."""
    return np.frombuffer(np.ascontiguousarray(raw).tobytes(), dtype=HIT_DTYPE)


def total_rays(stats):
    return stats.rays_primary + stats.rays_shadow + stats.rays_secondary
