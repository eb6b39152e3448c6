"""ctypes binding of oracle/liboracle.so - the CPU restatement used as the checker.

Test infrastructure only: nothing under raytracer_amd/ imports this.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")

HIT_DTYPE = np.dtype([("surface", "<i4"), ("triangle", "<i4"), ("t", "<f4"), ("pad", "<f4"),
                      ("position", "<f8", 3), ("normal", "<f8", 3)])


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays_primary", "rays_shadow", "rays_secondary", "env_lookups", "light_samples",
                 "draws", "bbox_tests", "surface_probes", "mesh_probes", "mesh_probe_levels")]

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_secondary


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle liboracle.so`")
        l = C.CDLL(LIB_PATH)
        l.oracle_scene_create.restype = C.c_void_p
        l.oracle_scene_create.argtypes = [C.c_void_p, C.c_size_t]
        l.oracle_scene_destroy.argtypes = [C.c_void_p]
        l.oracle_render.restype = C.c_int
        l.oracle_render.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_int]
        l.oracle_trace.restype = C.c_int
        l.oracle_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.oracle_camera_rays.restype = C.c_int
        l.oracle_camera_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                         C.c_size_t, C.c_void_p]
        l.oracle_mesh_bruteforce.restype = C.c_int
        l.oracle_mesh_bruteforce.argtypes = [C.c_void_p, C.c_int, C.c_void_p,
                                             C.POINTER(C.c_double)]
        for name in ("oracle_keyed_key0", "oracle_keyed_draw"):
            getattr(l, name).restype = C.c_uint32
            getattr(l, name).argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        l.oracle_keyed_child.restype = C.c_uint32
        l.oracle_keyed_child.argtypes = [C.c_uint32, C.c_uint32]
        _lib = l
    return _lib


class OracleScene:
    def __init__(self, blob):
        """blob: raytracer_amd.SceneBlob"""
        self.blob = blob
        self.handle = lib().oracle_scene_create(blob.address, blob.size)
        if not self.handle:
            raise RuntimeError("oracle_scene_create failed")
        self.width, self.height, self.spp = blob.width, blob.height, blob.spp

    def __del__(self):
        if getattr(self, "handle", None):
            lib().oracle_scene_destroy(self.handle)
            self.handle = None

    def render(self, seed=0x20071208, rng_mode=0, window=None, want_hits=False, threads=1):
        """Returns (colour [h,w,4] float32 bottom-up, hits or None, Stats)."""
        colour = np.zeros((self.height, self.width, 4), np.float32)
        x0, y0, x1, y1 = window if window else (0, 0, 0, 0)
        hits = None
        if want_hits:
            npix = (x1 - x0) * (y1 - y0) if window else self.width * self.height
            hits = np.zeros(npix * self.spp, HIT_DTYPE)
        st = Stats()
        rc = lib().oracle_render(self.handle, seed, rng_mode, x0, y0, x1, y1, colour.ctypes.data,
                                 hits.ctypes.data if hits is not None else None, C.byref(st),
                                 threads)
        if rc:
            raise RuntimeError("oracle_render failed")
        return colour, hits, st

    def trace(self, rays):
        rays = np.ascontiguousarray(rays, np.float64)
        hits = np.zeros(rays.shape[0], HIT_DTYPE)
        if lib().oracle_trace(self.handle, rays.ctypes.data, rays.shape[0], hits.ctypes.data):
            raise RuntimeError("oracle_trace failed")
        return hits

    def camera_rays(self, pixels, offsets, near_plane):
        pixels = np.ascontiguousarray(pixels, np.int32)
        offsets = np.ascontiguousarray(offsets, np.float32)
        rays = np.zeros((pixels.shape[0], 8), np.float64)
        if lib().oracle_camera_rays(self.handle, pixels.ctypes.data, offsets.ctypes.data,
                                    near_plane, pixels.shape[0], rays.ctypes.data):
            raise RuntimeError("oracle_camera_rays failed")
        return rays

    def mesh_bruteforce(self, mesh, ray):
        ray = np.ascontiguousarray(ray, np.float64)
        t = C.c_double()
        tri = lib().oracle_mesh_bruteforce(self.handle, mesh, ray.ctypes.data, C.byref(t))
        return tri, t.value
