"""Host-side scene objects: the flat scene blob produced by the SDL front end.

`load_sdl` stands where the reference's `sdl_load` does (scene.c:674); what it
returns is the flattened form of the reference's Config + Scene globals
(scene.h:69-106).
"""
import ctypes as C
import os

import numpy as np

from . import _abi


class SceneBlob:
    """A validated scene blob plus typed views into it."""

    def __init__(self, data):
        self.data = bytearray(data)
        self._buf = (C.c_ubyte * len(self.data)).from_buffer(self.data)
        _abi.check(_abi.lib().rtamd_blob_validate(C.addressof(self._buf), len(self.data)),
                   "rtamd_blob_validate")
        self.header = _abi.BlobHeader.from_buffer(self.data, 0)

    # -- raw access ---------------------------------------------------------
    @property
    def address(self):
        return C.addressof(self._buf)

    @property
    def size(self):
        return len(self.data)

    def tobytes(self):
        return bytes(self.data)

    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.data)

    @classmethod
    def load(cls, path):
        with open(path, "rb") as f:
            return cls(f.read())

    def copy(self):
        return SceneBlob(bytes(self.data))

    # -- config -------------------------------------------------------------
    @property
    def width(self):
        return self.header.width

    @property
    def height(self):
        return self.header.height

    @property
    def spp(self):
        """Primary samples per pixel: aa_samples^2 with AA on, else 1 (raytracer.c:38-51)."""
        return self.header.aa_samples ** 2 if self.header.antialiasing else 1

    def with_config(self, **kw):
        """Copy of the blob with <Config> fields replaced (width, height,
        antialiasing, aa_samples, shadow_samples, reflection_samples,
        max_reflections)."""
        new = self.copy()
        for k, v in kw.items():
            if k not in ("width", "height", "antialiasing", "aa_samples", "shadow_samples",
                         "reflection_samples", "max_reflections"):
                raise KeyError(k)
            setattr(new.header, k, int(v))
        return SceneBlob(bytes(new.data))

    # -- tables -------------------------------------------------------------
    def _table(self, ctype, off, n):
        return [ctype.from_buffer(self.data, off + i * C.sizeof(ctype)) for i in range(n)]

    @property
    def lights(self):
        return self._table(_abi.Light, self.header.off_lights, self.header.num_lights)

    @property
    def materials(self):
        return self._table(_abi.Material, self.header.off_materials, self.header.num_materials)

    @property
    def surfaces(self):
        return self._table(_abi.Surface, self.header.off_surfaces, self.header.num_surfaces)

    @property
    def meshes(self):
        return self._table(_abi.Mesh, self.header.off_meshes, self.header.num_meshes)

    def mesh_arrays(self, m):
        """(vertex[nv,3] f64, normal[nn,3] f64, triangle[nt,6] i32) of mesh m."""
        d = self.meshes[m]
        v = np.frombuffer(self.data, np.float64, 3 * d.num_vertices, d.off_vertex).reshape(-1, 3)
        n = np.frombuffer(self.data, np.float64, 3 * d.num_normals, d.off_normal).reshape(-1, 3)
        t = np.frombuffer(self.data, np.int32, 6 * d.num_triangles, d.off_triangle).reshape(-1, 6)
        return v, n, t

    def cubemap_faces(self):
        if not self.header.has_cubemap:
            return None
        c = _abi.Cubemap.from_buffer(self.data, self.header.off_cubemap)
        return [np.frombuffer(self.data, np.float32, 4 * c.width[f] * c.height[f],
                              c.off_face[f]).reshape(c.height[f], c.width[f], 4)
                for f in range(6)]

    def fixed_rng_consumption(self):
        """True when every primary sample draws exactly two numbers and nothing
        else does: no area light in the scene's light list and no material that
        is both reflective and glossy (shading.c:18-36,88-105)."""
        if any(l.type == _abi.LIGHT_AREA for l in self.lights):
            return False
        used = {s.material for s in self.surfaces}
        mats = self.materials
        return not any(mats[i].reflect > 0 and mats[i].glossiness > 0 for i in used)


def load_sdl(path, asset_dir=None):
    """Parse an SDL scene file; asset paths resolve against `asset_dir`
    (default: the directory of the SDL file)."""
    if asset_dir is None:
        asset_dir = os.path.dirname(os.path.abspath(path))
    blob = C.c_void_p()
    size = C.c_size_t()
    lib = _abi.lib()
    _abi.check(lib.rtamd_sdl_load(os.fsencode(path), os.fsencode(asset_dir),
                                  C.byref(blob), C.byref(size)), f"rtamd_sdl_load({path})")
    try:
        data = C.string_at(blob.value, size.value)
    finally:
        lib.rtamd_blob_free(blob)
    return SceneBlob(data)


def glibc_rand(seed, first, n):
    """The first..first+n-1 values of rand() after srand(seed) (raytracer.c:74)."""
    out = np.empty(n, np.int32)
    _abi.check(_abi.lib().rtamd_glibc_rand_fill(seed, first, n, out.ctypes.data),
               "rtamd_glibc_rand_fill")
    return out


def quantise_rgb8(colour):
    """float RGBA [h,w,4] bottom-up -> uint8 RGB [h,w,3] top-down (ppm.c:16-24)."""
    colour = np.ascontiguousarray(colour, np.float32)
    h, w, _ = colour.shape
    out = np.empty((h, w, 3), np.uint8)
    _abi.check(_abi.lib().rtamd_quantise_rgb8(colour.ctypes.data, w, h, out.ctypes.data),
               "rtamd_quantise_rgb8")
    return out


def ppm_write(colour, path):
    colour = np.ascontiguousarray(colour, np.float32)
    h, w, _ = colour.shape
    _abi.check(_abi.lib().rtamd_ppm_write(colour.ctypes.data, w, h, os.fsencode(path)),
               "rtamd_ppm_write")


def png_write(colour, path):
    """rtamd_png_write: float [h, w, 4] bottom-up frame -> 8-bit RGB PNG."""
    colour = np.ascontiguousarray(colour, np.float32)
    h, w, _ = colour.shape
    _abi.check(_abi.lib().rtamd_png_write(colour.ctypes.data, w, h, os.fsencode(path)),
               "rtamd_png_write")
