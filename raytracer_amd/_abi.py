"""ctypes view of include/rtamd.h: the structures and the loaded library.

The shared library is built in-tree by `make -C raytracer_amd/csrc` (see
__graft_entry__.build).  There is no Python or CPU fallback for the device
entry points: when the library is missing, importing this module's `lib()`
raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTAMD_LIB selects another build of the same library (A/B measurements).
LIB_PATH = os.environ.get("RTAMD_LIB") or os.path.join(_HERE, "csrc", "librtamd.so")

RTAMD_BLOB_MAGIC = 0x324353444D415452
RTAMD_MAX_LIGHTS = 8

SHAPE_PLANE, SHAPE_DISK, SHAPE_SPHERE, SHAPE_CYLINDER, SHAPE_CONE, SHAPE_MESH = range(6)
LIGHT_POINT, LIGHT_AREA = 0, 1
RNG_GLIBC, RNG_KEYED = 0, 1
RENDER_TIME_KERNELS, RENDER_PROGRESSIVE, RENDER_RESTART = 1, 2, 4     # rtamd_render_params.flags

STATUS_NAMES = {
    0: "RTAMD_OK", -1: "RTAMD_ERR_ARG", -2: "RTAMD_ERR_IO", -3: "RTAMD_ERR_SCENE",
    -4: "RTAMD_ERR_HIP", -5: "RTAMD_ERR_NOMEM", -6: "RTAMD_ERR_LIMIT", -7: "RTAMD_ERR_QUEUE",
}


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("intensity", C.c_float),
                ("position", C.c_double * 3), ("edge1", C.c_double * 3),
                ("edge2", C.c_double * 3), ("colour", C.c_float * 4)]


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_float * 4), ("specular", C.c_float * 4),
                ("shininess", C.c_int32), ("reflect", C.c_float),
                ("glossiness", C.c_float), ("pad", C.c_int32)]


class Surface(C.Structure):
    _fields_ = [("shape_type", C.c_int32), ("material", C.c_int32),
                ("mesh", C.c_int32), ("capped", C.c_int32),
                ("edge1", C.c_double * 3), ("edge2", C.c_double * 3),
                ("radius", C.c_float), ("height", C.c_float),
                ("model_to_world", C.c_double * 16), ("world_to_model", C.c_double * 16),
                ("bbox", C.c_float * 6)]


class Mesh(C.Structure):
    _fields_ = [("num_vertices", C.c_int32), ("num_normals", C.c_int32),
                ("num_triangles", C.c_int32), ("pad", C.c_int32),
                ("off_vertex", C.c_uint64), ("off_normal", C.c_uint64),
                ("off_triangle", C.c_uint64), ("off_name", C.c_uint64)]


class Cubemap(C.Structure):
    _fields_ = [("width", C.c_int32 * 6), ("height", C.c_int32 * 6),
                ("off_face", C.c_uint64 * 6)]


class BlobHeader(C.Structure):
    _fields_ = [("magic", C.c_uint64), ("total_size", C.c_uint64),
                ("width", C.c_int32), ("height", C.c_int32),
                ("antialiasing", C.c_int32), ("aa_samples", C.c_int32),
                ("shadow_samples", C.c_int32), ("reflection_samples", C.c_int32),
                ("max_reflections", C.c_int32), ("depth_of_field", C.c_int32),
                ("cam_position", C.c_double * 3), ("cam_u", C.c_double * 3),
                ("cam_v", C.c_double * 3), ("cam_w", C.c_double * 3),
                ("cam_fov", C.c_float), ("cam_near_plane", C.c_float),
                ("background", C.c_float * 4),
                ("num_lights", C.c_int32), ("num_materials", C.c_int32),
                ("num_surfaces", C.c_int32), ("num_meshes", C.c_int32),
                ("has_cubemap", C.c_int32), ("pad", C.c_int32),
                ("off_lights", C.c_uint64), ("off_materials", C.c_uint64),
                ("off_surfaces", C.c_uint64), ("off_meshes", C.c_uint64),
                ("off_cubemap", C.c_uint64)]


class HitRecord(C.Structure):
    _fields_ = [("surface", C.c_int32), ("triangle", C.c_int32),
                ("t", C.c_float), ("pad", C.c_float),
                ("position", C.c_double * 3), ("normal", C.c_double * 3)]


class RenderParams(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("rng_mode", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32),
                ("tile_first", C.c_int32), ("tile_stride", C.c_int32),
                ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("flags", C.c_int32), ("sample_first", C.c_int32), ("sample_stride", C.c_int32),
                ("sample_count", C.c_int32)]


class RenderStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("rays_secondary", C.c_uint64), ("env_lookups", C.c_uint64),
                ("queue_peak", C.c_uint64), ("samples", C.c_uint64),
                ("ms_total", C.c_float), ("ms_primary", C.c_float),
                ("launches_primary", C.c_int32), ("launches_bounce", C.c_int32),
                ("bvh_nodes", C.c_uint64), ("tri_tests", C.c_uint64)]


# Every symbol include/rtamd.h declares, with its prototype.
PROTOTYPES = {
    "rtamd_sdl_load": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_size_t)]),
    "rtamd_blob_free": (None, [C.c_void_p]),
    "rtamd_blob_validate": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rtamd_ppm_write": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "rtamd_png_write": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "rtamd_quantise_rgb8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rtamd_glibc_rand_fill": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]),
    "rtamd_init": (C.c_int, [C.c_int]),
    "rtamd_shutdown": (C.c_int, []),
    "rtamd_scene_create": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rtamd_scene_destroy": (C.c_int, [C.c_void_p]),
    "rtamd_render": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                               C.POINTER(RenderStats), C.c_void_p]),
    "rtamd_render_noalloc": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p]),
    "rtamd_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rtamd_camera_rays": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p,
                                    C.c_void_p]),
    "rtamd_tiles_owned": (C.c_int, [C.c_int, C.c_int, C.POINTER(RenderParams)]),
    "rtamd_tiles_pack": (C.c_int, [C.c_int, C.c_int, C.POINTER(RenderParams), C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "rtamd_tiles_unpack": (C.c_int, [C.c_int, C.c_int, C.POINTER(RenderParams), C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "rtamd_tiles_unpack_all": (C.c_int, [C.c_int, C.c_int, C.POINTER(RenderParams), C.c_int, C.c_void_p,
                                         C.c_uint64, C.c_void_p, C.c_void_p]),
    "rtamd_scene_take_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "rtamd_last_cubemap_load_seconds": (C.c_double, []),
    "rtamd_scene_mesh_build_seconds": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "rtamd_scene_mesh_built_on_device": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "rtamd_scene_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]),
    "rtamd_last_error": (C.c_char_p, []),
    "rtamd_version": (C.c_char_p, []),
}

_lib = None


def lib():
    """The loaded librtamd.so with prototypes attached (loaded once)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is not built; run `make -C raytracer_amd/csrc` "
                "(or __graft_entry__.build())")
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


class RtamdError(RuntimeError):
    def __init__(self, status, where):
        msg = lib().rtamd_last_error()
        text = msg.decode("utf-8", "replace") if msg else ""
        super().__init__(f"{where}: {STATUS_NAMES.get(status, status)}: {text}")
        self.status = status


def check(status, where):
    if status != 0:
        raise RtamdError(status, where)
