"""The C-ABI library loads and exports every symbol include/rtamd.h declares
(no compute calls: this runs without a GPU)."""
import ctypes as C
import os
import re

from conftest import ROOT
from raytracer_amd import _abi


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtamd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtamd_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_abi.PROTOTYPES)


def test_library_exports_every_declared_symbol():
    handle = C.CDLL(_abi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), name


def test_checking_build_of_the_library_exports_the_same():
    """librtamd_walkall.so (every shadow ray walked; the GPU tests compare its frames with the
    product's) is the same sources with one switch: same entry points."""
    handle = C.CDLL(os.path.join(os.path.dirname(_abi.LIB_PATH), "librtamd_walkall.so"))
    for name in declared_symbols():
        assert hasattr(handle, name), name


def test_struct_sizes_match_header_layout():
    # sizes the C compiler gives the structs of include/rtamd.h on x86-64
    assert C.sizeof(_abi.HitRecord) == 64
    assert C.sizeof(_abi.Light) == 96
    assert C.sizeof(_abi.Material) == 48
    assert C.sizeof(_abi.Surface) == 16 + 48 + 8 + 256 + 24
    assert C.sizeof(_abi.Mesh) == 48
    assert C.sizeof(_abi.Cubemap) == 96
    assert C.sizeof(_abi.RenderParams) == 56


def test_device_entry_points_fail_loudly_without_init():
    lib = _abi.lib()
    handle = C.c_void_p()
    blob = (C.c_ubyte * 8)()
    rc = lib.rtamd_scene_create(C.addressof(blob), 8, C.byref(handle))
    assert rc != 0
    assert lib.rtamd_last_error()
    assert lib.rtamd_version().startswith(b"rtamd")
