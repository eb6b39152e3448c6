"""The BVH build on the GPU where it is the default: meshes of at least 2^18
triangles (smaller meshes get the host's SAH tree unless RTAMD_BVH says
otherwise).  Two generated meshes:

  * a tessellated torus (263 168 triangles, well spread): the GPU-built tree is
    kept; frame and hit records must equal the host-built tree's bit for bit and
    the checker's on a window;
  * a height field of 2^18 triangles with thirty far outliers (one per power of
    two along each axis): all the field's centroids fall into ONE cell of the
    1024^3 Morton grid behind a chain of thirty single-outlier splits, so the
    radix tree is far deeper than the traversal stack and the scene must fall
    back to the host's builder by itself - same results.

This structure replaces the reference's kd-tree (mesh.c:350-451), whose shape is
not observable in its output; only nearest-hit results are compared."""
import math
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import ctypes as C

import raytracer_amd as ra
from raytracer_amd import _abi
import oracle_lib
from test_gpu_parity import assert_hits_equal, rgb8_within, render  # noqa: F401  (fixture)

SDL = """<?xml version='1.0' encoding='utf-8'?>
<Sdl>
<Config width="{w}" height="{h}" antialiasing="true" aa_samples="2" shadow_samples="2"
	reflection_samples="2" max_reflections="1" />
<Cameras><Camera position="{cam}" direction="{cdir}" up="0 1 0" fovy="50" name="cam" /></Cameras>
<Lights>
<PointLight position="3 9 4" intensity="0.7" color="1 1 1" name="point" />
<AreaLight position="-2 8 2" edge1="2 0 0" edge2="0 0 2" intensity="0.6" name="area" />
</Lights>
<Geometry>
<Plane edge1="0 0 40" edge2="40 0 0" name="floor" />
<Mesh src="{obj}" name="big" />
</Geometry>
<Materials>
<Material diffuse_color="0.8 0.8 0.8" name="grey" />
<Material diffuse_color="0.9 0.4 0.2" specular_exponent="50" reflect="0.3" name="shiny" />
</Materials>
<Scene camera="cam" lights="point,area" background="0.2 0.3 0.4">
<Translate vector="0 0 0">
<Translate vector="-20 -0.5 -20"><Shape geometry="floor" material="grey" /></Translate>
<Translate vector="0 1.2 0"><Rotate axis="1 0 0" angle="{tilt}"><Shape geometry="big" material="shiny" /></Rotate></Translate>
</Translate>
</Scene>
</Sdl>
"""


def write_obj(path, verts, normals, faces):
    """verts, normals: float arrays [n, 3] (same count, indexed together); faces: int [m, 3], 0-based."""
    with open(path, "w") as f:
        f.write("\n".join("v %.6f %.6f %.6f" % tuple(v) for v in verts))
        f.write("\n")
        f.write("\n".join("vn %.5f %.5f %.5f" % tuple(n) for n in normals))
        f.write("\n")
        f.write("\n".join("f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c) for a, b, c in (faces + 1)))
        f.write("\n")


def grid_faces(nu, nv, wrap_u, wrap_v):
    """Two triangles per cell of an nu x nv grid of vertices (row-major, u fastest)."""
    cu, cv = (nu if wrap_u else nu - 1), (nv if wrap_v else nv - 1)
    i, j = np.meshgrid(np.arange(cu), np.arange(cv), indexing="xy")
    i, j = i.ravel(), j.ravel()
    i1, j1 = (i + 1) % nu, (j + 1) % nv
    a, b, c, d = j * nu + i, j * nu + i1, j1 * nu + i1, j1 * nu + i
    return np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)])


def torus_obj(path):
    nu, nv = 514, 256                       # 2 * 514 * 256 = 263 168 triangles
    u, v = np.meshgrid(np.arange(nu) * 2 * math.pi / nu, np.arange(nv) * 2 * math.pi / nv, indexing="xy")
    u, v = u.ravel(), v.ravel()
    R, r = 1.6, 0.55
    pos = np.stack([(R + r * np.cos(v)) * np.cos(u), r * np.sin(v), (R + r * np.cos(v)) * np.sin(u)], 1)
    nrm = np.stack([np.cos(v) * np.cos(u), np.sin(v), np.cos(v) * np.sin(u)], 1)
    faces = grid_faces(nu, nv, True, True)
    assert len(faces) >= 1 << 18
    write_obj(path, pos, nrm, faces)
    return len(faces)


def outlier_field_obj(path):
    n = 363                                  # 2 * 362^2 = 262 088 ... one more row below
    nu, nv = n + 1, n + 1                    # 2 * 363^2 = 263 538 triangles
    x, z = np.meshgrid(np.linspace(-1.0, 1.0, nu), np.linspace(-1.0, 1.0, nv), indexing="xy")
    x, z = x.ravel(), z.ravel()
    y = 0.25 * np.sin(3.0 * x) * np.cos(2.5 * z)
    dydx, dydz = 0.75 * np.cos(3.0 * x) * np.cos(2.5 * z), -0.625 * np.sin(3.0 * x) * np.sin(2.5 * z)
    nrm = np.stack([-dydx, np.ones_like(x), -dydz], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pos = np.stack([x, y, z], 1)
    faces = grid_faces(nu, nv, False, False)
    assert len(faces) >= 1 << 18
    # thirty outliers: a small triangle at 2^k along each axis, k = 1 .. 10
    extra_pos, extra_nrm, extra_faces = [], [], []
    base = len(pos)
    for axis in range(3):
        for k in range(1, 11):
            c = np.zeros(3)
            c[axis] = float(2 ** k)
            for off in ((0.0, 0.0, 0.0), (0.3, 0.0, 0.0), (0.0, 0.0, 0.3)):
                extra_pos.append(c + np.array(off))
                extra_nrm.append((0.0, 1.0, 0.0))
            extra_faces.append((base, base + 1, base + 2))
            base += 3
    pos = np.concatenate([pos, np.array(extra_pos)])
    nrm = np.concatenate([nrm, np.array(extra_nrm)])
    faces = np.concatenate([faces, np.array(extra_faces)])
    write_obj(path, pos, nrm, faces)
    return len(faces)


def render_and_compare(render, monkeypatch, blob, expect_on_device, label):
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    monkeypatch.delenv("RTAMD_BVH", raising=False)
    t0 = time.perf_counter()
    dev = render.DeviceScene(blob)             # the default: no override
    t_default = time.perf_counter() - t0
    on_device = C.c_int(-1)
    _abi.check(_abi.lib().rtamd_scene_mesh_built_on_device(dev._handle, 0, C.byref(on_device)), "built_on_device")
    assert on_device.value == (1 if expect_on_device else 0)
    secs = C.c_double()
    _abi.check(_abi.lib().rtamd_scene_mesh_build_seconds(dev._handle, 0, C.byref(secs)), "build_seconds")
    fd, hd, sd = dev.render(p, want_hits=True)
    fd2, _, sd2 = dev.render(p)
    monkeypatch.setenv("RTAMD_BVH", "host")
    t0 = time.perf_counter()
    host = render.DeviceScene(blob)
    t_host = time.perf_counter() - t0
    monkeypatch.delenv("RTAMD_BVH")
    _abi.check(_abi.lib().rtamd_scene_mesh_built_on_device(host._handle, 0, C.byref(on_device)), "built_on_device")
    assert on_device.value == 0
    fh, hh, sh = host.render(p, want_hits=True)
    fh2, _, sh2 = host.render(p)
    print(f"\n[{label}] mesh build {secs.value * 1e3:.1f} ms (scene create {t_default:.2f} s default, "
          f"{t_host:.2f} s host tree); frame {sd2.ms_total:.2f} ms default tree, {sh2.ms_total:.2f} ms host tree")
    assert_hits_equal(hd, hh)
    assert np.array_equal(fd.cpu().numpy().view(np.uint32), fh.cpu().numpy().view(np.uint32))
    assert (sd.rays_primary, sd.rays_shadow, sd.rays_secondary) == (sh.rays_primary, sh.rays_shadow, sh.rays_secondary)
    assert (hd["triangle"] >= 0).mean() > 0.15
    # the checker on a window around the mesh
    w, h = blob.width, blob.height
    # ... where the mesh covers most of a 24 x 16 window
    mask = (hd["triangle"].reshape(h, w, blob.spp)[:, :, 0] >= 0).astype(np.int64)
    ii = np.zeros((h + 1, w + 1), np.int64)
    ii[1:, 1:] = mask.cumsum(0).cumsum(1)
    cover = ii[16:, 24:] - ii[:-16, 24:] - ii[16:, :-24] + ii[:-16, :-24]
    y0, x0 = np.unravel_index(int(cover.argmax()), cover.shape)
    win = (int(x0), int(y0), int(x0) + 24, int(y0) + 16)
    colour, ohits, _ = oracle_lib.OracleScene(blob).render(rng_mode=_abi.RNG_KEYED, window=win, want_hits=True, threads=8)
    _, whits, _ = dev.render(render.make_params(rng_mode=_abi.RNG_KEYED, window=win), want_hits=True)
    assert_hits_equal(whits, ohits)
    assert (ohits["triangle"] >= 0).mean() > 0.3
    got = fd.cpu().numpy()[win[1]:win[3], win[0]:win[2]]
    want = colour[win[1]:win[3], win[0]:win[2]]
    assert rgb8_within(got, want, 1) >= 0.999
    assert np.abs(got[..., :3] - want[..., :3]).max() < 2e-5
    dev.close()
    host.close()


def load_scene(tmp_path, name, obj, cam, cdir, tilt):
    path = os.path.join(str(tmp_path), name + ".sdl")
    with open(path, "w") as f:
        f.write(SDL.format(w=96, h=64, cam=cam, cdir=cdir, obj=obj, tilt=tilt))
    return ra.load_sdl(path)


def test_large_mesh_takes_the_device_built_tree_by_default(render, monkeypatch, tmp_path):
    obj = os.path.join(str(tmp_path), "torus-big.obj")
    ntri = torus_obj(obj)
    blob = load_scene(tmp_path, "torus-big", obj, "0 4 7", "0 -2.8 -7", 25)
    assert ntri >= 1 << 18
    render_and_compare(render, monkeypatch, blob, True, f"torus {ntri} triangles")


def test_tree_deeper_than_the_stack_falls_back_to_the_host_builder(render, monkeypatch, tmp_path):
    obj = os.path.join(str(tmp_path), "field-outliers.obj")
    ntri = outlier_field_obj(obj)
    blob = load_scene(tmp_path, "field-outliers", obj, "0 3 4", "0 -1.8 -4", 0)
    render_and_compare(render, monkeypatch, blob, False, f"height field with outliers, {ntri} triangles")


def test_depth_limit_knob_forces_the_fallback_on_a_small_mesh(render, monkeypatch):
    """RTAMD_BVH=device with RTAMD_BVH_MAX_DEPTH=4: every device-built tree is too deep, all
    meshes come back from the host builder, the frame is unchanged."""
    from conftest import small_blob
    blob = small_blob("showcase-uffizi")
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    monkeypatch.setenv("RTAMD_BVH", "device")
    monkeypatch.setenv("RTAMD_BVH_MAX_DEPTH", "4")
    fell_back = render.DeviceScene(blob)
    monkeypatch.delenv("RTAMD_BVH_MAX_DEPTH")
    kept = render.DeviceScene(blob)
    monkeypatch.delenv("RTAMD_BVH")
    flag = C.c_int(-1)
    _abi.check(_abi.lib().rtamd_scene_mesh_built_on_device(fell_back._handle, 0, C.byref(flag)), "flag")
    assert flag.value == 0
    _abi.check(_abi.lib().rtamd_scene_mesh_built_on_device(kept._handle, 0, C.byref(flag)), "flag")
    assert flag.value == 1
    fa, ha, _ = fell_back.render(p, want_hits=True)
    fb, hb, _ = kept.render(p, want_hits=True)
    assert_hits_equal(ha, hb)
    assert np.array_equal(fa.cpu().numpy().view(np.uint32), fb.cpu().numpy().view(np.uint32))
    fell_back.close()
    kept.close()
