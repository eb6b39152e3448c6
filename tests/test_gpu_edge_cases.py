"""Edge cases of the render path against the CPU checker (same keyed samples):
empty scenes, one-triangle meshes, frames smaller than a tile or a wave, sample
counts that do not divide a wave, more meshes than the light shafts list, more
surfaces than their skip mask covers.  The reference has no tests of its own;
these are the corners of its scene format and of this implementation's
structures."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import raytracer_amd as ra
from raytracer_amd import _abi
import oracle_lib
from conftest import ASSETS
from test_gpu_parity import assert_hits_equal, rgb8_within, render  # noqa: F401  (fixture)

HEAD = """<?xml version='1.0' encoding='utf-8'?>
<Sdl>
<Config width="{w}" height="{h}" antialiasing="{aa_on}" aa_samples="{aa}" shadow_samples="{ss}"
	reflection_samples="2" max_reflections="2" />
<Cameras><Camera position="0 4 9" direction="0 -3 -8" up="0 1 0" fovy="60" name="cam" /></Cameras>
<Lights>
<PointLight position="3 9 4" intensity="0.7" color="1 1 1" name="point" />
<AreaLight position="-2 8 2" edge1="2 0 0" edge2="0 0 2" intensity="0.6" name="area" />
</Lights>
<Geometry>
<Plane edge1="0 0 20" edge2="20 0 0" name="floor" />
<Sphere radius="0.3" name="bead" />
<Mesh src="{assets}/triangle.obj" name="tri" />
<Mesh src="{assets}/cube.obj" name="cube" />
<Mesh src="{assets}/torus.obj" name="torus" />
</Geometry>
<Materials>
<Material diffuse_color="0.8 0.8 0.8" name="grey" />
<Material diffuse_color="0.9 0.4 0.2" specular_exponent="50" reflect="0.3" name="shiny" />
</Materials>
<Scene camera="cam" lights="point,area" background="0.2 0.3 0.4"{cubemap}>
<Translate vector="0 0 0">
"""
TAIL = "</Translate>\n</Scene>\n</Sdl>\n"
FLOOR = '<Translate vector="-10 0 -10"><Shape geometry="floor" material="grey" /></Translate>\n'


def write_scene(tmp_path, name, body, w=48, h=32, aa=2, ss=3, cubemap=False):
    cm = f' cubemap="{ASSETS}/cubemap/uffizi_gallery"' if cubemap else ""
    text = HEAD.format(w=w, h=h, aa_on="true" if aa else "false", aa=max(aa, 1), ss=ss, assets=ASSETS,
                       cubemap=cm) + body + TAIL
    path = os.path.join(str(tmp_path), name + ".sdl")
    with open(path, "w") as f:
        f.write(text)
    return ra.load_sdl(path)


def check(render, blob, tile=(32, 32)):
    dev = render.DeviceScene(blob)
    p = render.make_params(rng_mode=_abi.RNG_KEYED, tile=tile)
    frame, hits, st = dev.render(p, want_hits=True)
    colour, ohits, ost = oracle_lib.OracleScene(blob).render(rng_mode=_abi.RNG_KEYED, want_hits=True, threads=8)
    assert_hits_equal(hits, ohits)
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary, st.env_lookups) == (
        ost.rays_primary, ost.rays_shadow, ost.rays_secondary, ost.env_lookups)
    g = frame.cpu().numpy()
    assert rgb8_within(g, colour, 1) >= 0.999
    assert np.abs(g[..., :3] - colour[..., :3]).max() < 2e-5
    dev.close()
    return st


def test_scene_without_geometry_is_background_or_environment(render, tmp_path):
    st = check(render, write_scene(tmp_path, "empty", ""))
    assert st.rays_shadow == 0 and st.env_lookups == 0
    st = check(render, write_scene(tmp_path, "empty-env", "", cubemap=True))
    assert st.env_lookups == st.rays_primary


def test_one_triangle_mesh_and_single_leaf_bvh(render, tmp_path):
    body = FLOOR + '<Translate vector="-0.5 1 0"><Scale scale="3 3 3"><Shape geometry="tri" material="shiny" /></Scale></Translate>\n'
    check(render, write_scene(tmp_path, "one-triangle", body))


@pytest.mark.parametrize("w,h,tile", [(1, 1, (32, 32)), (3, 5, (32, 32)), (37, 23, (16, 8)), (64, 1, (8, 8))])
def test_frames_smaller_than_a_tile_and_ragged_tiles(render, tmp_path, w, h, tile):
    body = FLOOR + '<Translate vector="0 1 0"><Shape geometry="cube" material="shiny" /></Translate>\n'
    check(render, write_scene(tmp_path, f"ragged-{w}x{h}", body, w=w, h=h), tile=tile)


@pytest.mark.parametrize("aa,ss", [(0, 1), (1, 1), (3, 2), (5, 1), (7, 2), (8, 4)])
def test_sample_counts_that_do_not_fill_or_divide_a_wave(render, tmp_path, aa, ss):
    body = FLOOR + ('<Translate vector="0 1.2 0"><Rotate axis="1 0 0" angle="30"><Shape geometry="torus" material="shiny" />'
                    '</Rotate></Translate>\n')
    check(render, write_scene(tmp_path, f"spp-{aa}-{ss}", body, w=24, h=16, aa=aa, ss=ss))


@pytest.mark.parametrize("ss", [3, 5, 6, 8, 9])
def test_bounced_hits_with_every_light_sample_layout(render, tmp_path, ss):
    """Reflective surfaces under area lights: the bounce kernel lays the n x n light
    samples of its hit points out over the lanes of a wave in runs (n x n < 64),
    in chunks of 64 (n x n >= 64) or keeps them per lane (short runs)."""
    body = FLOOR.replace("grey", "shiny") + (
        '<Translate vector="0 1.2 0"><Rotate axis="1 0 0" angle="30"><Shape geometry="torus" material="shiny" />'
        '</Rotate></Translate>\n<Translate vector="2 0.5 1"><Shape geometry="cube" material="grey" /></Translate>\n')
    st = check(render, write_scene(tmp_path, f"bounce-{ss}", body, w=20, h=12, aa=2, ss=ss))
    assert st.rays_secondary > 0


def test_more_meshes_than_shaft_lists(render, tmp_path):
    body = FLOOR
    for i in range(7):
        body += (f'<Translate vector="{-4.5 + 1.5 * i} {0.6 + 0.3 * (i % 3)} {-1 + 0.7 * (i % 2)}"><Rotate axis="0 1 0" angle="{17 * i}">'
                 f'<Scale scale="0.5 0.5 0.5"><Shape geometry="{"cube" if i % 2 else "torus"}" material="grey" /></Scale></Rotate></Translate>\n')
    check(render, write_scene(tmp_path, "many-meshes", body, w=64, h=40, aa=2, ss=3))


def test_more_surfaces_than_the_skip_mask(render, tmp_path):
    body = FLOOR
    for i in range(70):
        body += (f'<Translate vector="{-5 + (i % 10)} {0.3 + 0.05 * (i % 7)} {-4 + (i // 10)}">'
                 f'<Shape geometry="bead" material="{"shiny" if i % 5 == 0 else "grey"}" /></Translate>\n')
    body += '<Translate vector="0 1.5 1"><Shape geometry="cube" material="grey" /></Translate>\n'
    check(render, write_scene(tmp_path, "many-surfaces", body, w=64, h=40, aa=2, ss=2))


def test_limits_are_reported_not_clamped(render, tmp_path):
    blob = write_scene(tmp_path, "limits", FLOOR)
    for cfg in (dict(aa_samples=65), dict(shadow_samples=65), dict(reflection_samples=257), dict(max_reflections=61)):
        with pytest.raises(RuntimeError):
            render.DeviceScene(blob.with_config(**cfg)).close()


def test_queue_overflow_is_reported_and_harmless(render, tmp_path, monkeypatch):
    """A ray queue too small for the frame: the call with stats reports it, the
    asynchronous call reports it through the next call on the scene, nothing is
    written out of bounds (the frame is merely incomplete), and the scene renders
    correctly again once the queue is large enough."""
    import torch
    body = FLOOR.replace("grey", "shiny") + (
        '<Translate vector="0 1.2 0"><Shape geometry="torus" material="shiny" /></Translate>\n')
    blob = write_scene(tmp_path, "overflow", body, w=64, h=48, aa=2, ss=2)
    monkeypatch.setenv("RTAMD_QUEUE_OVERFLOW_AT", "300")
    dev = render.DeviceScene(blob)
    monkeypatch.delenv("RTAMD_QUEUE_OVERFLOW_AT")
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    with pytest.raises(_abi.RtamdError) as e:
        dev.render(p)
    assert e.value.status == -7
    frame, _, _ = dev.render(p, want_stats=False)     # asynchronous: returns 0
    torch.cuda.synchronize()
    with pytest.raises(_abi.RtamdError) as e:
        dev.render(p, want_stats=False)
    assert e.value.status == -7 and "earlier render" in str(e.value)
    dev.close()
    check(render, blob)


@pytest.mark.parametrize("name", ["showcase-uffizi", "3-aa", "8-glossy", "prims"])
def test_device_built_bvh_gives_the_same_frame_as_the_host_built_one(render, monkeypatch, name):
    """The BVH built by the GPU kernels (linear BVH) against the host's SAH tree: hit
    records (surface, triangle, t, position, normal), ray counts and the frame are
    bitwise equal - the tree's shape is not observable."""
    from conftest import small_blob
    blob = small_blob(name)
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    monkeypatch.setenv("RTAMD_BVH", "host")
    host = render.DeviceScene(blob)
    monkeypatch.setenv("RTAMD_BVH", "device")
    dev = render.DeviceScene(blob)
    monkeypatch.delenv("RTAMD_BVH")
    fh, hh, sh = host.render(p, want_hits=True)
    fd, hd, sd = dev.render(p, want_hits=True)
    assert_hits_equal(hd, hh)
    assert np.array_equal(fd.cpu().numpy().view(np.uint32), fh.cpu().numpy().view(np.uint32))
    assert (sd.rays_primary, sd.rays_shadow, sd.rays_secondary) == (sh.rays_primary, sh.rays_shadow, sh.rays_secondary)
    host.close()
    dev.close()


def test_device_built_bvh_on_degenerate_meshes(render, tmp_path, monkeypatch):
    """One triangle, two triangles, many coincident centroids (equal Morton codes)."""
    monkeypatch.setenv("RTAMD_BVH", "device")
    body = FLOOR + '<Translate vector="-0.5 1 0"><Scale scale="3 3 3"><Shape geometry="tri" material="shiny" /></Scale></Translate>\n'
    check(render, write_scene(tmp_path, "dev-one-triangle", body))
    # a fan of 40 triangles that share a centroid up to rounding, plus a far outlier
    lines = ["vn 0 0 1"]
    import math
    for k in range(40):
        a = 2 * math.pi * k / 40
        lines += [f"v {math.cos(a):.6f} {math.sin(a):.6f} {0.01 * k:.6f}",
                  f"v {math.cos(a + 2.0944):.6f} {math.sin(a + 2.0944):.6f} {0.01 * k:.6f}",
                  f"v {math.cos(a + 4.1888):.6f} {math.sin(a + 4.1888):.6f} {0.01 * k:.6f}"]
    lines += ["v 30 30 0", "v 31 30 0", "v 30 31 0"]
    for k in range(41):
        lines.append(f"f {3 * k + 1}//1 {3 * k + 2}//1 {3 * k + 3}//1")
    obj = tmp_path / "fan.obj"
    obj.write_text("\n".join(lines) + "\n")
    body = FLOOR + ('<Translate vector="0 1.5 0"><Rotate axis="1 0 0" angle="-60">'
                    '<Shape geometry="fan" material="shiny" /></Rotate></Translate>\n')
    text = HEAD.format(w=48, h=32, aa_on="true", aa=2, ss=3, assets=ASSETS, cubemap="").replace(
        "</Geometry>", f'<Mesh src="{obj}" name="fan" />\n</Geometry>') + body + TAIL
    path = os.path.join(str(tmp_path), "dev-fan.sdl")
    with open(path, "w") as f:
        f.write(text)
    check(render, ra.load_sdl(path))
