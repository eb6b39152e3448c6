"""Randomised scenes against the CPU checker (same keyed samples).  The light
shafts, the packet box tests and the early triangle rejections are conservative
filters with rounding margins; a filter that drops something it must keep shows
up here as a shadow (or a hit) that differs from the checker's.  Meshes are
placed under random rotations and non-uniform scales so that the general
affine path of the shaft construction is exercised, lights and cameras move,
and 64 samples per pixel keep the production mapping (one wave = one pixel)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import raytracer_amd as ra
from raytracer_amd import _abi
import oracle_lib
from conftest import ASSETS
from test_gpu_parity import assert_hits_equal, rgb8_within, render  # noqa: F401  (fixture)

TEMPLATE = """<?xml version='1.0' encoding='utf-8'?>
<Sdl>
<Config width="{w}" height="{h}" antialiasing="true" aa_samples="{aa}" shadow_samples="{ss}"
	reflection_samples="2" max_reflections="2" />
<Cameras><Camera position="{cam}" direction="{cdir}" up="0 1 0" fovy="{fov}" name="cam" /></Cameras>
<Lights>
<PointLight position="{pl}" intensity="0.6" color="1 1 1" name="point" />
<AreaLight position="{al}" edge1="{e1}" edge2="{e2}" intensity="0.7" name="area" />
<AreaLight position="{al2}" edge1="{e3}" edge2="{e4}" intensity="0.4" name="area2" />
</Lights>
<Geometry>
<Plane edge1="0 0 24" edge2="24 0 0" name="floor" />
<Sphere radius="0.8" name="ball" />
<Cylinder radius="0.5" height="1.5" capped="true" name="can" />
<Mesh src="{assets}/torus.obj" name="torus" />
<Mesh src="{assets}/cube.obj" name="cube" />
<Mesh src="{assets}/teapot.obj" name="teapot" />
<Mesh src="{assets}/sphere.obj" name="msphere" />
</Geometry>
<Materials>
<Material diffuse_color="0.8 0.8 0.8" name="grey" />
<Material diffuse_color="0.9 0.4 0.2" specular_exponent="80" name="orange" />
<Material diffuse_color="0.5 0.6 0.7" specular_color="0.9 0.9 1" specular_exponent="120" reflect="0.45" name="mirror" />
<Material diffuse_color="0.3 0.8 0.4" specular_color="1 1 1" specular_exponent="60" reflect="0.5" glossiness="{gloss}" name="glossy" />
</Materials>
<Scene camera="cam" lights="point,area,area2" background="0.2 0.3 0.4">
<Translate vector="0 0 0">
<Translate vector="-12 0 -12"><Shape geometry="floor" material="{floor_mat}" /></Translate>
{body}</Translate>
</Scene>
</Sdl>
"""


def vec(v):
    return " ".join(f"{x:.6g}" for x in v)


def random_scene(rng, tmp_path, idx, reflective=False):
    """reflective: objects (and sometimes the floor) also draw from a mirror and a
    glossy material, so bounced rays - the bounce kernel's nearest-hit walks, its
    light loops and the children of depth 1 - meet the same random geometry."""
    body = ""
    shapes = ["torus", "cube", "teapot", "msphere", "ball", "can"]
    mats = ["grey", "orange", "mirror", "glossy"] if reflective else ["grey", "orange"]
    floor_mat = mats[int(rng.integers(0, len(mats)))] if reflective and rng.random() < 0.5 else "grey"
    for k in range(int(rng.integers(3, 7))):
        g = shapes[int(rng.integers(0, len(shapes)))]
        pos = rng.uniform([-4, 0.3, -4], [4, 2.5, 4])
        axis = rng.normal(size=3)
        scale = rng.uniform(0.4, 1.6, size=3) if rng.random() < 0.7 else np.ones(3)
        if g == "teapot":
            scale = scale * 0.5
        body += (f'<Translate vector="{vec(pos)}"><Rotate axis="{vec(axis)}" angle="{rng.uniform(0, 360):.4g}">'
                 f'<Scale scale="{vec(scale)}"><Shape geometry="{g}" material="{mats[int(rng.integers(0, len(mats)))] if reflective else ("orange" if k % 2 else "grey")}" />'
                 f'</Scale></Rotate></Translate>\n')
    cam = rng.uniform([-7, 2, 5], [7, 8, 10])
    target = rng.uniform([-1.5, 0, -1.5], [1.5, 1.5, 1.5])
    # area lights: one overhead, one low and slanted (grazing shafts)
    al = rng.uniform([-5, 5, -5], [5, 11, 5])
    e1 = rng.uniform(-2, 2, size=3)
    e2 = np.cross(e1, rng.normal(size=3))
    e2 = e2 / np.linalg.norm(e2) * rng.uniform(0.5, 2.5)
    al2 = rng.uniform([-9, 0.5, -9], [9, 3, 9])
    e3 = rng.uniform(-1.5, 1.5, size=3)
    e4 = rng.uniform(-1.5, 1.5, size=3)
    text = TEMPLATE.format(w=32 if reflective else 40, h=20 if reflective else 24, aa=8, ss=int(rng.integers(2, 5)),
                           gloss=f"{rng.uniform(0.05, 0.4):.3g}" if reflective else "0.2", floor_mat=floor_mat, cam=vec(cam), cdir=vec(target - cam),
                           fov=f"{rng.uniform(35, 75):.4g}", pl=vec(rng.uniform([-6, 3, -6], [6, 10, 6])),
                           al=vec(al), e1=vec(e1), e2=vec(e2), al2=vec(al2), e3=vec(e3), e4=vec(e4),
                           assets=ASSETS, body=body)
    path = os.path.join(str(tmp_path), f"fuzz-{idx}.sdl")
    with open(path, "w") as f:
        f.write(text)
    return ra.load_sdl(path)


@pytest.mark.parametrize("idx", range(24))
def test_random_scene_matches_checker(render, tmp_path, idx):
    rng = np.random.default_rng(20071208 + idx)
    blob = random_scene(rng, tmp_path, idx)
    dev = render.DeviceScene(blob)
    frame, hits, st = dev.render(render.make_params(rng_mode=_abi.RNG_KEYED), want_hits=True)
    colour, ohits, ost = oracle_lib.OracleScene(blob).render(rng_mode=_abi.RNG_KEYED, want_hits=True, threads=8)
    assert_hits_equal(hits, ohits)
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary) == (ost.rays_primary, ost.rays_shadow, ost.rays_secondary)
    g = frame.cpu().numpy()
    # one light sample that is shadowed on one side only moves a pixel by at least
    # 0.4 / 16 / 64 = 4e-4; what is allowed here is the last-bit difference between
    # the device's and the host's powf
    assert np.abs(g[..., :3] - colour[..., :3]).max() < 2e-5
    assert rgb8_within(g, colour, 0) >= 0.999
    dev.close()


@pytest.mark.parametrize("idx", range(16))
def test_random_reflective_scene_matches_checker(render, tmp_path, idx):
    """Mirror and glossy materials on meshes and primitives: every bounced ray, its
    shadow rays and its children against the checker (same keyed samples)."""
    rng = np.random.default_rng(19 + 1000 * idx)
    blob = random_scene(rng, tmp_path, 100 + idx, reflective=True)
    dev = render.DeviceScene(blob)
    frame, hits, st = dev.render(render.make_params(rng_mode=_abi.RNG_KEYED), want_hits=True)
    colour, ohits, ost = oracle_lib.OracleScene(blob).render(rng_mode=_abi.RNG_KEYED, want_hits=True, threads=8)
    assert_hits_equal(hits, ohits)
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary, st.env_lookups) == (
        ost.rays_primary, ost.rays_shadow, ost.rays_secondary, ost.env_lookups)
    assert st.rays_secondary > 0
    g = frame.cpu().numpy()
    assert np.abs(g[..., :3] - colour[..., :3]).max() < 2e-5
    # reflected light is added per pixel in fixed point, not in the recursion's order:
    # a last-bit difference can cross a quantisation step, so the bar is the stated
    # +-1 LSB (and nearly all pixels are equal)
    assert rgb8_within(g, colour, 1) >= 0.999
    assert rgb8_within(g, colour, 0) >= 0.98
    dev.close()
