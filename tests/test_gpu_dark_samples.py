"""Light samples that add exactly nothing are counted but not walked (sample_is_dark).

Whether a sample may be left out is decided on the values of the two Phong terms
(lighting.c:7-34), so the cases to pin are the materials and lights for which a surface
that faces away from the light still receives something, or receives nothing at all:
  * a low specular exponent: pow(H.N, e) does not underflow, the sample must be walked
    (an occluder between the surface and the light then decides);
  * exponent 0: pow(0, 0) = 1, every sample of every surface counts;
  * a light of intensity 0: every sample is dark, on lit and unlit sides alike (a point light
    of that kind is left out of the device's light list altogether, its queries counted:
    RTAMD_KEEP_UNLIT_LIGHTS=1 keeps it, which is what the comparison build runs with);
  * a negative light colour: the zero diffuse term is -0, which the sum must not see;
  * the scenes' own exponent 400 on bodies that hide half of themselves from the light.
Against the CPU checker with the same keyed samples: hit records and ray counts exact,
float colours within 2e-5 and RGB8 within 1 LSB as everywhere; and against the build of the
same library that walks every shadow ray (librtamd_walkall.so): the same frame bit for bit.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import raytracer_amd as ra
from raytracer_amd import _abi
import oracle_lib
from conftest import ASSETS
from test_gpu_parity import assert_hits_equal, rgb8_within, render  # noqa: F401  (fixture)

SCENE = """<?xml version='1.0' encoding='utf-8'?>
<Sdl>
<Config width="40" height="32" antialiasing="true" aa_samples="{aa}" shadow_samples="3"
	reflection_samples="2" max_reflections="2" />
<Cameras><Camera position="0 3 9" direction="0 -2 -8" up="0 1 0" fovy="55" name="cam" /></Cameras>
<Lights>
<PointLight position="-6 1.5 -5" intensity="{point_intensity}" color="{point_colour}" name="low" />
<AreaLight position="-1 7 -9" edge1="2 0 0" edge2="0 0 2" intensity="{area_intensity}" color="{area_colour}" name="back" />
</Lights>
<Geometry>
<Plane edge1="0 0 24" edge2="24 0 0" name="floor" />
<Sphere radius="1.2" name="ball" />
<Cylinder radius="0.7" height="2" capped="true" name="can" />
<Mesh src="{assets}/torus.obj" name="torus" />
<Mesh src="{assets}/cube.obj" name="cube" />
</Geometry>
<Materials>
<Material diffuse_color="0.7 0.7 0.7" specular_color="1 1 1" specular_exponent="{e_floor}" name="floor" />
<Material diffuse_color="0.9 0.4 0.2" specular_color="1 1 1" specular_exponent="{e_body}" reflect="{reflect}" name="body" />
</Materials>
<Scene camera="cam" lights="{lights}" background="0.1 0.1 0.2">
<Translate vector="0 0 0">
<Translate vector="-12 0 -12"><Shape geometry="floor" material="floor" /></Translate>
<Translate vector="-2.2 1.2 0"><Shape geometry="ball" material="body" /></Translate>
<Translate vector="2.4 0 0.5"><Shape geometry="can" material="body" /></Translate>
<Translate vector="0 1.1 1.5"><Rotate axis="1 0 0" angle="70"><Shape geometry="torus" material="body" /></Rotate></Translate>
<Translate vector="0.3 0.5 -2.5"><Scale scale="1.5 1.5 1.5"><Shape geometry="cube" material="body" /></Scale></Translate>
</Translate>
</Scene>
</Sdl>
"""

DEFAULTS = dict(aa=2, point_intensity="0.6", point_colour="1 1 1", area_intensity="0.8", area_colour="1 1 1",
                e_floor=400, e_body=400, reflect="0.0", lights="low,back")


# name -> what differs from DEFAULTS
CASES = {f"exp-{e}": dict(e_body=e, e_floor=e) for e in (0, 1, 2, 8, 60, 400)}
CASES.update({
    "area-off": dict(area_intensity="0.0"),
    "all-off": dict(area_intensity="0.0", point_intensity="0"),
    "point-off-last": dict(point_intensity="0", lights="back,low"),
    "point-off-only": dict(point_intensity="0", lights="low"),
    "negative": dict(area_colour="-1 0.5 -0.25", point_colour="1 -1 1"),
    "refl-2": dict(aa=2, reflect="0.4", e_body=8),
    "plain-8": dict(aa=8, e_body=8),
    "plain-3": dict(aa=3, e_body=8),
})
WALKALL = os.path.join(os.path.dirname(_abi.LIB_PATH), "librtamd_walkall.so")


@pytest.fixture(scope="module")
def cases(tmp_path_factory):
    """The scene files, and their frames from the build that walks every shadow ray
    (librtamd_walkall.so: the same sources with -DRT_SKIP_DARK=0), rendered in a process of
    its own because the library is chosen at import."""
    root = tmp_path_factory.mktemp("dark")
    paths = {}
    for name, kw in CASES.items():
        args = dict(DEFAULTS, assets=ASSETS)
        args.update(kw)
        paths[name] = os.path.join(str(root), name + ".sdl")
        with open(paths[name], "w") as f:
            f.write(SCENE.format(**args))
    assert os.path.exists(WALKALL), "librtamd_walkall.so is not built (make -C raytracer_amd/csrc)"
    out = os.path.join(str(root), "walked.npz")
    env = dict(os.environ, RTAMD_LIB=WALKALL, RTAMD_KEEP_UNLIT_LIGHTS="1")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_frames.py")
    subprocess.run([sys.executable, script, out] + list(paths.values()), env=env, check=True, timeout=300)
    walked = np.load(out, allow_pickle=False)
    assert str(walked["lib"][0]) == WALKALL
    return paths, walked


def check(render, cases, name):
    paths, walked = cases
    blob = ra.load_sdl(paths[name])
    dev = render.DeviceScene(blob)
    frame, hits, st = dev.render(render.make_params(rng_mode=_abi.RNG_KEYED), want_hits=True)
    colour, ohits, ost = oracle_lib.OracleScene(blob).render(rng_mode=_abi.RNG_KEYED, want_hits=True, threads=8)
    assert_hits_equal(hits, ohits)
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary, st.env_lookups) == (
        ost.rays_primary, ost.rays_shadow, ost.rays_secondary, ost.env_lookups)
    g = frame.cpu().numpy()
    assert rgb8_within(g, colour, 1) >= 0.999
    assert np.abs(g[..., :3] - colour[..., :3]).max() < 2e-5
    dev.close()
    # every shadow ray walked: the same frame, the same count
    assert np.array_equal(walked[name].view(np.uint32), g.view(np.uint32))
    assert int(walked[name + ":shadow"][0]) == st.rays_shadow
    return g, st


@pytest.mark.parametrize("exponent", [0, 1, 2, 8, 60, 400])
def test_specular_exponents_on_surfaces_that_face_away(render, cases, exponent):
    """The back light sits behind the bodies: the camera sees their far sides.  With a small
    exponent those still receive specular light unless the body itself is in the way."""
    g, st = check(render, cases, f"exp-{exponent}")
    assert st.rays_shadow > 0 and g[..., :3].max() > 0.05


def test_lights_that_give_nothing(render, cases):
    """Intensity 0: every sample of that light is dark on every surface; both lights at 0
    leave a black frame with the full count of shadow queries."""
    g, st = check(render, cases, "area-off")
    g2, st2 = check(render, cases, "all-off")
    assert st2.rays_shadow == st.rays_shadow
    # only the background is left, and surfaces are black
    assert g2[..., :3].max() <= 0.2 + 1e-6
    assert (g2[..., :3].sum(axis=2) == 0).any()


def test_unlit_point_light_left_out_of_the_device_list(render, cases):
    """A zero-intensity point light behind the last area light is not walked at all (its
    queries are added to the count on readback); before an area light it stays in the list,
    and as the only light it stays too, so that shaded hits can be counted."""
    g, st = check(render, cases, "point-off-last")
    g2, st2 = check(render, cases, "all-off")
    assert st.rays_shadow == st2.rays_shadow
    g3, st3 = check(render, cases, "point-off-only")
    assert st3.rays_shadow > 0 and st3.rays_shadow * 10 == st.rays_shadow


def test_negative_light_colour_and_signed_zero(render, cases):
    """0 x (negative colour) = -0 for the diffuse term of a surface that faces away; the sums
    start at +0 and must end as the reference's do."""
    check(render, cases, "negative")


@pytest.mark.parametrize("name", ["refl-2", "plain-8", "plain-3"])
def test_reflective_bodies_and_other_sample_counts(render, cases, name):
    """With reflections the bounce kernels shade the far sides too (depth 1 hits behind the
    bodies); 64 and 9 samples per pixel take the ordered and the fixed-point pixel sum."""
    g, st = check(render, cases, name)
    if name == "refl-2":
        assert st.rays_secondary > 0


def test_benchmark_scenes_at_their_own_size(render, tmp_path):
    """The scenes the skipping matters for (a third of 8-glossy's shadow queries, a fifth of
    7-reflections'), at the size and sample counts of their files: product against the
    walk-everything build, whole frame bit for bit, same ray counts."""
    from conftest import scene_path
    names = ["8-glossy", "7-reflections", "6-soft", "5-primitives", "3-aa"]
    out = os.path.join(str(tmp_path), "walked.npz")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_frames.py")
    subprocess.run([sys.executable, script, out] + [scene_path(n) for n in names],
                   env=dict(os.environ, RTAMD_LIB=WALKALL, RTAMD_KEEP_UNLIT_LIGHTS="1"), check=True, timeout=600)
    walked = np.load(out, allow_pickle=False)
    assert str(walked["lib"][0]) == WALKALL
    for name in names:
        dev = render.DeviceScene(ra.load_sdl(scene_path(name)))
        frame, _, st = dev.render(render.make_params(rng_mode=_abi.RNG_KEYED))
        dev.close()
        assert np.array_equal(walked[name].view(np.uint32), frame.cpu().numpy().view(np.uint32)), name
        assert int(walked[name + ":shadow"][0]) == st.rays_shadow, name
