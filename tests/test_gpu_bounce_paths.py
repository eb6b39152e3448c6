"""Every form of the bounce pass against the CPU checker (same keyed samples).

A bounce level of a scene with area lights is shaded either by k_bounce_shade
(one lane per hit in spatial order, batches from a cursor: levels of at least
RTAMD_SMALL_LEVEL hits, 2^18 by default; shadow rays walked per lane, or as
wave-wide packets with RTAMD_BOUNCE_SHADE=packet), by k_bounce_shade_small (a
hit's light samples across the lanes of a wave: smaller levels) or by k_bounce
(the unsorted pass, RTAMD_BOUNCE=legacy; the default for scenes without area
lights; bounced rays walked per lane or as packets, RTAMD_BOUNCE_WALK).  Test
frames are small, so without the overrides below only the second form would
ever meet the checker.  The overrides are read when the scene is created.

Bar, as everywhere: hit records and ray counts per kind bit-exact, RGB8 within
+-1 LSB on >= 99.9 % of pixels, float colours within 2e-5; and because pixel sums
are fixed-point integers, every form must give the SAME frame bit for bit
(shading.c:7-58,70-112).
"""
import ctypes as C
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import raytracer_amd as ra
from raytracer_amd import _abi
import oracle_lib
from conftest import load_golden, scene_path, small_blob
from test_gpu_parity import assert_hits_equal, rgb8_within, render  # noqa: F401  (fixture)
from test_gpu_fuzz import random_scene

# name -> environment of rtamd_scene_create
FORMS = {
    "default": {},
    "per-lane-sorted": {"RTAMD_BOUNCE": "sorted", "RTAMD_SMALL_LEVEL": "0"},
    "samples-across-lanes": {"RTAMD_BOUNCE": "sorted", "RTAMD_SMALL_LEVEL": "4000000000"},
    "straddle": {"RTAMD_BOUNCE": "sorted", "RTAMD_SMALL_LEVEL": "3000"},
    "unsorted": {"RTAMD_BOUNCE": "legacy", "RTAMD_BOUNCE_WALK": "lane"},
    "unsorted-packets": {"RTAMD_BOUNCE": "legacy", "RTAMD_BOUNCE_WALK": "packet"},
    "sorted-packets": {"RTAMD_BOUNCE": "sorted", "RTAMD_SMALL_LEVEL": "0", "RTAMD_BOUNCE_SHADE": "packet"},
}


def scene_with(render, monkeypatch, blob, env):
    for k in ("RTAMD_BOUNCE", "RTAMD_SMALL_LEVEL", "RTAMD_SORT_BITS", "RTAMD_BOUNCE_WALK", "RTAMD_BOUNCE_SHADE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = render.DeviceScene(blob)
    for k in env:
        monkeypatch.delenv(k)
    return dev


_checker_cache = {}


def checker(key, blob):
    if key not in _checker_cache:
        _checker_cache[key] = oracle_lib.OracleScene(blob).render(rng_mode=_abi.RNG_KEYED, want_hits=True, threads=8)
    return _checker_cache[key]


def assert_matches_checker(frame, hits, st, want):
    colour, ohits, ost = want
    assert_hits_equal(hits, ohits)
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary, st.env_lookups) == (
        ost.rays_primary, ost.rays_shadow, ost.rays_secondary, ost.env_lookups)
    g = frame.cpu().numpy()
    assert rgb8_within(g, colour, 1) >= 0.999
    assert np.abs(g[..., :3] - colour[..., :3]).max() < 2e-5
    return g


def fixture_blob(fixture):
    g = load_golden(fixture)
    return ra.load_sdl(scene_path(str(g["scene"]))).with_config(**json.loads(str(g["config"])))


@pytest.mark.parametrize("form", [f for f in FORMS if f != "default"])
@pytest.mark.parametrize("fixture", ["spp64_7-reflections", "spp64_8-glossy"])
def test_every_form_of_the_bounce_pass_against_checker(render, monkeypatch, fixture, form):
    blob = fixture_blob(fixture)
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    dev = scene_with(render, monkeypatch, blob, FORMS[form])
    frame, hits, st = dev.render(p, want_hits=True)
    g = assert_matches_checker(frame, hits, st, checker(fixture, blob))
    assert st.rays_secondary > 0
    base = scene_with(render, monkeypatch, blob, FORMS["default"])
    want, _, _ = base.render(p)
    assert np.array_equal(g.view(np.uint32), want.cpu().numpy().view(np.uint32))
    dev.close()
    base.close()


def test_sorted_pass_on_a_scene_without_area_lights(render, monkeypatch):
    """4-environmentmap (two point lights, cubemap, reflect 0.3) takes the unsorted pass
    by default; forced through trace / sort / shade it must give the same frame."""
    blob = small_blob("4-environmentmap")
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    want = checker("small_4-environmentmap", blob)
    frames = []
    for form in ("per-lane-sorted", "samples-across-lanes", "sorted-packets", "unsorted", "unsorted-packets",
                 "default"):
        dev = scene_with(render, monkeypatch, blob, FORMS[form])
        frame, hits, st = dev.render(p, want_hits=True)
        frames.append(assert_matches_checker(frame, hits, st, want))
        assert st.env_lookups > 0 and st.rays_secondary > 0
        dev.close()
    for f in frames[:-1]:
        assert np.array_equal(f.view(np.uint32), frames[-1].view(np.uint32))


@pytest.mark.parametrize("idx", range(16))
def test_random_reflective_scenes_through_every_form(render, monkeypatch, tmp_path, idx):
    """The reflective fuzz scenes of test_gpu_fuzz (mirror and glossy materials on random
    meshes and primitives, two area lights, a point light) through the per-lane sorted
    kernel and the unsorted pass: against the checker, and bitwise against the default."""
    rng = np.random.default_rng(19 + 1000 * idx)
    blob = random_scene(rng, tmp_path, 100 + idx, reflective=True)
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    want = oracle_lib.OracleScene(blob).render(rng_mode=_abi.RNG_KEYED, want_hits=True, threads=8)
    frames = {}
    forms = ("per-lane-sorted", "sorted-packets", "unsorted", "unsorted-packets")
    for form in forms + ("default",):
        dev = scene_with(render, monkeypatch, blob, FORMS[form])
        frame, hits, st = dev.render(p, want_hits=True)
        frames[form] = assert_matches_checker(frame, hits, st, want)
        dev.close()
    for form in forms:
        assert np.array_equal(frames[form].view(np.uint32), frames["default"].view(np.uint32)), form


def test_quantisation_differences_are_last_bit_sums(render, tmp_path):
    """Reflected light is added per pixel in 2^-32 fixed point, the checker adds it in the
    recursion's float order.  Wherever the two frames' RGB8 differ, the float values must
    sit within a few float roundings of each other and on either side of a quantisation
    step: a dropped or added shadow sample would move a pixel by at least
    0.4 / 16 / 64 = 4e-4, a thousand times more."""
    differing = 0
    for idx in range(16):
        rng = np.random.default_rng(19 + 1000 * idx)
        blob = random_scene(rng, tmp_path, 100 + idx, reflective=True)
        dev = render.DeviceScene(blob)
        frame, _, _ = dev.render(render.make_params(rng_mode=_abi.RNG_KEYED))
        dev.close()
        colour, _, _ = oracle_lib.OracleScene(blob).render(rng_mode=_abi.RNG_KEYED, threads=8)
        g = frame.cpu().numpy()[..., :3].astype(np.float64)
        c = colour[..., :3].astype(np.float64)
        # ppm.c:20-24 in place (no row flip): truncation of clamp(c * 255, 0, 255) in float
        def quantise(a):
            return np.clip(a[..., :3].astype(np.float32) * np.float32(255), 0, 255).astype(np.uint8).astype(int)
        q = quantise(frame.cpu().numpy()) != quantise(colour)
        differing += int(q.sum())
        if not q.any():
            continue
        gd, cd = g[q], c[q]
        # 64 samples x (lights + children) float additions of values <= the sum: 2^-24 each
        assert (np.abs(gd - cd) <= 256 * 2.0 ** -24 * np.maximum(np.abs(cd), 2.0 ** -10)).all()
        # the two values straddle a multiple of 1/255 (ppm.c:20-24 truncates c * 255)
        lo, hi = np.minimum(gd, cd) * 255.0, np.maximum(gd, cd) * 255.0
        assert (np.floor(hi) - np.floor(lo) == 1).all()
        assert (hi - lo < 255 * 2e-5).all()
    assert differing < 16 * 32 * 20 * 3 * 0.02


def test_noalloc_through_the_per_lane_sorted_kernel(render, monkeypatch):
    """rtamd_render_noalloc lays the sort scratch out in the static array with at most 6
    grid bits per axis; with every level sent to k_bounce_shade it must still give the
    frame of rtamd_render and of the checker."""
    import torch
    blob = ra.load_sdl(scene_path("8-glossy")).with_config(width=40, height=32, antialiasing=1, aa_samples=4)
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    dev = scene_with(render, monkeypatch, blob, FORMS["per-lane-sorted"])
    want, hits, st = dev.render(p, want_hits=True)
    assert_matches_checker(want, hits, st, checker("noalloc-8-glossy", blob))
    got = torch.full((blob.height, blob.width, 4), -7.0, dtype=torch.float32, device=dev.device)
    s = torch.cuda.current_stream(dev.device)
    _abi.check(_abi.lib().rtamd_render_noalloc(dev._handle, C.byref(p), got.data_ptr(), s.cuda_stream),
               "rtamd_render_noalloc")
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.cpu().numpy().view(np.uint32))
    dev.close()


@pytest.mark.parametrize("name,least_secondary", [("8-glossy", 10 << 21), ("7-reflections", 1 << 21)])
def test_frame_large_enough_for_the_default_dispatch(render, monkeypatch, name, least_secondary):
    """The reflective benchmark scenes at their full 640 x 640 with 64 samples per pixel.  8-glossy:
    one chunk, about 20 M rays of depth 1 and as many deeper ones, so the default dispatch
    sends the dense levels to the packet form of k_bounce_shade (from 2^18 hits) and the
    thin ones to k_bounce_shade_small, all after the radix sort of the hits.  A window of the frame is compared with the checker, the
    whole frame and the ray counts with the unsorted pass, bit for bit."""
    blob = ra.load_sdl(scene_path(name)).with_config(antialiasing=1, aa_samples=8)
    assert (blob.width, blob.height, blob.spp) == (640, 640, 64)
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    dev = scene_with(render, monkeypatch, blob, FORMS["default"])
    frame, _, st = dev.render(p)
    assert st.rays_secondary > least_secondary
    assert st.launches_primary == 1
    win = (300, 250, 316, 262)
    colour, _, _ = oracle_lib.OracleScene(blob).render(rng_mode=_abi.RNG_KEYED, window=win, threads=8)
    g = frame.cpu().numpy()
    got, want = g[win[1]:win[3], win[0]:win[2]], colour[win[1]:win[3], win[0]:win[2]]
    assert want[..., :3].max() > 0.05
    assert rgb8_within(got, want, 1) >= 0.999
    assert np.abs(got[..., :3] - want[..., :3]).max() < 2e-5
    dev.close()
    legacy = scene_with(render, monkeypatch, blob, FORMS["unsorted"])
    other, _, st2 = legacy.render(p)
    assert np.array_equal(other.cpu().numpy().view(np.uint32), g.view(np.uint32))
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary, st.env_lookups) == (
        st2.rays_primary, st2.rays_shadow, st2.rays_secondary, st2.env_lookups)
    legacy.close()
