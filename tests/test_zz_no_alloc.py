"""rtamd_render_noalloc (the entry point that takes all its work memory from the
library's static per-GPU scratch array) against rtamd_render, through the C ABI:
the frames must be bit-identical, on two calls in a row, the second on a
smaller input - the array keeps the first call's contents, so whatever must
start at zero has to be cleared by the call itself."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import raytracer_amd as ra
from raytracer_amd import _abi
from conftest import scene_path


def frames(render, blob, params):
    import torch
    dev = render.DeviceScene(blob)
    want, _, _ = dev.render(params)
    got = torch.full((blob.height, blob.width, 4), -7.0, dtype=torch.float32, device=dev.device)
    s = torch.cuda.current_stream(dev.device)
    _abi.check(_abi.lib().rtamd_render_noalloc(dev._handle, C.byref(params), got.data_ptr(), s.cuda_stream),
               "rtamd_render_noalloc")
    torch.cuda.synchronize()
    out = got.cpu().numpy(), want.cpu().numpy()
    dev.close()
    return out


def test_noalloc_matches_render_twice_second_smaller():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from raytracer_amd import render
    render.init(0)
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    # glossy + mirror + area lights: accumulators, both queues and the sort scratch are used
    big = ra.load_sdl(scene_path("8-glossy")).with_config(width=40, height=32, antialiasing=1, aa_samples=4)
    got, want = frames(render, big, p)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert want[..., :3].max() > 0.1
    small = ra.load_sdl(scene_path("7-reflections")).with_config(width=17, height=9, antialiasing=1, aa_samples=3)
    got, want = frames(render, small, p)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # a scene without reflections writes pixels directly (no scratch besides the counters)
    direct = ra.load_sdl(scene_path("6-soft")).with_config(width=24, height=12, antialiasing=1, aa_samples=2)
    got, want = frames(render, direct, p)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_noalloc_rejects_what_does_not_fit():
    import torch
    from raytracer_amd import render
    render.init(0)
    blob = ra.load_sdl(scene_path("8-glossy")).with_config(width=4096, height=2048)
    dev = render.DeviceScene(blob)
    frame = dev.new_frame()
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    rc = _abi.lib().rtamd_render_noalloc(dev._handle, C.byref(p), frame.data_ptr(), None)
    assert rc == -6 and b"scratch" in _abi.lib().rtamd_last_error()
    dev.close()


def test_noalloc_serial_stream_mode():
    """The reference's rand() stream (generated on the device into the scratch array),
    two seeds in a row: the second call must not reuse the first seed's tables."""
    from raytracer_amd import render
    render.init(0)
    blob = ra.load_sdl(scene_path("3-aa")).with_config(width=70, height=40)
    for seed in (0x20071208, 99):
        got, want = frames(render, blob, render.make_params(seed=seed, rng_mode=_abi.RNG_GLIBC))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_noalloc_accepts_small_calls_like_render_does():
    """A call whose samples could spawn fewer rays than the queue floor (one 32 x 32 tile
    of 8-glossy without antialiasing, reflection_samples 8) and an owner without tiles
    in the frame: rtamd_render renders both, so must the entry point that allocates
    nothing (its queue capacity is floored the same way, and no pixels is no work)."""
    import torch
    from raytracer_amd import render
    render.init(0)
    blob = ra.load_sdl(scene_path("8-glossy")).with_config(width=32, height=32, antialiasing=0)
    assert blob.header.reflection_samples > 4
    got, want = frames(render, blob, render.make_params(rng_mode=_abi.RNG_KEYED))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert want[..., :3].max() > 0.1
    # one tile in the frame, owner 1 of 2 has none
    dev = render.DeviceScene(blob)
    p = render.make_params(rng_mode=_abi.RNG_KEYED, tile_first=1, tile_stride=2)
    frame = torch.full((32, 32, 4), -7.0, dtype=torch.float32, device=dev.device)
    rc = _abi.lib().rtamd_render_noalloc(dev._handle, C.byref(p), frame.data_ptr(), None)
    assert rc == 0, _abi.lib().rtamd_last_error()
    torch.cuda.synchronize()
    assert float(frame.min()) == -7.0 and float(frame.max()) == -7.0
    dev.render(p, frame=frame)
    assert float(frame.min()) == -7.0 and float(frame.max()) == -7.0
    dev.close()
