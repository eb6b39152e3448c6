"""Renders scene files with the library RTAMD_LIB names and stores the float frames.

    RTAMD_LIB=<librtamd variant> python tests/render_frames.py <out.npz> <scene.sdl> ...

The library is chosen when raytracer_amd is imported, so a test that wants the frames of
another build of it runs this in a process of its own (keyed samples, whole frame).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import raytracer_amd as ra                    # noqa: E402
from raytracer_amd import _abi, render        # noqa: E402

out, paths = sys.argv[1], sys.argv[2:]
frames = {}
for path in paths:
    name = os.path.splitext(os.path.basename(path))[0]
    dev = render.DeviceScene(ra.load_sdl(path))
    frame, _, st = dev.render(render.make_params(rng_mode=_abi.RNG_KEYED))
    frames[name] = frame.cpu().numpy()
    frames[name + ":shadow"] = np.array([st.rays_shadow], dtype=np.uint64)
    dev.close()
np.savez(out, lib=np.array([_abi.LIB_PATH]), **frames)
