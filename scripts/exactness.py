"""Reports how many pixels of the full-size deterministic frames are bit-identical
(RGB8) to the compiled reference's images, and the PPM checksums."""
import hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra
from raytracer_amd import _abi, render
md5 = json.load(open(os.path.join(ROOT, "tests/golden/full_md5.json")))
for name in ["1-phong", "2-transform", "3-aa", "4-environmentmap", "5-primitives"]:
    blob = ra.load_sdl(os.path.join(ROOT, "assets", name + ".sdl"))
    dev = render.DeviceScene(blob)
    frame, _, st = dev.render(render.make_params(rng_mode=_abi.RNG_GLIBC))
    got = ra.quantise_rgb8(frame.cpu().numpy())
    want = np.load(os.path.join(ROOT, "tests/golden", f"full_{name}.npz"))["rgb8"]
    ppm = f"P6\n{blob.width} {blob.height}\n255\n".encode() + got.tobytes()
    d = np.abs(got.astype(int) - want.astype(int))
    print(name, "exact px", int((d == 0).all(axis=2).sum()), "of", d.shape[0] * d.shape[1], "max diff", d.max(),
          "md5", hashlib.md5(ppm).hexdigest(), "== reference md5:", hashlib.md5(ppm).hexdigest() == md5[name]["md5_ieee"],
          f"ms={st.ms_total:.2f}")
