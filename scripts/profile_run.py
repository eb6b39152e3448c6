"""Profiling helper: renders the showcase scene a few times at a chosen size
so rocprofv3 has a short, kernel-dominated run to attach counters to."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import raytracer_amd as ra                 # noqa: E402
from raytracer_amd import _abi, render     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="showcase-uffizi")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--aa", type=int, default=8)
ap.add_argument("--repeat", type=int, default=2)
args = ap.parse_args()

blob = ra.load_sdl(os.path.join(ROOT, "assets", args.scene + ".sdl"))
kw = dict(width=args.width, height=args.height)
if args.aa > 0:
    kw.update(antialiasing=1, aa_samples=args.aa)
blob = blob.with_config(**kw)
scene = render.DeviceScene(blob)
p = render.make_params(rng_mode=_abi.RNG_KEYED)
for _ in range(args.repeat):
    frame, _, st = scene.render(p)
print(f"{args.scene} {blob.width}x{blob.height} spp={blob.spp} rays={render.total_rays(st)} "
      f"nodes/ray={st.bvh_nodes / max(1, render.total_rays(st)):.2f} tris/ray={st.tri_tests / max(1, render.total_rays(st)):.2f} "
      f"ms={st.ms_total:.3f} primary_ms={st.ms_primary:.3f} Mrays/s={render.total_rays(st) / st.ms_total / 1e3:.1f}")
