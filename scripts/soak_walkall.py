"""GPU box: many random scenes (the generator of tests/test_gpu_fuzz.py, larger frames) rendered
with the product library and with librtamd_walkall.so (every shadow ray walked, unlit lights
kept); the frames must agree bit for bit and the shadow-ray counts exactly.

    python3 scripts/soak_walkall.py [scenes=48] [first_seed=1000]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import raytracer_amd as ra                    # noqa: E402
from raytracer_amd import _abi, render        # noqa: E402
from test_gpu_fuzz import random_scene        # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 48
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
tmp = tempfile.mkdtemp(prefix="soak")
paths = []
for i in range(count):
    rng = np.random.default_rng(first + i)
    blob = random_scene(rng, tmp, i, reflective=(i % 2 == 1))
    path = os.path.join(tmp, f"fuzz-{i}.sdl")
    # larger frames than the tests': more waves whose samples all face away
    text = open(path).read().replace('width="32"', 'width="160"').replace('height="20"', 'height="100"')
    text = text.replace('width="40"', 'width="200"').replace('height="24"', 'height="120"')
    open(path, "w").write(text)
    paths.append(path)
walkall = os.path.join(os.path.dirname(_abi.LIB_PATH), "librtamd_walkall.so")
out = os.path.join(tmp, "walked.npz")
subprocess.run([sys.executable, os.path.join(ROOT, "tests", "render_frames.py"), out] + paths,
               env=dict(os.environ, RTAMD_LIB=walkall, RTAMD_KEEP_UNLIT_LIGHTS="1"), check=True)
walked = np.load(out, allow_pickle=False)
bad = 0
rays = 0
for i, path in enumerate(paths):
    name = os.path.splitext(os.path.basename(path))[0]
    dev = render.DeviceScene(ra.load_sdl(path))
    frame, _, st = dev.render(render.make_params(rng_mode=_abi.RNG_KEYED))
    dev.close()
    same = np.array_equal(walked[name].view(np.uint32), frame.cpu().numpy().view(np.uint32))
    same_count = int(walked[name + ":shadow"][0]) == st.rays_shadow
    rays += st.rays_shadow
    if not (same and same_count):
        bad += 1
        print(f"{name}: frames {'equal' if same else 'DIFFER'}, shadow counts {'equal' if same_count else 'DIFFER'}")
print(f"{count} scenes, {rays} shadow queries, {bad} disagreements")
sys.exit(1 if bad else 0)
