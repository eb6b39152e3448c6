"""Generates the fixtures under tests/golden/ from the REFERENCE's own code.

Runs only where /root/reference exists (the build container): it drives
oracle/_ref/ref_harness, which is the reference's ray.c / shading.c / lighting.c /
texture.c / mesh.c compiled from where they lie (oracle/Makefile).  The fixtures
are data: scene parameters, rays, and the reference's outputs for them.

    python tests/golden/make_golden.py [--full]

--full also renders the five deterministic scenes at their full size (a few
minutes) and stores their RGB8 images and PPM checksums.
"""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import raytracer_amd as ra   # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
REF_FAST = os.path.join(ROOT, "oracle", "_ref", "ref_harness_fastmath")
HIT_DTYPE = np.dtype([("surface", "<i4"), ("triangle", "<i4"), ("t", "<f4"), ("pad", "<f4"),
                      ("position", "<f8", 3), ("normal", "<f8", 3)])

# scene -> reduced <Config> used for the small fixtures
SMALL = {
    "1-phong": dict(width=96, height=96),
    "2-transform": dict(width=96, height=96),
    "3-aa": dict(width=32, height=32),
    "4-environmentmap": dict(width=48, height=48),
    "5-primitives": dict(width=128, height=64),
    "6-soft": dict(width=48, height=24),
    "7-reflections": dict(width=32, height=32),
    "8-glossy": dict(width=24, height=24),
    "showcase-uffizi": dict(width=64, height=36, antialiasing=1, aa_samples=4),
    "cornell": dict(width=64, height=64),
    "prims": dict(),
    "prims-aa": dict(width=36, height=24, antialiasing=1, aa_samples=3),
}
FULL = ["1-phong", "2-transform", "3-aa", "4-environmentmap", "5-primitives"]
# the stochastic scenes at 64 samples per pixel (aa_samples = 8), reduced frames:
# fixture name -> (scene, <Config> overrides)
SPP64 = {
    "spp64_6-soft": ("6-soft", dict(width=32, height=16, antialiasing=1, aa_samples=8)),
    "spp64_8-glossy": ("8-glossy", dict(width=20, height=20, antialiasing=1, aa_samples=8)),
    "spp64_7-reflections": ("7-reflections", dict(width=20, height=20, antialiasing=1, aa_samples=8)),
}


def scene_path(name):
    if name.startswith("prims"):
        return os.path.join(ROOT, "tests", "scenes", "prims.sdl")
    return os.path.join(ROOT, "assets", name + ".sdl")


def small_blob(name):
    return ra.load_sdl(scene_path(name)).with_config(**SMALL[name])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_ref(exe, blob, tmp, *extra):
    bpath = os.path.join(tmp, "scene.blob")
    blob.save(bpath)
    prefix = os.path.join(tmp, "out")
    out = subprocess.run([exe, "render", bpath, prefix, *extra], check=True, capture_output=True,
                         text=True).stdout
    next_rand = int(out.split()[1])
    colour = np.fromfile(prefix + ".colour.f32", np.float32).reshape(blob.height, blob.width, 4)
    return prefix, colour, next_rand


def make_small():
    for name in SMALL:
        blob = small_blob(name)
        with tempfile.TemporaryDirectory() as tmp:
            prefix, colour, next_rand = run_ref(REF, blob, tmp, "--hits", "--no-ppm")
            hits = np.fromfile(prefix + ".hits.bin", HIT_DTYPE)
        np.savez_compressed(
            os.path.join(HERE, f"small_{name}.npz"),
            config=json.dumps(SMALL[name]), colour=colour[..., :3].copy(),
            surface=hits["surface"].astype(np.int8), t=hits["t"],
            sha_position=sha(hits["position"]), sha_normal=sha(hits["normal"]),
            next_rand=next_rand)
        print("small", name, colour.shape, len(hits), "next_rand", next_rand)


def make_spp64():
    """Reference images (serial rand() stream) of the stochastic scenes at 64 spp."""
    for fixture, (name, cfg) in SPP64.items():
        blob = ra.load_sdl(scene_path(name)).with_config(**cfg)
        with tempfile.TemporaryDirectory() as tmp:
            _, colour, next_rand = run_ref(REF, blob, tmp, "--no-ppm")
        np.savez_compressed(os.path.join(HERE, fixture + ".npz"), scene=name, config=json.dumps(cfg),
                            colour=colour[..., :3].copy(), next_rand=next_rand)
        print(fixture, colour.shape, "next_rand", next_rand)


def make_rays():
    """Single-ray vectors for ray_intersect: random rays plus edge cases
    (axis-parallel directions, origins inside shapes, short windows)."""
    rng = np.random.default_rng(20071208)
    for name in ("prims", "5-primitives"):
        blob = small_blob(name)
        n = 4096
        origin = rng.uniform(-8, 8, (n, 3))
        origin[:, 1] = rng.uniform(-1, 10, n)
        target = rng.uniform(-5, 5, (n, 3))
        target[:, 1] = rng.uniform(0, 3, n)
        direction = target - origin
        # a third of them normalised, the rest left at arbitrary length
        norm = np.linalg.norm(direction, axis=1, keepdims=True)
        direction[: n // 3] /= norm[: n // 3]
        near = np.zeros(n)
        far = np.full(n, np.inf)
        far[n // 2:] = rng.uniform(0.2, 3.0, n - n // 2)
        near[3 * n // 4:] = rng.uniform(0.0, 0.5, n - 3 * n // 4)
        # edge cases: axis-parallel rays
        k = 0
        for axis in range(3):
            for sign in (-1.0, 1.0):
                for _ in range(40):
                    d = np.zeros(3)
                    d[axis] = sign * rng.uniform(0.5, 2.0)
                    direction[k] = d
                    origin[k] = rng.uniform(-4, 4, 3) - 12 * d / np.abs(d).max()
                    near[k], far[k] = 0.0, np.inf
                    k += 1
        # rays starting inside the big sphere / cylinder of the scene
        for _ in range(100):
            origin[k] = rng.uniform(-0.3, 0.3, 3) + (np.array([-1, 1.2, 1]) if name == "prims"
                                                   else np.array([0, 1, 2]))
            direction[k] = rng.normal(size=3)
            near[k], far[k] = 0.0, np.inf
            k += 1
        rays = np.concatenate([origin, direction, near[:, None], far[:, None]], axis=1)
        with tempfile.TemporaryDirectory() as tmp:
            bpath = os.path.join(tmp, "scene.blob")
            rpath = os.path.join(tmp, "rays.f64")
            opath = os.path.join(tmp, "hits.bin")
            blob.save(bpath)
            rays.astype(np.float64).tofile(rpath)
            subprocess.run([REF, "trace", bpath, rpath, opath], check=True)
            hits = np.fromfile(opath, HIT_DTYPE)
        np.savez_compressed(os.path.join(HERE, f"rays_{name}.npz"), rays=rays,
                            surface=hits["surface"], t=hits["t"], position=hits["position"],
                            normal=hits["normal"])
        print("rays", name, n, "hits", int((hits["surface"] >= 0).sum()))


def make_rand():
    libc = ctypes.CDLL("libc.so.6")
    out = {}
    for seed in (0x20071208, 1, 12345, 0):
        libc.srand(seed)
        out[f"seed_{seed}"] = np.array([libc.rand() for _ in range(400)], np.int32)
    np.savez_compressed(os.path.join(HERE, "glibc_rand.npz"), **out)
    print("rand ok")


def make_full():
    record = {}
    for name in FULL:
        blob = ra.load_sdl(scene_path(name))
        with tempfile.TemporaryDirectory() as tmp:
            prefix, colour, next_rand = run_ref(REF, blob, tmp)
            ppm = open(prefix + ".ppm", "rb").read()
            md5_ieee = hashlib.md5(ppm).hexdigest()
            header_len = len(f"P6\n{blob.width} {blob.height}\n255\n")
            rgb8 = np.frombuffer(ppm[header_len:], np.uint8).reshape(blob.height, blob.width, 3)
        with tempfile.TemporaryDirectory() as tmp:
            prefix, _, _ = run_ref(REF_FAST, blob, tmp)
            md5_fast = hashlib.md5(open(prefix + ".ppm", "rb").read()).hexdigest()
        np.savez_compressed(os.path.join(HERE, f"full_{name}.npz"), rgb8=rgb8, next_rand=next_rand)
        record[name] = dict(md5_ieee=md5_ieee, md5_fastmath=md5_fast, width=blob.width,
                            height=blob.height)
        print("full", name, record[name])
    json.dump(record, open(os.path.join(HERE, "full_md5.json"), "w"), indent=1)


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_harness is not built (needs /root/reference): "
                 "make -C oracle ref")
    make_rand()
    make_small()
    make_spp64()
    make_rays()
    if "--full" in sys.argv:
        make_full()
