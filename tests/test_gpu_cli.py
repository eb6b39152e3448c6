"""The drop-in driver end to end: `raytracer scene.sdl` (the reference's
one-argument command line, raytracer.c:55-99) must leave in ./ray.ppm the very
file the reference writes - compared by the md5 of the compiled reference's
output (tests/golden/full_md5.json) - print the reference's console lines and
keep its exit codes."""
import hashlib
import json
import os
import shutil
import subprocess

import pytest

from conftest import ASSETS, GOLDEN, ROOT

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "raytracer_amd", "csrc", "raytracer")


@pytest.mark.parametrize("name", ["1-phong", "2-transform", "3-aa", "5-primitives"])
def test_cli_writes_the_reference_image(tmp_path, name):
    with open(os.path.join(GOLDEN, "full_md5.json")) as f:
        want = json.load(f)[name]["md5_ieee"]
    # mesh files are opened relative to the current directory, as in the reference
    # (scene.c:213-214): run where the reference is run, next to the .obj files
    for f in os.listdir(ASSETS):
        if f.endswith(".obj") or f == name + ".sdl":
            shutil.copy(os.path.join(ASSETS, f), str(tmp_path))
    r = subprocess.run([CLI, name + ".sdl"], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Rendering took" in r.stdout and "kilopixels per second" in r.stdout
    with open(os.path.join(str(tmp_path), "ray.ppm"), "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == want


def test_cli_error_paths(tmp_path):
    # no argument: exit code 1, nothing printed (raytracer.c:63-64)
    r = subprocess.run([CLI], cwd=str(tmp_path), capture_output=True, text=True, timeout=60)
    assert r.returncode == 1
    # unreadable scene: message + exit code 1 (raytracer.c:66-68)
    r = subprocess.run([CLI, "does-not-exist.sdl"], cwd=str(tmp_path), capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and r.stdout.strip() != ""
    assert not os.path.exists(os.path.join(str(tmp_path), "ray.ppm"))


def run_cli(tmp_path, *args):
    return subprocess.run([CLI, *args], cwd=str(tmp_path), capture_output=True, text=True, timeout=200)


def read_ppm(path):
    import numpy as np
    data = open(path, "rb").read()
    magic, dims, maxval, rest = data.split(b"\n", 3)
    w, h = (int(x) for x in dims.split())
    assert magic == b"P6" and maxval == b"255"
    return np.frombuffer(rest, np.uint8).reshape(h, w, 3)


def test_console_lines_are_the_references(tmp_path):
    """scene.c:570-573 (cubemap), :662-666 (one kd-tree per mesh, named, in the order
    the surface walk meets them), raytracer.c:13-30,83-92 (bar, timing lines)."""
    import re
    r = run_cli(tmp_path, os.path.join(ASSETS, "showcase-uffizi.sdl"), "--assets", ASSETS, "--width", "128",
                "--height", "96")
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.replace("\r", "\n").split("\n")
    text = [l for l in lines if l and not l.startswith("[")]
    took = r"took \d+ s \d{3} msec$"
    assert re.match(r"Loading cubemap " + took, text[0])
    # traversal order is reverse document order: the teapot is met before the tablecloth
    assert text[1] == "Building kd-tree for teapot" and re.match("Building kd-tree " + took, text[2])
    assert text[3] == "Building kd-tree for tablecloth" and re.match("Building kd-tree " + took, text[4])
    assert re.match("Rendering " + took, text[5])
    assert re.match(r"\d+\.\d\d kilopixels per second$", text[6])
    assert len(text) == 7
    bars = [l for l in lines if l.startswith("[")]
    assert bars and all(len(b) == 73 for b in bars)            # '[' + 70 + '>' ... + ']'
    assert bars[-1] == "[" + "=" * 70 + ">]"
    # without a cubemap there is no cubemap line; --stats adds the ray line at the end
    r = run_cli(tmp_path, os.path.join(ASSETS, "6-soft.sdl"), "--assets", ASSETS, "--width", "64", "--height", "32",
                "--stats")
    text = [l for l in r.stdout.replace("\r", "\n").split("\n") if l and not l.startswith("[")]
    assert text[0] == "Building kd-tree for teapot" and "Loading cubemap" not in r.stdout
    assert re.match(r"\d+ rays, ", text[-1])


def test_overrides_bands_and_devices(tmp_path):
    """--width/--height/--aa/--seed/-o change what they name; the number of bands and
    the list of devices (two "devices" that are the same GPU here) change nothing."""
    import numpy as np
    scene = os.path.join(ASSETS, "7-reflections.sdl")
    base = ["--assets", ASSETS, "--width", "96", "--height", "80", "--aa", "2"]
    r = run_cli(tmp_path, scene, *base, "-o", "a.ppm")
    assert r.returncode == 0, r.stdout + r.stderr
    assert not os.path.exists(os.path.join(str(tmp_path), "ray.ppm"))
    a = read_ppm(os.path.join(str(tmp_path), "a.ppm"))
    assert a.shape == (80, 96, 3) and a.max() > 50
    for extra, name in ((["--bands", "1"], "b.ppm"), (["--bands", "3"], "c.ppm"),
                        (["--devices", "0,0"], "d.ppm"), (["--devices", "0,0,0", "--bands", "2"], "e.ppm")):
        r = run_cli(tmp_path, scene, *base, *extra, "-o", name)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(read_ppm(os.path.join(str(tmp_path), name)), a), extra
    r = run_cli(tmp_path, scene, *base, "--seed", "12345", "-o", "s.ppm")
    s = read_ppm(os.path.join(str(tmp_path), "s.ppm"))
    assert s.shape == a.shape and not np.array_equal(s, a)
    assert np.abs(s.astype(int) - a.astype(int)).mean() < 3.0      # another sample set, the same picture
    r = run_cli(tmp_path, scene, *base[:-2], "--no-aa", "-o", "n.ppm")
    assert r.returncode == 0 and not np.array_equal(read_ppm(os.path.join(str(tmp_path), "n.ppm")), a)
    # a .png name selects the PNG writer: the same pixels as the PPM
    import struct
    import zlib
    r = run_cli(tmp_path, scene, *base, "-o", "a.png")
    assert r.returncode == 0, r.stdout + r.stderr
    data = open(os.path.join(str(tmp_path), "a.png"), "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and data[12:16] == b"IHDR"
    n = struct.unpack(">I", data[33:37])[0]
    assert data[37:41] == b"IDAT"
    rows = np.frombuffer(zlib.decompress(data[41:41 + n]), np.uint8).reshape(80, 1 + 3 * 96)
    assert np.array_equal(rows[:, 1:].reshape(80, 96, 3), a)
    assert run_cli(tmp_path, scene, "--frobnicate").returncode == 1
    assert run_cli(tmp_path, scene, *base, "--device", "99").returncode == 1


def test_progressive_viewer_fills_in_and_ends_on_the_batch_frame(tmp_path):
    """rayviewer (rayviewer.c:91-174 without the window): refreshes by interleaved tile
    subsets in a shuffled order; every refresh adds pixels, and the frame it leaves in
    ray.ppm is the batch driver's frame without antialiasing, whatever the pass count."""
    import numpy as np
    viewer = os.path.join(ROOT, "raytracer_amd", "csrc", "rayviewer")
    scene = os.path.join(ASSETS, "7-reflections.sdl")
    r = subprocess.run([CLI, scene, "--assets", ASSETS, "--no-aa", "--rng", "keyed", "-o", "batch.ppm"],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout
    batch = read_ppm(os.path.join(str(tmp_path), "batch.ppm"))
    for passes in (1, 7):
        r = subprocess.run([viewer, scene, "--assets", ASSETS, "--passes", str(passes), "--dump", f"v{passes}"],
                           cwd=str(tmp_path), capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stdout
        assert "Rendering took" in r.stdout and "kilopixels per second" in r.stdout
        assert np.array_equal(read_ppm(os.path.join(str(tmp_path), "ray.ppm")), batch)
        filled = [int((read_ppm(os.path.join(str(tmp_path), f"v{passes}-{i:03d}.ppm")).sum(axis=2) > 0).sum())
                  for i in range(passes)]
        assert all(b > a for a, b in zip(filled, filled[1:])) and filled[-1] > 0.9 * batch.shape[0] * batch.shape[1]


def test_viewer_refines_by_samples_and_ends_on_the_batch_frame(tmp_path):
    """rayviewer --progressive: every pass adds a subset of each pixel's samples; the first
    refresh already covers the whole picture, the error against the finished frame falls
    from refresh to refresh, and the frame left in ray.ppm is the batch driver's
    antialiased frame byte for byte (7-reflections: per-pixel sums are fixed point).  A
    window of that batch frame is compared with the CPU checker."""
    import numpy as np
    import raytracer_amd as ra
    import oracle_lib
    viewer = os.path.join(ROOT, "raytracer_amd", "csrc", "rayviewer")
    scene = os.path.join(ASSETS, "7-reflections.sdl")
    size = ["--width", "160", "--height", "120"]
    r = subprocess.run([CLI, scene, "--assets", ASSETS, "--aa", "4", "--rng", "keyed", "-o", "batch.ppm"] + size,
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout
    batch = read_ppm(os.path.join(str(tmp_path), "batch.ppm")).astype(int)
    # the batch frame itself against the checker, on a window
    blob = ra.load_sdl(scene).with_config(width=160, height=120, antialiasing=1, aa_samples=4)
    win = (60, 40, 100, 70)
    colour, _, _ = oracle_lib.OracleScene(blob).render(rng_mode=1, window=win, threads=8)
    want = ra.quantise_rgb8(colour).astype(int)[120 - win[3]:120 - win[1], win[0]:win[2]]
    got = batch[120 - win[3]:120 - win[1], win[0]:win[2]]
    assert (np.abs(got - want) <= 1).all(axis=2).mean() >= 0.999
    assert want.max() > 30
    # the viewer needs the same frame size: a scene file with that <Config>
    text = open(scene).read().replace('width =  "640"', 'width="160"').replace('height = "640"', 'height="120"')
    small = os.path.join(str(tmp_path), "small.sdl")
    with open(small, "w") as f:
        f.write(text)
    for passes in (1, 4, 16):
        r = subprocess.run([viewer, small, "--assets", ASSETS, "--progressive", str(passes), "--aa", "4",
                            "--dump", f"p{passes}"], cwd=str(tmp_path), capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stdout
        assert np.array_equal(read_ppm(os.path.join(str(tmp_path), "ray.ppm")).astype(int), batch)
        shots = [read_ppm(os.path.join(str(tmp_path), f"p{passes}-{i:03d}.ppm")).astype(int) for i in range(passes)]
        # every refresh is a whole picture (the floor and the teapot are lit everywhere they are in the end)
        lit = batch.sum(axis=2) > 24
        for s in shots:
            assert (s.sum(axis=2)[lit] > 0).mean() > 0.99
        if passes > 1:
            err = [np.abs(s - shots[-1]).mean() for s in shots]
            assert err[0] > err[len(err) // 2] > err[-1] == 0
