import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")
ASSETS = os.path.join(ROOT, "assets")
REF_HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
REFERENCE_DIR = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer-running CPU check")


def scene_path(name):
    if name.startswith("prims"):
        return os.path.join(ROOT, "tests", "scenes", "prims.sdl")
    return os.path.join(ASSETS, name + ".sdl")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def small_blob(name):
    """The scene at the reduced <Config> its small fixture was generated with."""
    import raytracer_amd as ra
    g = load_golden("small_" + name)
    cfg = json.loads(str(g["config"]))
    return ra.load_sdl(scene_path(name)).with_config(**cfg)


SMALL_SCENES = ["1-phong", "2-transform", "3-aa", "4-environmentmap", "5-primitives", "6-soft",
                "7-reflections", "8-glossy", "showcase-uffizi", "cornell", "prims", "prims-aa"]
FIXED_SCENES = ["1-phong", "2-transform", "3-aa", "4-environmentmap", "5-primitives"]


@pytest.fixture(scope="session")
def have_reference_harness():
    return os.path.isdir(REFERENCE_DIR) and os.path.exists(REF_HARNESS)
