"""The N>1 frame path with the real kernels: two ranks share the one GPU of the
test box and exchange their packed tiles over gloo (RCCL refuses two ranks on
one device); everything else - owner parameters, HIP pack / unpack kernels, the
gather - is the code the multi-GPU bench runs."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("scene", ["showcase-uffizi", "7-reflections"])
def test_two_ranks_on_one_gpu_rebuild_the_single_rank_frame(tmp_path, scene):
    out = str(tmp_path / "result.txt")
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"),
                                       scene, out], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    same, rays_split, rays_whole = open(out).read().split()
    assert same == "1"                       # bitwise the same image
    assert rays_split == rays_whole          # and the same rays, counted once each
