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


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` as the driver invokes it (no launcher, no
    WORLD_SIZE): bench.py starts the ranks itself.  Two ranks share the one GPU
    here, so the exchange runs over gloo; the ray count of the frame must not
    depend on the number of ranks."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    bench = os.path.join(ROOT, "bench.py")
    out2 = subprocess.run([sys.executable, bench, "--gpus", "2", "--backend", "gloo", "--steps", "2",
                           "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=400)
    assert out2.returncode == 0, out2.stderr[-2000:]
    line2 = json.loads([l for l in out2.stdout.splitlines() if l.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and line2["steps"] == 2 and line2["value"] > 0
    out1 = subprocess.run([sys.executable, bench, "--steps", "1", "--warmup", "0", "--no-cpu"], env=env,
                          capture_output=True, text=True, timeout=400)
    assert out1.returncode == 0, out1.stderr[-2000:]
    line1 = json.loads([l for l in out1.stdout.splitlines() if l.startswith("{")][-1])
    assert line1["n_gpus"] == 1
    assert line1["config"]["rays_per_frame"] == line2["config"]["rays_per_frame"]
    assert "roofline" in line1
