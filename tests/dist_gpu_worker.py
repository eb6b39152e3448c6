"""Worker of test_gpu_dist.py: one of N ranks that share ONE GPU and talk over
gloo.  Renders its tiles with the HIP path, gathers the frame on rank 0 and
compares it bit for bit with a single-owner render of the same scene."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import raytracer_amd as ra                      # noqa: E402
from raytracer_amd import _abi, render          # noqa: E402
from raytracer_amd import dist as rdist         # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    scene_name, out = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = ra.load_sdl(os.path.join(ROOT, "assets", scene_name + ".sdl")).with_config(
        width=200, height=120, antialiasing=1, aa_samples=2)
    scene = render.DeviceScene(blob, device=0)
    base = render.make_params(rng_mode=_abi.RNG_KEYED, tile=(32, 32))
    frame = scene.new_frame()
    frame, st = rdist.render_frame(scene, base, rank, world, frame=frame)
    rays = torch.tensor([render.total_rays(st)], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        whole, _, st1 = scene.render(base)
        same = bool(torch.equal(whole, frame))
        with open(out, "w") as f:
            f.write(f"{int(same)} {int(rays.item())} {render.total_rays(st1)}\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
