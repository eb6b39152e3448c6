"""Times what one rank of an N-GPU run does per frame (its tiles + the pack kernel),
on one GPU: an estimate of strong-scaling efficiency without the gather."""
import ctypes as C
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra
from raytracer_amd import _abi, render
from raytracer_amd import dist as rdist

blob = ra.load_sdl(os.path.join(ROOT, "assets", "showcase-4k.sdl"))
scene = render.DeviceScene(blob)
base = render.make_params(rng_mode=_abi.RNG_KEYED, tile=(32, 32))
frame = scene.new_frame()
packed = torch.zeros_like(frame)
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in range(world):
        p = rdist.owner_params(base, rank, world)
        scene.render(p, frame=frame)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            _, _, st = scene.render(p, frame=frame, want_stats=True)
            if world > 1:
                _abi.check(_abi.lib().rtamd_tiles_pack(blob.width, blob.height, C.byref(p), frame.data_ptr(),
                                                       packed.data_ptr(), None), "pack")
        torch.cuda.synchronize()
        worst = max(worst, (time.perf_counter() - t0) / 3 * 1e3)
    print(f"world={world} slowest rank {worst:.2f} ms/frame")
