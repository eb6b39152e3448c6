"""Development helper (GPU box): renders every scene on the GPU at a reduced
size, compares hit records and colours with the CPU checker, then times the
full-size frames.  Writes gpurun_out/gpu_check.log."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import raytracer_amd as ra                     # noqa: E402
from raytracer_amd import _abi, render         # noqa: E402
import oracle_lib                              # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "gpu_check.log"), "w")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


SMALL = {"1-phong": (96, 96), "2-transform": (96, 96), "3-aa": (48, 48), "4-environmentmap": (64, 64),
         "5-primitives": (128, 64), "6-soft": (64, 32), "7-reflections": (40, 40), "8-glossy": (32, 32),
         "showcase-uffizi": (64, 36), "cornell": (64, 64)}


def compare(name, blob, mode):
    dev = render.DeviceScene(blob)
    orc = oracle_lib.OracleScene(blob)
    p = render.make_params(rng_mode=mode)
    frame, hits, st = dev.render(p, want_hits=True)
    g = frame.cpu().numpy()
    oc, oh, ost = orc.render(rng_mode=mode, want_hits=True, threads=8 if mode == 1 else 1)
    same = ((hits["surface"] == oh["surface"]) & (hits["triangle"] == oh["triangle"]) &
            (hits["t"].view(np.uint32) == oh["t"].view(np.uint32)))
    pos_same = (hits["position"].view(np.uint64) == oh["position"].view(np.uint64)).all(axis=1)
    nrm_same = (hits["normal"].view(np.uint64) == oh["normal"].view(np.uint64)).all(axis=1)
    q_g = ra.quantise_rgb8(g)
    q_o = ra.quantise_rgb8(oc)
    d8 = np.abs(q_g.astype(int) - q_o.astype(int))
    rays_g = render.total_rays(st)
    log(f"{name:18s} mode={mode} samples={len(hits)} id/t mismatch={int((~same).sum())} "
        f"pos mismatch={int((~pos_same).sum())} nrm mismatch={int((~nrm_same).sum())} "
        f"float maxabs={np.abs(g[..., :3] - oc[..., :3]).max():.3e} "
        f"rgb8 exact={100.0 * (d8 == 0).all(axis=2).mean():.3f}% within1={100.0 * (d8 <= 1).all(axis=2).mean():.3f}% "
        f"rays gpu={rays_g} oracle={ost.rays} ms={st.ms_total:.2f}")
    if (~same).any():
        idx = np.where(~same)[0][:5]
        for i in idx:
            log("   sample", i, "gpu", hits[i], "oracle", oh[i])
    dev.close()


def main():
    log("device", torch.cuda.get_device_name(0))
    for name, (w, h) in SMALL.items():
        blob = ra.load_sdl(os.path.join(ROOT, "assets", name + ".sdl"))
        kw = dict(width=w, height=h)
        if name == "showcase-uffizi":
            kw.update(antialiasing=1, aa_samples=4)
        blob = blob.with_config(**kw)
        mode = _abi.RNG_GLIBC if blob.fixed_rng_consumption() else _abi.RNG_KEYED
        try:
            compare(name, blob, mode)
            if mode == _abi.RNG_GLIBC:
                compare(name, blob, _abi.RNG_KEYED)
        except Exception as e:     # keep going: this is a survey of the state
            log(name, "ERROR", repr(e))
    # full-size timing
    for name in ["1-phong", "3-aa", "4-environmentmap", "6-soft", "7-reflections", "8-glossy",
                 "showcase-uffizi"]:
        blob = ra.load_sdl(os.path.join(ROOT, "assets", name + ".sdl"))
        if name == "showcase-uffizi":
            blob = blob.with_config(width=1920, height=1080, antialiasing=1, aa_samples=8)
        try:
            dev = render.DeviceScene(blob)
            p = render.make_params(rng_mode=_abi.RNG_KEYED)
            dev.render(p)
            t0 = time.time()
            frame, _, st = dev.render(p)
            torch.cuda.synchronize()
            dt = time.time() - t0
            rays = render.total_rays(st)
            log(f"FULL {name:18s} {blob.width}x{blob.height} spp={blob.spp} rays={rays} "
                f"ms={st.ms_total:.2f} wall={dt * 1e3:.2f} Mrays/s={rays / st.ms_total / 1e3:.1f} "
                f"primary_ms={st.ms_primary:.2f} launches={st.launches_primary}+{st.launches_bounce} "
                f"qpeak={st.queue_peak}")
            ra.ppm_write(frame.cpu().numpy(), os.path.join(ROOT, "gpurun_out", name + ".ppm"))
            dev.close()
        except Exception as e:
            log(name, "ERROR", repr(e))


if __name__ == "__main__":
    main()
