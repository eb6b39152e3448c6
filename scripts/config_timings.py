"""GPU box: times every benchmark configuration at its stated size and writes a
Markdown table to gpurun_out/config_timings.md.

For each scene: the file's own settings, and where the configuration asks for
64 samples per pixel, the same scene with 8x8 supersampling.  The compiled
reference and the CPU restatement are timed beside the first configuration
(1-phong, the one the reference itself runs in seconds) on one host core."""
import json
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import raytracer_amd as ra                     # noqa: E402
from raytracer_amd import _abi, render         # noqa: E402

CONFIGS = [
    ("1-phong", {}),
    ("3-aa", {}),
    ("4-environmentmap", {}),
    ("4-environmentmap", dict(width=1024, height=1024, aa_samples=8)),
    ("7-reflections", {}),
    ("7-reflections", dict(aa_samples=8)),
    ("6-soft", {}),
    ("6-soft", dict(aa_samples=8)),
    ("8-glossy", {}),
    ("8-glossy", dict(aa_samples=8)),
    ("cornell", {}),
    ("cornell", dict(width=3840, height=2160, antialiasing=1, aa_samples=8)),
    ("showcase-4k", {}),
]


def cpu_seconds(blob):
    """(reference seconds, restatement seconds, rays) on one host core, or None."""
    cli = os.path.join(ROOT, "oracle", "oracle_cli")
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not (os.path.exists(cli) and os.path.exists(ref)):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        bpath = os.path.join(tmp, "s.blob")
        blob.save(bpath)
        port = json.loads(subprocess.run([cli, "render", bpath, os.path.join(tmp, "o")], check=True,
                                         capture_output=True, text=True).stdout)
        txt = subprocess.run([ref, "render", bpath, os.path.join(tmp, "r"), "--no-ppm"], check=True,
                             capture_output=True, text=True, timeout=600).stdout
        secs = float(txt.split("render_seconds")[1].split()[0])
    return secs, port["rays"] / port["mrays_per_s"] / 1e6, port["rays"]


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rows = ["| scene | size | spp | samples | rays / frame | ms / frame | Grays/s | primary ms | launches |",
            "|---|---|---|---|---|---|---|---|---|"]
    for name, kw in CONFIGS:
        blob = ra.load_sdl(os.path.join(ROOT, "assets", name + ".sdl"))
        if kw:
            blob = blob.with_config(**kw)
        modes = [("keyed", _abi.RNG_KEYED)]
        if blob.fixed_rng_consumption():
            modes.append(("rand()", _abi.RNG_GLIBC))
        dev = render.DeviceScene(blob)
        for label, mode in modes:
            p = render.make_params(rng_mode=mode)
            dev.render(p)
            best = None
            for _ in range(3):
                _, _, st = dev.render(p)
                if best is None or st.ms_total < best.ms_total:
                    best = st
            rays = render.total_rays(best)
            rows.append(f"| {name} | {blob.width}x{blob.height} | {blob.spp} | {label} | {rays} | "
                        f"{best.ms_total:.2f} | {rays / best.ms_total / 1e6:.2f} | {best.ms_primary:.2f} | "
                        f"{best.launches_primary}+{best.launches_bounce} |")
            print(rows[-1], flush=True)
        dev.close()
    torch.cuda.synchronize()
    cpu = cpu_seconds(ra.load_sdl(os.path.join(ROOT, "assets", "1-phong.sdl")))
    if cpu:
        rows.append("")
        rows.append(f"1-phong 512x512 on one host core: compiled reference {cpu[0]:.2f} s, CPU restatement "
                    f"{cpu[1]:.2f} s ({cpu[2]} rays).")
        print(rows[-1], flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "config_timings.md"), "w") as f:
        f.write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
