"""Benchmark of the render path: Mrays/s (primary + shadow + secondary rays) and
ms/frame on the 4K / 64 spp showcase frame, tile-partitioned over N GPUs.

    python bench.py --gpus N --steps K --warmup W

One "step" is one frame: every rank renders its screen tiles (cyclic
assignment) and rank 0 gathers the framebuffer (RCCL).  The frame is fixed, so
scaling is strong.  The scene (assets/showcase-4k.sdl: the reference's
showcase.sdl with the one cubemap the reference tree ships, 3840x2160,
antialiasing 8x8) is loaded and flattened into HBM before the timed region.
Samples come from the keyed generator (the scene has an area light, so the
reference's serial rand() stream cannot be replayed in parallel).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import raytracer_amd as ra                      # noqa: E402
from raytracer_amd import _abi, render          # noqa: E402
from raytracer_amd import dist as rdist         # noqa: E402

WORKLOAD = "showcase-4k"
# CPU sample: the same scene and camera at 1/40 of the resolution in each
# direction, same 64 samples per pixel (~5.4 M rays, ~10-30 s on one core).
CPU_SAMPLE = (96, 54)
PEAK_FP64_VECTOR_TFLOPS = 78.65   # half of the 157.3 TFLOP/s FP32 vector peak (MI355X, 256 CUs)


def algorithmic_flops(st):
    """Nominal flop count of the reference's own work for the rays it traced:
    30 per surface-box slab test, 46 per box-hit probe (point + direction
    transform), 60*ceil(log2(T/4)) + 416 per mesh probe (binary hierarchy over
    4-triangle leaves: two box tests per level, eight 52-flop triangle tests),
    60 per analytic-primitive probe, 90 per shaded light sample, 50 per
    environment lookup."""
    prim = st["surface_probes"] - st["mesh_probes"]
    return (30 * st["bbox_tests"] + 46 * st["surface_probes"] + 60 * st["mesh_probe_levels"]
            + 416 * st["mesh_probes"] + 60 * prim + 90 * st["light_samples"] + 50 * st["env_lookups"])


def measured_traffic(scene):
    """HBM bytes per k_primary launch on this workload (2 x FETCH_SIZE + WRITE_SIZE
    in KB, the gfx950 correction), from the committed counter passes; counters
    cannot be read from inside this process, so the figure is null when no profile
    of this workload is present."""
    t = measured_counters(scene)
    return t["bytes_per_launch"] if t else None


def cpu_baseline(blob):
    """Times the CPU renderer on a bounded sample of the workload (rank 0, N=1
    only).  Prefers the compiled reference (oracle/_ref, built in the build
    container from the reference's own sources); the repository's CPU
    restatement supplies the ray and flop counters either way."""
    w, h = CPU_SAMPLE
    sample = blob.with_config(width=w, height=h)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        bpath = os.path.join(tmp, "sample.blob")
        sample.save(bpath)
        cli = os.path.join(ROOT, "oracle", "oracle_cli")
        ref = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
        if not os.path.exists(cli):
            return None, None
        port = json.loads(subprocess.run([cli, "render", bpath, os.path.join(tmp, "o")],
                                         check=True, capture_output=True, text=True).stdout)
        desc = (f"{WORKLOAD} scene and camera at {w}x{h}, 64 spp, serial rand() stream, "
                f"{port['rays']} rays")
        out = {"value": port["mrays_per_s"], "unit": "Mrays/s", "cores": 1, "kind": "port",
               "sample": desc}
        # the restatement on all host cores this job may use (rows of a 2x larger sample
        # dealt out over threads, keyed samples: the serial stream cannot be split)
        cores = max(1, min(16, os.cpu_count() or 1))
        big = os.path.join(tmp, "sample2.blob")
        blob.with_config(width=2 * w, height=2 * h).save(big)
        try:
            allc = json.loads(subprocess.run([cli, "render", big, os.path.join(tmp, "a"), "--rng", "keyed",
                                              "--threads", str(cores)], check=True, capture_output=True,
                                             text=True, timeout=600).stdout)
            all_cores = {"value": allc["mrays_per_s"], "unit": "Mrays/s", "cores": cores,
                         "sample": f"{2 * w}x{2 * h}, 64 spp, keyed samples, {allc['rays']} rays"}
        except Exception:
            all_cores = None
        out["port_all_cores"] = all_cores
        if os.path.exists(ref):
            try:
                txt = subprocess.run([ref, "render", bpath, os.path.join(tmp, "r"), "--no-ppm"],
                                     check=True, capture_output=True, text=True, timeout=600).stdout
                secs = float(txt.split("render_seconds")[1].split()[0])
                out = {"value": port["rays"] / secs / 1e6, "unit": "Mrays/s", "cores": 1,
                       "kind": "reference", "sample": desc + "; reference sources compiled -O2 IEEE, "
                       "kd-tree counters zero-initialised by the compiler flag",
                       "port_value": port["mrays_per_s"], "port_all_cores": all_cores}
            except Exception:
                pass
    return out, port


def measured_counters(scene):
    """Hardware counters of k_primary on this workload, from the committed counter
    passes (profiles/bench_pmc.json, written by scripts/collect_profiles.sh on the
    GPU box for the same scene at the same size).  None when no profile matches."""
    path = os.path.join(ROOT, "profiles", "bench_pmc.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("workload") == scene:
            return t
    except (OSError, ValueError, KeyError):
        pass
    return None


def secondary_record():
    """The second hot kernel, measured on the GPU box at its own configuration
    (8-glossy 640x640 at 64 samples per pixel: the bounce pass) and committed with its
    counter passes; None when the record is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "secondary_8-glossy.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start one fresh process per
    rank (this parent has made no GPU call and makes none), wait for them, pass
    rank 0's JSON line through.  Ranks that die take the others down with them."""
    port = free_port()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + 3000
    pending = list(procs)
    while pending:
        for p in list(pending):
            r = p.poll()
            if r is not None:
                pending.remove(p)
                if r != 0:
                    rc = rc or r
        if rc or time.time() > deadline:
            for p in pending:
                p.terminate()
            for p in pending:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = rc or 1
            break
        time.sleep(0.05)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default=WORKLOAD)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo rehearses the N-rank path on fewer GPUs than ranks (ranks share devices)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    blob = ra.load_sdl(os.path.join(ROOT, "assets", args.scene + ".sdl"))
    scene = render.DeviceScene(blob, device=local_rank)
    base = render.make_params(rng_mode=_abi.RNG_KEYED, tile=(32, 32))
    frame = scene.new_frame()
    gather = rdist.FrameGather(frame, base, rank, world)
    # reductions of scalars: on the device over RCCL, on the host over gloo
    red_dev = scene.device if args.backend == "nccl" else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(want_stats=False):
        return rdist.render_frame(scene, base, rank, world, frame=frame, want_stats=want_stats, gather=gather)

    for _ in range(args.warmup):
        step()
    # one untimed, instrumented frame: ray counters and the primary kernel's
    # device time (HIP events on the launch stream, inside the library)
    _, st = step(want_stats=True)
    rays_local = render.total_rays(st)
    counts = torch.tensor([rays_local, st.rays_primary, st.rays_shadow, st.rays_secondary],
                          dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(counts)
    rays_frame = int(counts[0].item())

    # Timed region: frames without statistics, so no rank waits for its device before the
    # gather.  Every primary-kernel launch is bracketed by HIP events on the launch stream
    # inside the library (RTAMD_RENDER_TIME_KERNELS); they are read after the region.
    timed = render.make_params(rng_mode=_abi.RNG_KEYED, tile=(32, 32), flags=_abi.RENDER_TIME_KERNELS)
    scene.take_kernel_times()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rdist.render_frame(scene, timed, rank, world, frame=frame, want_stats=False, gather=gather)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms_sum, kernel_launches = scene.take_kernel_times()
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    ok = torch.tensor([1.0], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # every rank has rendered and gathered all its frames when it gets here
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    elapsed = float(t.item())
    ranks_ok = bool(ok.item() == 1.0)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_frame * args.steps / elapsed / 1e6
        result = {
            "metric": "Mrays/s (primary+secondary)", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "ranks_ok": ranks_ok,
            "config": {"workload": f"{args.scene}.sdl {blob.width}x{blob.height} {blob.spp} spp, "
                                   "keyed samples, 32x32 tiles dealt cyclically over ranks, "
                                   "framebuffer gather to rank 0",
                       "rays_per_frame": rays_frame, "ms_per_frame": ms_per_step,
                       "backend": args.backend if world > 1 else None},
        }
        cpu, port = (None, None)
        if world == 1 and not args.no_cpu:
            cpu, port = cpu_baseline(blob)
        kernel_ms = kernel_ms_sum / max(1, kernel_launches)
        launches_per_frame = max(1, st.launches_primary)
        rays_per_launch = rays_local / launches_per_frame
        pmc = measured_counters(args.scene) if world == 1 else None
        roof = {"bound": "valu", "kernel": "k_primary", "kernel_ms": kernel_ms, "unit": "TFLOP/s",
                "peak": PEAK_FP64_VECTOR_TFLOPS, "achieved": None, "frac": None,
                "traffic": measured_traffic(args.scene),
                "note": "vector-ALU bound: no dense contraction, scene data cache resident, HBM traffic "
                        "negligible.  achieved = floating-point operations the kernel EXECUTED per launch "
                        "(SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32/F64 x active lanes, FMA = 2; counter passes "
                        "of this workload in profiles/bench_pmc.json) / the kernel time measured here; "
                        "peak = 78.65 TFLOP/s, the vector FMA rate both FP64 and unpacked FP32 issue at.  "
                        "Most issued instructions are not flops (min/max, compares, conversions, integer "
                        "and mask work): valu_busy is the issue-slot occupancy that bounds the kernel."}
        if pmc and kernel_ms > 0:
            flops = pmc["flops_executed_per_launch"] * (rays_per_launch / pmc["rays_per_launch"])
            achieved = flops / (kernel_ms * 1e-3) / 1e12
            roof.update({
                "achieved": achieved, "frac": achieved / PEAK_FP64_VECTOR_TFLOPS,
                "frac_vs_fp32_peak": achieved / (2 * PEAK_FP64_VECTOR_TFLOPS),
                # (numerator and denominator come from two counter passes, i.e. two runs: a kernel
                # that issues every cycle can come out a fraction of a percent above one)
                "valu_busy": min(1.0, pmc["valu_busy"]), "valu_insts_per_ray": pmc["valu_insts_per_ray"],
                "fp64_inst_share": pmc["fp64_inst_share"], "lane_utilisation": pmc["lane_utilisation"],
                "salu_per_valu": pmc["salu_per_valu"], "counters_kernel_ms": pmc["kernel_ms"],
                # what the counter passes say about the memory side and the occupancy: hit rate of
                # the L2 (BVH nodes, triangles and scene tables are all served from it), HBM
                # bandwidth actually moved, waves resident per SIMD
                "l2_hit_rate": pmc.get("l2_hit_rate"), "hbm_gbps": pmc.get("hbm_gbps"),
                "waves_per_simd": pmc.get("waves_per_simd")})
        if port:
            # the survey's implementation-independent model of the REFERENCE's work for these
            # rays (not executed flops): kept for comparison with the survey's roof
            flops_per_ray = algorithmic_flops(port) / port["rays"]
            nominal = flops_per_ray * rays_per_launch / (kernel_ms * 1e-3) / 1e12
            roof.update({"nominal_reference_flops_per_ray": flops_per_ray,
                         "nominal_reference_tflops": nominal,
                         "nominal_frac_of_fp64_peak": nominal / PEAK_FP64_VECTOR_TFLOPS})
        result["roofline"] = roof
        sec = secondary_record()
        if sec:
            result["secondary"] = sec
        if cpu:
            result["cpu_baseline"] = cpu
            result["speedup_vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
