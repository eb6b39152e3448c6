"""Condenses the counter passes of scripts/pmc_passes.sh for one kernel into the
JSON record bench.py reads (profiles/bench_pmc.json).

    python3 scripts/pmc_to_json.py <pmc dir> <kernel> <workload> <out.json>

Derived figures (per launch of the kernel, averaged over its dispatches):
  flops_executed_per_launch  (ADD + MUL + TRANS + 2 FMA) of F32 and F64 wave
                             instructions x 64 lanes x lane_utilisation
  lane_utilisation           SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)
  valu_busy                  4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles per XCD),
                             cycles per XCD = GRBM_GUI_ACTIVE / 8
  l2_hit_rate                TCC_HIT / (TCC_HIT + TCC_MISS)
  hbm_gbps                   bytes_per_launch / kernel time
  waves_per_simd             4 SQ_WAVE_CYCLES / (1024 SIMDs x cycles per XCD)
  bytes_per_launch           HBM traffic: 2 x FETCH_SIZE + WRITE_SIZE, in KB of 1024 B
                             (gfx950 tallies 128-byte fetches at 64 bytes)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root, kernel, workload, out = sys.argv[1:5]
tot = defaultdict(float)
disp = defaultdict(set)
for path in sorted(glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv"))):
    for row in csv.DictReader(open(path)):
        # "k_primary(...)" or the template form "void k_primary<5>(...)"
        name = row["Kernel_Name"].split("(")[0].split("<")[0].split()[-1]
        if name != kernel:
            continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
        disp[row["Counter_Name"]].add(row["Dispatch_Id"])
c = {k: v / len(disp[k]) for k, v in tot.items()}
launches = max(len(d) for d in disp.values())
rays = ms = None
for log in sorted(glob.glob(os.path.join(root, "*.log"))):
    m = re.search(r"rays=(\d+).* ms=([\d.]+) primary_ms=([\d.]+)", open(log).read())
    if m:
        rays, ms = int(m.group(1)), float(m.group(3))
f32 = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_TRANS_F32"] + 2 * c["SQ_INSTS_VALU_FMA_F32"]
f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_TRANS_F64"] + 2 * c["SQ_INSTS_VALU_FMA_F64"]
n64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_TRANS_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
lane = c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])
rec = {
    "workload": workload, "kernel": kernel, "launches_profiled": launches,
    "rays_per_launch": rays / launches if rays else None, "kernel_ms": ms / launches if ms else None,
    "flops_executed_per_launch": (f32 + f64) * 64 * lane,
    "flops_f32_share": f32 / (f32 + f64),
    "lane_utilisation": lane,
    "valu_busy": 4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8),
    "valu_insts_per_ray": c["SQ_INSTS_VALU"] * 64 * launches / rays if rays else None,
    "fp64_inst_share": n64 / c["SQ_INSTS_VALU"],
    "salu_per_valu": c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"],
    "wait_inst_share_of_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
    "bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "hbm_gbps": ((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / (ms / launches * 1e-3) / 1e9) if ms else None,
    # SQ_WAVE_CYCLES counts resident wave time in units of four cycles, like the other SQ cycle counters
    "waves_per_simd": 4 * c["SQ_WAVE_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8),
    "counters": c,
    "note": "rocprofv3 --pmc, one run per pass (scripts/pmc_passes.sh), program directly after --; "
            "valu_insts_per_ray counts lane-instructions (wave instructions x 64)",
}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in rec.items() if k != "counters"}, indent=1))
