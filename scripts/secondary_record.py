"""Condenses the profile of the reflective benchmark scene (scripts/bounce_profile.sh) into
the record bench.py attaches as `secondary`: frame time, rays, the dominant bounce kernel
and its occupancy / lane figures from the counter passes.

    python3 scripts/secondary_record.py <bounce profile dir> <scene> <out.json>
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root, scene, out = sys.argv[1:4]
stats = list(csv.DictReader(open(os.path.join(root, f"kernel_stats_{scene}.csv"))))
log = open(os.path.join(root, f"trace_{scene}.log")).read()
m = re.search(r"(\d+)x(\d+) spp=(\d+) rays=(\d+).* ms=([\d.]+) primary_ms=([\d.]+)", log)
w, h, spp, rays, ms = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))
frames = 2   # bounce_profile.sh renders two frames under the tracer
kernels = sorted(((float(r["TotalDurationNs"]) / frames / 1e6, r["Name"].split("(")[0].replace("void ", "")) for r in stats),
                 reverse=True)
bounce = [k for k in kernels if "bounce" in k[1] or "sort" in k[1] or "hit_keys" in k[1]]
dominant = bounce[0][1]
base = dominant.split("<")[0]
tot = defaultdict(float)
for path in sorted(glob.glob(os.path.join(root, f"pmc_{scene}", "*", "*", "*_counter_collection.csv"))):
    for row in csv.DictReader(open(path)):
        if row["Kernel_Name"].split("(")[0].split("<")[0].split()[-1] == base:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
rec = {
    "workload": f"{scene}.sdl {w}x{h} {spp} spp, keyed samples", "rays_per_frame": rays, "ms_per_frame": ms,
    "value": rays / ms / 1e3, "unit": "Mrays/s",
    "kernels_ms_per_frame": {name: round(t, 3) for t, name in kernels[:8]},
    "dominant_bounce_kernel": dominant, "dominant_bounce_kernel_ms_per_frame": bounce[0][0],
}
if tot.get("SQ_ACTIVE_INST_VALU"):
    rec.update({
        "valu_busy": 4 * tot["SQ_ACTIVE_INST_VALU"] / (1024 * tot["GRBM_GUI_ACTIVE"] / 8),
        "lane_utilisation": tot["SQ_THREAD_CYCLES_VALU"] / (64 * tot["SQ_ACTIVE_INST_VALU"]),
        "waves_per_simd": 4 * tot["SQ_WAVE_CYCLES"] / (1024 * tot["GRBM_GUI_ACTIVE"] / 8),
        "wait_any_share_of_wave_cycles": tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"],
        "l2_hit_rate": tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]),
        "note": "counters summed over all dispatches of the kernel in one frame (rocprofv3 --pmc, one run per pass)",
    })
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
