#!/bin/bash
# Collects the profiles that back bench.py's numbers (run on the GPU box):
#   1. rocprofv3 --kernel-trace --stats of the bench command itself,
#   2. the counter passes (instruction mix, busy / wait cycles, HBM traffic) of
#      k_primary on the bench workload itself (4K, 64 spp), condensed into
#      bench_pmc.json, the record bench.py reads.
# Usage: collect_profiles.sh <outdir>
set -u
export TMPDIR=/tmp
OUT=$(realpath -m "$1")
mkdir -p "$OUT"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 bench.py --steps 3 --warmup 1 --no-cpu > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" || { echo "kernel trace failed"; tail -5 "$OUT/bench_under_rocprof.err"; exit 1; }
find "$OUT/bench_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
bash scripts/pmc_passes.sh "$OUT/pmc_4k" --repeat 1 --scene showcase-4k --aa 0 --width 3840 --height 2160 || exit 1
python3 scripts/pmc_summary.py "$OUT/pmc_4k" k_ > "$OUT/pmc_k_primary_4k.txt"
python3 scripts/pmc_to_json.py "$OUT/pmc_4k" k_primary showcase-4k "$OUT/bench_pmc.json" > "$OUT/bench_pmc_summary.txt" || exit 1
echo "profiles collected in $OUT"
