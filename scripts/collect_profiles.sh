#!/bin/bash
# Collects the profiles that back bench.py's numbers (run on the GPU box):
#   1. rocprofv3 --kernel-trace --stats of the bench command itself,
#   2. HBM traffic of the dominant kernel on the bench workload (FETCH_SIZE and
#      WRITE_SIZE in separate counter-only passes),
#   3. the instruction / busy / wait counters of k_primary on the 1080p frame.
# Usage: collect_profiles.sh <outdir>
set -u
export TMPDIR=/tmp
OUT=$(realpath -m "$1")
mkdir -p "$OUT"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 bench.py --steps 3 --warmup 1 --no-cpu > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" || { echo "kernel trace failed"; tail -5 "$OUT/bench_under_rocprof.err"; exit 1; }
find "$OUT/bench_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/bench_fetch" -- python3 bench.py --steps 1 --warmup 0 --no-cpu > "$OUT/bench_fetch.log" 2>&1 || { echo "fetch pass failed"; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/bench_write" -- python3 bench.py --steps 1 --warmup 0 --no-cpu > "$OUT/bench_write.log" 2>&1 || { echo "write pass failed"; exit 1; }
python3 scripts/pmc_summary.py "$OUT" k_primary > "$OUT/bench_traffic.txt" 2>&1
bash scripts/pmc_passes.sh "$OUT/pmc_1080p" --repeat 1 || exit 1
python3 scripts/pmc_summary.py "$OUT/pmc_1080p" k_ > "$OUT/pmc_k_primary_1080p.txt"
echo "profiles collected in $OUT"
