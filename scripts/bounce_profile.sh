#!/bin/bash
# Profiles the reflective scenes (the bounce path): rocprofv3 kernel trace + the
# counter passes of pmc_passes.sh, per scene.  Run on the GPU box.
# Usage: bounce_profile.sh <outdir> [tag]
set -u
export TMPDIR=/tmp
OUT=$(realpath -m "$1")
mkdir -p "$OUT"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for sc in "8-glossy 640 640" "7-reflections 640 640"; do
  set -- $sc
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$1" -- python3 scripts/profile_run.py --repeat 2 --scene $1 --width $2 --height $3 > "$OUT/trace_$1.log" 2>&1 || { echo "trace $1 failed"; tail -5 "$OUT/trace_$1.log"; exit 1; }
  find "$OUT/trace_$1" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$1.csv" \;
  bash scripts/pmc_passes.sh "$OUT/pmc_$1" --repeat 1 --scene $1 --width $2 --height $3 || exit 1
  python3 scripts/pmc_summary.py "$OUT/pmc_$1" k_ > "$OUT/pmc_$1.txt"
done
echo "bounce profiles collected in $OUT"
