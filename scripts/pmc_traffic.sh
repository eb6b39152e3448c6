#!/bin/bash
# Memory-side counter passes for quick A/B comparisons: vector memory instructions and the
# bytes the L2 exchanges with memory.  Usage: pmc_traffic.sh <outdir> [profile_run args];
# RTAMD_LIB selects the library.
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
ARGS="$*"
pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 scripts/profile_run.py --repeat 1 $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
pass sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 &&
pass tcc_fetch FETCH_SIZE TCC_HIT_sum &&
pass tcc_write WRITE_SIZE TCC_MISS_sum TCC_ATOMIC_sum
python3 scripts/pmc_summary.py "$OUT" k_primary | grep -E "==|VMEM|FETCH|WRITE|TCC|INSTS_VALU "
