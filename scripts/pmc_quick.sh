#!/bin/bash
# Two PMC passes (instruction counts, busy/wait cycles) for quick A/B comparisons.
# Usage: pmc_quick.sh <outdir> [profile_run args]; RTAMD_LIB selects the library.
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
ARGS="$*"
pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 scripts/profile_run.py --repeat 1 $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
pass sq_time SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA &&
pass sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 &&
pass sq_mix SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32
python3 scripts/pmc_summary.py "$OUT" k_primary
