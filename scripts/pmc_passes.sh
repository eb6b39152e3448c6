#!/bin/bash
# Runs the PMC counter passes for the render kernels, one rocprofv3 run per pass
# (counters only: no trace domains in the same run).  Usage: pmc_passes.sh <outdir> [profile_run args]
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 scripts/profile_run.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
ARGS="$*"
pass sq_time SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU &&
pass sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 &&
pass sq_mix SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 &&
pass sq_mem SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH &&
pass tcc_fetch FETCH_SIZE TCC_HIT_sum &&
pass tcc_write WRITE_SIZE TCC_MISS_sum TCC_ATOMIC_sum &&
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo "pmc passes done"
