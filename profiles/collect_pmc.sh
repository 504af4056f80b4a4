#!/bin/bash
# Collects PMC counters for the wavefront kernels, one rocprofv3 pass per counter set (gfx950 slot limits:
# 8 SQ, 4 TCC with FETCH_SIZE = 3 and WRITE_SIZE = 2).  Run on the GPU box from the repo root.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/pmc_${1:-r01}
mkdir -p $OUT
BENCH="python3 bench.py --spp 16 --steps 1 --warmup 0 --no-cpu-baseline --lanes 1"
run_pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.log 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY || exit 1
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
