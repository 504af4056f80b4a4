#!/bin/bash
# usage (GPU box): scratch/pmc_mem.sh <tag> <bench args...>   TA / TCP / TD counters of one bench frame, per kernel
set -u
export TMPDIR=/tmp
tag=$1; shift
OUT=gpurun_out/pmcmem_$tag
mkdir -p $OUT
BARGS="$*"
run_pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 $BARGS > $OUT/$name.log 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run_pass ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE || exit 1
run_pass tcp TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum || exit 1
run_pass td TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum || exit 1
python3 profiles/summarize_pmc.py $OUT $OUT/summary.json | grep -A12 "trace_\|^shade"
