#!/bin/bash
# usage (GPU box): scratch/pmc_scene.sh <tag> <bench args...>   SQ counters of one bench frame, per kernel
set -u
export TMPDIR=/tmp
tag=$1; shift
OUT=gpurun_out/pmc_$tag
mkdir -p $OUT
run_pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 $BARGS > $OUT/$name.log 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
BARGS="$*"
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY || exit 1
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
python3 profiles/summarize_pmc.py $OUT $OUT/summary.json | grep -A17 "trace_"
