#!/bin/bash
# Collects PMC counters for the kernels of one BASELINE.json configuration, one rocprofv3 pass per counter set (gfx950 slot
# limits: 8 SQ, 4 TCC with FETCH_SIZE = 3 and WRITE_SIZE = 2, 4 TCP, 2 TA, 2 TD), at a reduced sample count, one lane (no
# overlap between kernels), then folds them into profiles/pmc_summary.json under "config<C>".
# usage (GPU box, repo root): profiles/collect_pmc.sh <config 2..5> <round tag, e.g. r02>
set -u
export TMPDIR=/tmp
C=$1; TAG=${2:-r02}
case $C in
  2) ARGS="--config 2 --spp 16" ;;
  3) ARGS="--config 3 --spp 8" ;;
  4) ARGS="--config 4 --spp 16" ;;
  5) ARGS="--config 5 --spp 16 --frames 1 --no-rebuild" ;;
  *) echo "config 2..5"; exit 2 ;;
esac
OUT=gpurun_out/pmc_${TAG}_c$C
mkdir -p $OUT
BENCH="python3 bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --lanes 1"
run_pass() {
  name=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.log 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY || exit 1
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
run_pass tatd TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum || exit 1
python3 profiles/summarize_pmc.py $OUT $OUT/raw.json > $OUT/summary.txt || exit 1
echo "now, where the repository is writable (only gpurun_out/ comes back from the GPU box): python3 profiles/summarize_pmc.py $OUT profiles/${TAG}_pmc_raw_config$C.json; python3 profiles/make_pmc_summary.py $C profiles/${TAG}_pmc_raw_config$C.json $OUT/sq1.log \"$ARGS\""
