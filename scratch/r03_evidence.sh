# usage (GPU box): bash scratch/r03_evidence.sh <part>   -- round-3 records -> gpurun_out/ev3/
#   pmc C    PMC counters of configuration C (profiles/collect_pmc.sh)
#   stats    rocprofv3 --kernel-trace --stats of configurations 2, 3, 4, 5
#   bench    one bench.py line per configuration, and the two-worker rehearsal lines
#   phases   ordered-walk phase statistics (needs build/var_phases and build/var_ptime), device guide-build times
export TMPDIR=/tmp
O=gpurun_out/ev3; mkdir -p $O
case $1 in
pmc) bash profiles/collect_pmc.sh $2 r03 > $O/pmc_c$2.txt 2>&1; tail -3 $O/pmc_c$2.txt ;;
stats)
  for c in 2 3 4 5; do
    extra=""; [ $c = 5 ] && extra="--frames 4"; [ $c = 4 ] && extra="--steps 2"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$c -o ks -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline $extra > $O/ks$c.log 2>&1 || exit 1
    cp $(find $O/ks$c -name "*kernel_stats.csv" | head -1) $O/r03_kernel_stats_config$c.csv
    grep '^{' $O/ks$c.log > $O/r03_bench_under_rocprof_config$c.json
    echo "kernel stats config $c done"
  done ;;
bench)
  for c in 2 1 3 4 5; do
    timeout -k 10 500 python -u bench.py --config $c > $O/r03_bench_config$c.json 2> $O/b$c.err || exit 1
    echo "config $c: $(cut -c1-160 $O/r03_bench_config$c.json)"
  done
  timeout -k 10 300 python -u bench.py --config 2 --gpus 2 --devices 0,0 > $O/r03_rehearsal_2workers_config2.json 2> $O/reh2.err || exit 1
  timeout -k 10 300 python -u bench.py --config 5 --frames 8 --gpus 2 --devices 0,0 --steps 1 > $O/r03_rehearsal_2workers_config5.json 2> $O/reh5.err || exit 1
  echo rehearsals done ;;
phases)
  ( export PT_LIBDIR=rust_pathtracer_amd/build/var_phases
    timeout -k 10 200 python -u scratch/phase_stats.py SphereField 2>&1 | grep -v amdgpu.ids
    timeout -k 10 200 python -u scratch/phase_stats.py mesh 2>&1 | grep -v amdgpu.ids ) | grep -v "wave-time\|guide lane-steps" > $O/r03_ordered_walk_phases.txt
  ( export PT_LIBDIR=rust_pathtracer_amd/build/var_ptime
    timeout -k 10 200 python -u scratch/phase_stats.py SphereField 2>&1 | grep -E "rays|wave-time"
    timeout -k 10 200 python -u scratch/phase_stats.py mesh 2>&1 | grep -E "rays|wave-time" ) >> $O/r03_ordered_walk_phases.txt
  timeout -k 10 200 python -u scratch/guide_build_times.py > $O/r03_guide_build_times.txt 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gb -o gb -- python3 scratch/guide_build_kernels.py > $O/gb.log 2>&1
  cp $(find $O/gb -name "*kernel_stats.csv" | head -1) $O/r03_guide_build_kernel_stats.csv
  cat $O/r03_ordered_walk_phases.txt; tail -5 $O/r03_guide_build_times.txt; head -8 $O/r03_guide_build_kernel_stats.csv ;;
esac
