# usage (GPU box): bash scratch/final_evidence.sh   -- instruction rates, kernel stats, ordered-walk phase and step statistics -> gpurun_out/ev/
set -e
export TMPDIR=/tmp
O=gpurun_out/ev; mkdir -p $O
timeout -k 10 120 ./scratch/ubench3 > $O/r02_instruction_rates_f16.txt
for c in 3 5; do
  extra=""; [ $c = 5 ] && extra="--frames 4"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$c -o ks -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline $extra > $O/ks$c.log 2>&1
  echo "kernel stats config $c done"
done
( export PT_LIBDIR=rust_pathtracer_amd/build/var_phases
  timeout -k 10 200 python -u scratch/phase_stats.py SphereField 2>&1 | grep -v amdgpu.ids
  timeout -k 10 200 python -u scratch/phase_stats.py mesh 2>&1 | grep -v amdgpu.ids ) | grep -v "wave-time" > $O/r02_ordered_walk_phases.txt
( export PT_LIBDIR=rust_pathtracer_amd/build/var_ptime
  timeout -k 10 200 python -u scratch/phase_stats.py SphereField 2>&1 | grep -E "rays|wave-time"
  timeout -k 10 200 python -u scratch/phase_stats.py mesh 2>&1 | grep -E "rays|wave-time" ) >> $O/r02_ordered_walk_phases.txt
timeout -k 10 300 python -u scratch/step_stats.py gen > /dev/null 2>&1
( export PT_LIBDIR=rust_pathtracer_amd/build/var_steps
  timeout -k 10 200 python -u scratch/step_stats.py count guide_split=0 2>&1 | grep -v amdgpu.ids
  timeout -k 10 200 python -u scratch/step_stats.py count guide_split=4 2>&1 | grep -v amdgpu.ids
  timeout -k 10 200 python -u scratch/step_stats.py count guide_split=4 guide_build=1 2>&1 | grep -v amdgpu.ids ) > $O/r02_step_histograms.txt
echo all done
