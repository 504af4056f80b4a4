# usage (on the GPU box): scratch/run_variants.sh "<bench args>" variant...   ("" = the shipped lib)
args=$1; shift
for v in "$@"; do
  if [ "$v" = "base" ]; then unset PT_LIBDIR; else export PT_LIBDIR=$PWD/rust_pathtracer_amd/build/ablate_$v; fi
  timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > gpurun_out/var_$v.log 2>&1 || { tail -3 gpurun_out/var_$v.log; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/var_$v.log').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config']['rays_per_frame'])"
done
