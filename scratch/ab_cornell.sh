# usage (GPU box): scratch/ab_cornell.sh variant...   Cornell box 1080p 256 spp (config 2), alternating variants twice ("base" = shipped lib)
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset PT_LIBDIR; else export PT_LIBDIR=$PWD/rust_pathtracer_amd/build/var_$v; fi
  a=$(timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k={e['kernel']:e['avg_launch_ms'] for e in d['roofline']['kernels']}; i={e['kernel']:e['avg_launch_ms'] for e in d['roofline']['isolated']['kernels']}
print(d['ms_per_step'], d['value'], 'co-run', k, 'isolated', i)") || exit 1
  echo "$v: $a"
done
done
