# usage (GPU box): bash scratch/ab_bench.sh "<bench options>" base v1 ...   -- one bench.py line (value, ms_per_step) per library variant
o=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset PT_LIBDIR; else export PT_LIBDIR=$PWD/rust_pathtracer_amd/build/var_$v; fi
  r=$(timeout -k 10 200 python bench.py $o --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_busy_ms_per_step'])") || exit 1
  echo "$v [$o]: $r"
done
