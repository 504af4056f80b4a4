# usage (GPU box): scratch/ab_ms.sh "<bench args>" variant...   ms per step only, three alternating rounds ("base" = shipped library)
args=$1; shift
for rep in 1 2 3; do
for v in "$@"; do
  if [ "$v" = base ]; then unset PT_LIBDIR; else export PT_LIBDIR=$PWD/rust_pathtracer_amd/build/var_$v; fi
  a=$(timeout -k 10 100 python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])") || exit 1
  echo "$v: $a"
done
done
