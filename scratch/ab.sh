# usage (GPU box): scratch/ab.sh variant...   ("base" = the shipped library; others: rust_pathtracer_amd/build/var_<name>, scratch/build_variant.sh)
# SphereField 1080p 32 spp and the 1M-triangle mesh 1080p 16 spp per variant: ms per frame, busy ms of trace and of the tail launch
for v in "$@"; do
  if [ "$v" = base ]; then unset PT_LIBDIR; else export PT_LIBDIR=$PWD/rust_pathtracer_amd/build/var_$v; fi
  a=$(timeout -k 10 100 python bench.py --scene SphereField --spp 32 --steps 5 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_busy_ms_per_step']['trace'])") || exit 1
  b=$(timeout -k 10 100 python bench.py --config 3 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_busy_ms_per_step']['trace'], d['kernel_busy_ms_per_step']['path'])") || exit 1
  echo "$v  spherefield32: $a   mesh16: $b"
done
