# usage (GPU box): scratch/ab_ordered.sh   -- SphereField 32 spp and the 1M mesh at 16 spp for each library variant x blocks_per_cu
for v in base ow5 ow6 ow8; do
  for bpc in 0 4 5 6 8; do
    if [ $v = base ]; then export PT_LIBDIR=; unset PT_LIBDIR; else export PT_LIBDIR=$PWD/rust_pathtracer_amd/build/var_$v; fi
    a=$(timeout -k 10 100 python bench.py --scene SphereField --spp 32 --steps 5 --warmup 1 --no-cpu-baseline --no-isolated --blocks-per-cu $bpc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_busy_ms_per_step']['trace'])") || exit 1
    b=$(timeout -k 10 100 python bench.py --config 3 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --blocks-per-cu $bpc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_busy_ms_per_step']['trace'], d['kernel_busy_ms_per_step']['path'])") || exit 1
    echo "$v bpc=$bpc  spherefield32: $a   mesh16: $b"
  done
done
