# usage (GPU box): scratch/bpc_sweep.sh  -- persistent-grid size (blocks of 4 waves per CU) for the ordered-walk scenes
for bpc in 6 5 8 10 12; do
  a=$(timeout -k 10 100 python bench.py --scene SphereField --spp 32 --steps 5 --warmup 1 --no-cpu-baseline --no-isolated --blocks-per-cu $bpc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_busy_ms_per_step']['trace'])") || exit 1
  b=$(timeout -k 10 100 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --blocks-per-cu $bpc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_busy_ms_per_step']['trace'], d['kernel_busy_ms_per_step']['path'])") || exit 1
  echo "bpc=$bpc  spherefield32: $a   mesh64: $b"
done
