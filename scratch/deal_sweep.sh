# usage (GPU box): scratch/deal_sweep.sh  -- option deal_from (survivors dealt out over the waves before every trace launch from that bounce on)
for f in -1 1 2 3; do
  b=$(timeout -k 10 100 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --opt deal_from=$f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['kernel_busy_ms_per_step']['trace'], d['kernel_busy_ms_per_step']['path'])") || exit 1
  echo "mesh64 deal_from=$f: $b"
done
for f in -1 2 4 6; do
  a=$(timeout -k 10 100 python bench.py --scene SphereField --spp 32 --steps 5 --warmup 1 --no-cpu-baseline --no-isolated --opt deal_from=$f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['kernel_busy_ms_per_step']['trace'])") || exit 1
  echo "spherefield32 deal_from=$f: $a"
done
