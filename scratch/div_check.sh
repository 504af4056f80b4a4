# usage (GPU box): scratch/div_check.sh -- config 3 at 1, 8, 64 spp: tail dealt over all waves (div 1) or half of them (div 2)
for spp in 1 8 64; do
for rep in 1 2; do
  for d in 1 2; do
    a=$(timeout -k 10 100 python bench.py --config 3 --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --opt tail_waves_div=$d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])") || exit 1
    echo "spp=$spp div=$d: $a"
  done
done
done
