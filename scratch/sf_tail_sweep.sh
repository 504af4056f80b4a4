# usage (GPU box): scratch/sf_tail_sweep.sh   -- SphereField 1080p 128 spp: the per-wave tail launch from bounce F on, dealt over 1/D of the waves
for rep in 1 2; do
a=$(timeout -k 10 100 python bench.py --scene SphereField --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])") || exit 1
echo "no tail: $a"
for f in 8 12 16 24; do
  for d in 2 4 8; do
    a=$(timeout -k 10 100 python bench.py --scene SphereField --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --opt tail_from=$f --opt tail_waves_div=$d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])") || exit 1
    echo "tail_from=$f div=$d: $a"
  done
done
done
