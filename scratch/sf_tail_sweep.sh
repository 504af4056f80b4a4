# usage (GPU box): scratch/sf_tail_sweep.sh   -- SphereField 1080p 32 spp: the per-wave tail launch from bounce F on, dealt over 1/D of the waves
for f in 50 6 10 16; do
  for d in 1 4 16; do
    a=$(timeout -k 10 100 python bench.py --scene SphereField --spp 32 --steps 5 --warmup 1 --no-cpu-baseline --no-isolated --opt tail_from=$f --opt tail_waves_div=$d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_busy_ms_per_step'])") || exit 1
    echo "tail_from=$f div=$d: $a"
    if [ $f = 50 ]; then break; fi
  done
done
