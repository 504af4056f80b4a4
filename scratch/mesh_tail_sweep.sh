# usage (GPU box): scratch/mesh_tail_sweep.sh   -- config 3 (1M triangles, 1080p 64 spp): tail launch from bounce F on, dealt over 1/D of the waves
for rep in 1 2; do
for f in 3 4 6; do
  for d in 1 2 3; do
    a=$(timeout -k 10 100 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --opt tail_from=$f --opt tail_waves_div=$d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])") || exit 1
    echo "tail_from=$f div=$d: $a"
  done
done
done
