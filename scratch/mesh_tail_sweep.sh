# usage (GPU box): scratch/mesh_tail_sweep.sh   -- config 3 (1M triangles, 1080p 64 spp): tail launch from bounce F on
for f in 4 2 3 5 6; do
  a=$(timeout -k 10 100 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --opt tail_from=$f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_busy_ms_per_step'])") || exit 1
  echo "tail_from=$f: $a"
done
