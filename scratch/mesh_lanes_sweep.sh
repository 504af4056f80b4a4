# usage (GPU box): scratch/mesh_lanes_sweep.sh   -- config 3 and config 5 (12 frames) by number of lanes (sub-passes on separate streams)
for l in 3 4 6 8; do
  for fg in 0 1; do
  a=$(timeout -k 10 100 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --lanes $l --opt lane_full_grid=$fg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_busy_ms_per_step'])") || exit 1
  echo "config3 lanes=$l full_grid=$fg: $a"
  done
done
for l in 3 4 6; do
  a=$(timeout -k 10 100 python bench.py --config 5 --frames 6 --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --lanes $l 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_frame'], d['kernel_busy_ms_per_step'])") || exit 1
  echo "config5 lanes=$l: $a"
done
