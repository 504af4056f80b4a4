# usage (GPU box): scratch/two_ctx_anim.sh   -- SphereField animation, 16 frames: one context, or two contexts on the one GPU (frames modulo 2, a host thread each)
for q in 4 8; do
export GPU_MAX_HW_QUEUES=$q
for l in 2 3 4; do
  a=$(timeout -k 10 200 python bench.py --config 5 --frames 16 --steps 1 --warmup 1 --no-cpu-baseline --lanes $l 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_frame'], d['value'])") || exit 1
  echo "queues=$q one context lanes=$l: $a"
  a=$(timeout -k 10 200 python bench.py --config 5 --frames 16 --steps 1 --warmup 1 --no-cpu-baseline --gpus 2 --devices 0,0 --lanes $l 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_frame'], d['value'])") || exit 1
  echo "queues=$q two contexts lanes=$l: $a"
done
done
