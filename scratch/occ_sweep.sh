# usage (GPU box): scratch/occ_sweep.sh "<bench args>" [lanes...]   -- lanes x persistent blocks per CU, ms per frame and Mrays/s
args=$1; shift
lanes=${@:-3 4}
for rep in 1 2; do
  for l in $lanes; do
    for b in 6 8 12; do
      a=$(timeout -k 10 100 python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --lanes $l --opt blocks_per_cu=$b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])") || exit 1
      echo "lanes=$l blocks_per_cu=$b: $a"
    done
  done
done
