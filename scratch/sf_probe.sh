# usage (GPU box): bash scratch/sf_probe.sh "<bench options>" ...   -- SphereField frames (config 5, 8 frames) under sets of bench.py options
set -e
for o in "$@"; do
  echo "== $o"
  timeout -k 10 200 python -u bench.py --config 5 --frames 8 --steps 1 --warmup 1 --no-cpu-baseline $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], {k:d.get(k) for k in ('kernel_busy_ms_per_step',)})"
done
