# usage (GPU box): scratch/opt_sweep.sh "<bench args>" "optset" ...   each optset = "--opt k=v --opt k=v" (or "" for defaults)
args=$1; shift
i=0
for o in "$@"; do
  i=$((i+1))
  timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline $args $o > gpurun_out/opt_$i.log 2>&1 || { tail -3 gpurun_out/opt_$i.log; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/opt_$i.log').read().strip().splitlines()[-1]); print('[$o]', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
