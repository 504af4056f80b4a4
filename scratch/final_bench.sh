# usage (GPU box): bash scratch/final_bench.sh <tag>   -- one bench.py record per BASELINE.json configuration -> gpurun_out/<tag>_b{1..5}.json
set -e
TAG=${1:-r02}
for c in 2 1 3 4 5; do
  timeout -k 10 400 python -u bench.py --config $c > gpurun_out/${TAG}_b$c.json 2> gpurun_out/${TAG}_b$c.err
  echo "config $c done: $(cut -c1-200 gpurun_out/${TAG}_b$c.json)"
done
