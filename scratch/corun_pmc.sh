# usage (GPU box): scratch/corun_pmc.sh <config> [bench args]  -- SQ counters of a frame with the default three overlapped lanes, summed over all dispatches
export TMPDIR=/tmp
C=$1; shift
O=gpurun_out/corun_c$C; rm -rf $O; mkdir -p $O
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- python3 bench.py --config $C --steps 1 --warmup 0 --no-cpu-baseline --no-isolated "$@" > $O/sq.log 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/grbm -- python3 bench.py --config $C --steps 1 --warmup 0 --no-cpu-baseline --no-isolated "$@" > $O/grbm.log 2>&1 || exit 1
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
tot = collections.defaultdict(float); per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(o + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); per[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()): print(f"{k:24s} {v:.4g}")
cyc = tot["GRBM_GUI_ACTIVE"] / 8.0
print("GPU-active cycles (per XCD average)", f"{cyc:.4g}")
print("VALU busy (SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * cycles))", round(tot["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cycles_sum if (cycles_sum := cyc) else 1), 3))
print("mean waves resident per SIMD (SQ_WAVE_CYCLES / (1024 * cycles))", round(tot["SQ_WAVE_CYCLES"] / (1024 * cyc), 2))
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:6]:
    print(f"  {k[:40]:40s} VALU {v.get('SQ_INSTS_VALU', 0):.4g} SALU {v.get('SQ_INSTS_SALU', 0):.4g} wave-cycles {v.get('SQ_WAVE_CYCLES', 0):.4g} waves {v.get('SQ_WAVES', 0):.4g}")
PY
