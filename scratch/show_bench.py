"""Prints the headline numbers of bench.py JSON lines: python scratch/show_bench.py gpurun_out/x/*.json"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e)
        continue
    print(f, d["value"], "Mrays/s", d["ms_per_step"], "ms/step", d.get("ms_per_frame"), "tail_from", d["roofline"].get("tail_from"))
    for k in d["roofline"]["kernels"]:
        print("   ", k["kernel"], k["launches"], k.get("rays"), "sum", k["launch_ms_sum"], "avg", k["avg_launch_ms"], "frac", k.get("frac"))
    print("    busy", d["kernel_busy_ms_per_step"])
