#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel and counter.
usage: summarize_pmc.py <dir with pass subdirs> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k][row["Counter_Name"]].add(row["Dispatch_Id"])
out = {}
for k in sorted(agg):
    out[k] = {c: {"sum": v, "dispatches": len(calls[k][c])} for c, v in sorted(agg[k].items())}
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:28s} {v:18.0f}  over {len(calls[k][c])} dispatches")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
