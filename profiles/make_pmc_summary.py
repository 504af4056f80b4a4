#!/usr/bin/env python3
"""pmc_summary.json (what bench.py scales to its launch size for roofline.traffic) from the raw sums of collect_pmc.sh.
usage: make_pmc_summary.py <raw.json from summarize_pmc.py> <rays of the profiled frame> > pmc_summary.json
FETCH_SIZE and WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md)."""
import json, sys
raw = json.load(open(sys.argv[1]))
rays = float(sys.argv[2])
out = {"_about": "HBM bytes per ray from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB), CornellBox 1920x1080 16 spp depth 50, lanes=1; "
                 "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md); see profiles/README.md",
       "workload_rays": rays}
for key, prefix in (("trace_kernel", "trace_walk_kernel"), ("shade_kernel", "shade_kernel<")):
    name = next(k for k in raw if k.startswith(prefix))
    f = raw[name]["FETCH_SIZE"]["sum"] * 1024 * 2
    w = raw[name]["WRITE_SIZE"]["sum"] * 1024
    out[key] = {"rocprof_name": name, "fetch_bytes": f, "write_bytes": w, "dispatches": raw[name]["FETCH_SIZE"]["dispatches"],
                "hbm_bytes_per_ray": round((f + w) / rays, 2),
                "valu_per_64_rays": round(raw[name]["SQ_INSTS_VALU"]["sum"] / (rays / 64), 1),
                "salu_per_64_rays": round(raw[name]["SQ_INSTS_SALU"]["sum"] / (rays / 64), 1),
                "l2_hit_rate": round(raw[name]["TCC_HIT_sum"]["sum"] / (raw[name]["TCC_HIT_sum"]["sum"] + raw[name]["TCC_MISS_sum"]["sum"]), 3)}
print(json.dumps(out, indent=1))
