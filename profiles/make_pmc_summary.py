#!/usr/bin/env python3
"""Folds the raw per-kernel counter sums of collect_pmc.sh into profiles/pmc_summary.json under "config<C>": what bench.py scales
to its launch size for roofline.traffic and uses for the VALU-issue roofline.
usage: make_pmc_summary.py <config> <raw.json from summarize_pmc.py> <log of one pass (holds bench.py's JSON line)> "<bench args>"
FETCH_SIZE and WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md)."""
import json, os, sys

cfg, raw_path, log_path, args = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
raw = json.load(open(raw_path))
line = next(l for l in open(log_path) if l.startswith("{") and '"rays_per_frame"' in l)
bench = json.loads(line)
rays = float(bench["config"]["rays_per_frame"])
# rays each kernel traced in that run (round 3 on: bench.py splits rays_per_bounce by PtStats.tail_from)
rays_of = {e["kernel"]: float(e["rays"]) for e in (bench.get("roofline") or {}).get("kernels", []) if "rays" in e}
here = os.path.dirname(os.path.abspath(__file__))
path = os.path.join(here, "pmc_summary.json")
doc = json.load(open(path)) if os.path.exists(path) else {}
if "trace_kernel" in doc:  # round 1's flat layout was config 2
    doc = {"config2": {k: v for k, v in doc.items() if not k.startswith("_") and k != "workload_rays"}}
doc["_about"] = ("per BASELINE.json configuration: HBM bytes per ray from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB; FETCH_SIZE doubled: "
                 "gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md), wave instructions per 64 rays, L2 hit rate, texture "
                 "addresser / data-return busy fractions; collected by profiles/collect_pmc.sh at a reduced sample count with lanes=1")
out = {"_source": f"PMC, bench.py {args} --lanes 1 ({os.path.basename(raw_path)})", "workload_rays": rays}
gui = None
for key, prefixes in (("trace_kernel", ("trace_walk_kernel", "trace_ordered_kernel", "trace_stream_kernel")), ("shade_kernel", ("shade_kernel<",)),
                      ("path_kernel", ("tail_kernel<", "path_kernel<"))):
    names = [k for k in raw if k.startswith(prefixes)]
    if not names:
        continue
    name = max(names, key=lambda k: raw[k].get("SQ_INSTS_VALU", {"sum": 0})["sum"])
    r = raw[name]
    g = lambda c: r[c]["sum"] if c in r else None
    f, w = g("FETCH_SIZE") * 1024 * 2, g("WRITE_SIZE") * 1024
    e = {"rocprof_name": name, "fetch_bytes": f, "write_bytes": w, "dispatches": r["FETCH_SIZE"]["dispatches"]}
    kr = rays_of.get(key, rays if key != "path_kernel" else None)  # (before round 3 a tail launch's rays were not split out)
    if kr:
        e.update({"rays": kr, "hbm_bytes_per_ray": round((f + w) / kr, 2), "valu_per_64_rays": round(g("SQ_INSTS_VALU") / (kr / 64), 1),
                  "salu_per_64_rays": round(g("SQ_INSTS_SALU") / (kr / 64), 1)})
    e["l2_hit_rate"] = round(g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum")), 3)
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        e["valu_lane_utilisation"] = round(g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0), 3)
    if g("GRBM_GUI_ACTIVE"):
        cycles = g("GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs
        e["kernel_cycles"] = cycles
        if g("SQ_ACTIVE_INST_VALU"):
            e["valu_busy"] = round(g("SQ_ACTIVE_INST_VALU") * 4.0 / (1024.0 * cycles), 3)  # quad-cycles over 1024 SIMDs
        if g("TA_TA_BUSY_sum"):
            e["ta_busy"] = round(g("TA_TA_BUSY_sum") / (256.0 * cycles), 3)
        if g("TD_TD_BUSY_sum"):
            e["td_busy"] = round(g("TD_TD_BUSY_sum") / (256.0 * cycles), 3)
    out[key] = e
doc[f"config{cfg}"] = out
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
