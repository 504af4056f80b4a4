# usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pb -- python3 scratch/per_bounce.py SphereField 32 [opt=val ...]
# then: python3 scratch/per_bounce.py --report gpurun_out/pb gpurun_out/pb_rays.json
# One frame on ONE lane (no overlap): duration of every trace / shade launch next to the rays that entered the bounce.
import sys, json, glob, csv
sys.path.insert(0, "."); sys.path.insert(0, "tests")
if sys.argv[1] == "--report":
    rays = json.load(open(sys.argv[3]))
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    tr = [r for r in rows if "trace_ordered" in r["Kernel_Name"] or "trace_walk" in r["Kernel_Name"]]
    sh = [r for r in rows if "shade_kernel" in r["Kernel_Name"]]
    tl = [r for r in rows if "tail_kernel" in r["Kernel_Name"]]
    n = len(rays["rpb"])
    tr, sh = tr[-rays["launches_trace"]:], sh[-rays["launches_shade"]:]  # the last frame's launches (a warm-up frame precedes it)
    tot_t = tot_s = 0.0
    for b in range(len(tr)):
        dt = (int(tr[b]["End_Timestamp"]) - int(tr[b]["Start_Timestamp"])) / 1e3
        ds = (int(sh[b]["End_Timestamp"]) - int(sh[b]["Start_Timestamp"])) / 1e3
        tot_t += dt; tot_s += ds
        r = rays["rpb"][b]
        print(f"bounce {b:2d} rays {r:10d}  trace {dt:9.1f} us ({r / max(dt, 1e-9):8.0f} rays/us)  shade {ds:8.1f} us")
    print(f"sum trace {tot_t / 1e3:.2f} ms, shade {tot_s / 1e3:.2f} ms; tail launches {len(tl)}: " +
          ", ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f} us" for r in tl[-rays['launches_path']:] if rays['launches_path']))
    sys.exit(0)
import rust_pathtracer_amd as pt
which, spp = sys.argv[1], int(sys.argv[2])
if which == "mesh":
    s = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0).set_image(1920, 1080, spp, 50, (1, 1, 1)).build()
else:
    s = pt.Scene(which).set_seeds(1, 1, 1).set_image(1920, 1080, spp, 50, (1, 1, 1)).build()
ctx = pt.Context(0)
ctx.set_option("lanes", 1)
for a in sys.argv[3:]:
    k, v = a.split("="); ctx.set_option(k, int(v))
ctx.upload(s)
w, h = 1920, 1080
for rep in range(2):
    ctx.accum_reset(w, h)
    st = ctx.render_accumulate(s.camera(0.0), w, h, (1, 1, 1), 0, spp, 50, 1)
print("ms_total", st.ms_total, "trace", st.ms_trace, "shade", st.ms_shade, "path", st.ms_path)
json.dump({"rpb": [int(x) for x in st.rays_per_bounce][:50], "launches_trace": st.launches_trace, "launches_shade": st.launches_shade,
           "launches_path": st.launches_path}, open("gpurun_out/pb_rays.json", "w"))
