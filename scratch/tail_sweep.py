# frame time against tail_from (the first bounce of the per-wave tail launch) on the larger scenes
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
name = sys.argv[1]; spp = int(sys.argv[2]); lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w, h, depth = 1920, 1080, 50
s = pt.Scene(name).set_seeds(1, 1, 1)
if name == "ModelTest":
    s.set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0)
s.set_image(w, h, spp, depth, (1, 1, 1)).build()
cam = s.camera(0.0)
ctx = pt.Context(0)
ctx.set_option("lanes", lanes)
ctx.set_option("profile", 1)
for tail in [int(x) for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else "4096,0,1,2,3,4,6,8,12".split(","))]:
    ctx.set_option("tail_from", tail)
    ctx.upload(s)
    best = None
    for rep in range(3):
        ctx.accum_reset(w, h)
        t0 = time.perf_counter()
        st = ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, spp, depth, 1)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]: best = (dt, st)
    dt, st = best
    print(name, "lanes", lanes, "tail_from", tail, "rays", st.rays, "ms %.1f" % (dt * 1e3), "Mrays/s %.0f" % (st.rays / dt / 1e6),
          "trace %.1f shade %.1f tail %.1f" % (st.ms_trace, st.ms_shade, st.ms_path), flush=True)
