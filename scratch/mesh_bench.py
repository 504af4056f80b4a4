# ModelTest with a synthetic 1,000,000-triangle mesh (BASELINE.json configs[2]): trace modes compared on 1080p, a few spp
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
w, h, spp, depth = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 50
s = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0)
s.set_image(w, h, spp, depth, (1, 1, 1)).build()
cam = s.camera(0.0)
for mode in (2, 3):
    ctx = pt.Context(0)
    ctx.set_option("trace_mode", mode)
    ctx.set_option("lanes", 1)
    t0 = time.perf_counter()
    ctx.upload(s)
    print('upload s', round(time.perf_counter() - t0, 2))
    for rep in range(2):
        ctx.accum_reset(w, h)
        t0 = time.perf_counter()
        st = ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, spp, depth, 1)
        dt = time.perf_counter() - t0
    print("mode", mode, "rays", st.rays, "ms", round(dt * 1e3, 1), "Mrays/s", round(st.rays / dt / 1e6, 1), "trace ms", round(st.ms_trace, 1), "shade ms", round(st.ms_shade, 1))
    ctx.close()
