# usage (GPU box): python scratch/split_probe.py "guide_split=0" "guide_split=4,tail_deal=0" ...   -- the mesh frame under sets of options
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
m = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0).set_image(1920, 1080, 16, 50, (1, 1, 1)).build()
cam = m.camera(0.0)
w, h = 1920, 1080
ref = None
for opts in sys.argv[1:]:
    ctx = pt.Context(0)
    for kv in opts.split(","):
        if kv:
            k, v = kv.split("="); ctx.set_option(k, int(v))
    t0 = time.perf_counter(); ctx.upload(m); up = time.perf_counter() - t0
    for spp in (1, 16, 64):
        best = None
        for k in range(3):
            ctx.accum_reset(w, h)
            st = ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, spp, 50, 1)
            if best is None or st.ms_total < best.ms_total: best = st
        st = best
        print(f"[{opts}] (upload {up*1e3:.0f} ms) spp {spp}: total {st.ms_total:.2f} trace busy {st.ms_busy_trace:.2f} shade busy {st.ms_busy_shade:.2f} tail busy {st.ms_busy_path:.2f} repack {st.ms_repack:.2f} rays {st.rays}", flush=True)
    img = ctx.accum_download()
    if ref is None: ref = img
    else: print("   identical to the first:", bool((img == ref).all()), flush=True)
    ctx.close()
