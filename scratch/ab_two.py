# usage (GPU box): [PT_LIBDIR=...] python scratch/ab_two.py [key=value ...]  -- SphereField 1080p 32 spp and the 1M mesh 1080p 16 spp, best of a few frames
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
def run(scene, spp, reps):
    ctx = pt.Context(0)
    for a in sys.argv[1:]:
        k, v = a.split("="); ctx.set_option(k, int(v))
    ctx.upload(scene)
    cam = scene.camera(0.0)
    best = None
    for k in range(reps):
        ctx.accum_reset(1920, 1080)
        st = ctx.render_accumulate(cam, 1920, 1080, (1, 1, 1), 0, spp, 50, 1)
        if best is None or st.ms_total < best.ms_total: best = st
    ctx.close()
    return best
a = run(pt.Scene("SphereField").set_seeds(1, 1, 1).set_image(1920, 1080, 32, 50, (1, 1, 1)).build(), 32, 5)
b = run(pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0).set_image(1920, 1080, 16, 50, (1, 1, 1)).build(), 16, 4)
print(f"spherefield32 {a.ms_total:.2f} (trace busy {a.ms_busy_trace:.2f})   mesh16 {b.ms_total:.2f} (trace busy {b.ms_busy_trace:.2f}, tail {b.ms_busy_path:.2f})", flush=True)
