# usage (GPU box, under rocprofv3 --kernel-trace): one 1-spp mesh frame with every bounce as its own trace + shade launch
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
m = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0).set_image(1920, 1080, 16, 50, (1, 1, 1)).build()
ctx = pt.Context(0)
for k, v in [a.split("=") for a in sys.argv[1:]]:
    ctx.set_option(k, int(v))
ctx.upload(m)
cam = m.camera(0.0)
w, h = 1920, 1080
ctx.set_option("lanes", 1)
ctx.set_option("tail_from", 50)
ctx.accum_reset(w, h)
st = ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, 1, 50, 1)
print("total", st.ms_total, "trace", st.ms_trace, "shade", st.ms_shade)
