# one frame of the 1M-triangle ModelTest at 8 spp with the default trace mode (for rocprofv3 --kernel-trace)
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
w, h, spp, depth = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 50
s = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0)
s.set_image(w, h, spp, depth, (1, 1, 1)).build()
cam = s.camera(0.0)
ctx = pt.Context(0)
ctx.set_option("lanes", 1)
if len(sys.argv) > 2: ctx.set_option("trace_mode", int(sys.argv[2]))
ctx.upload(s)
ctx.accum_reset(w, h)
st = ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, spp, depth, 1)
print("rays", st.rays, "trace ms", round(st.ms_trace, 1), "per bounce", list(st.rays_per_bounce[:50]))
