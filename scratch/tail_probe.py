# usage (GPU box): python scratch/tail_probe.py   -- how the per-wave tail launch of the mesh frame scales with the sample count
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
m = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0).set_image(1920, 1080, 16, 50, (1, 1, 1)).build()
ctx = pt.Context(0)
ctx.upload(m)
cam = m.camera(0.0)
w, h = 1920, 1080
for lanes in (1, 3):
    ctx.set_option("lanes", lanes)
    for spp in (1, 4, 16, 64):
        for k in range(2):
            ctx.accum_reset(w, h)
            st = ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, spp, 50, 1)
        rb = [int(x) for x in st.rays_per_bounce[:50]]
        print(f"lanes {lanes} spp {spp}: total {st.ms_total:.2f} trace {st.ms_trace:.2f} shade {st.ms_shade:.2f} tail {st.ms_path:.2f} raygen {st.ms_raygen:.2f}; rays/bounce {rb[:6]} ... last nonzero bounce {max(i for i, x in enumerate(rb) if x)}; b10 {rb[10]} b20 {rb[20]} b30 {rb[30]} b40 {rb[40]}", flush=True)
