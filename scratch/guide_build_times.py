# usage (GPU box): python scratch/guide_build_times.py   -- upload time and frame time with the guide tree built on the host / on the device
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import rust_pathtracer_amd as pt

def run(name, scene, w, h, spp, depth, bg):
    for build in (0, 1):
        ctx = pt.Context(0)
        ctx.set_option("guide_build", build)
        ups = []
        for _ in range(3):
            t0 = time.perf_counter(); ctx.upload(scene); ups.append(time.perf_counter() - t0)
        cam = scene.camera(0.0)
        ms = []
        for k in range(4):
            ctx.accum_reset(w, h)
            t0 = time.perf_counter()
            st = ctx.render_accumulate(cam, w, h, bg, 0, spp, depth, 1)
            ms.append((time.perf_counter() - t0) * 1e3)
        print(f"{name}: guide_build={build}: upload {min(ups)*1e3:.1f} ms, frame {min(ms[1:]):.1f} ms, rays {st.rays}, trace busy {st.ms_busy_trace:.1f} path {st.ms_busy_path:.1f}", flush=True)
        ctx.close()

s = pt.Scene("SphereField").set_seeds(1, 1, 1).set_image(1920, 1080, 32, 50, (1, 1, 1)).build()
run("SphereField 1080p 32spp", s, 1920, 1080, 32, 50, (1, 1, 1))
m = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0).set_image(1920, 1080, 16, 50, (1, 1, 1)).build()
run("mesh 1M 1080p 16spp", m, 1920, 1080, 16, 50, (1, 1, 1))
