# usage (GPU box): PT_LIBDIR=.../var_steps python scratch/step_latency.py count ; rocprofv3 --kernel-trace ... -- python3 scratch/step_latency.py time
# one ray per launch (rays of bounce 10 from step_stats.py gen): kernel duration against the ray's step count
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import rust_pathtracer_amd as pt
m = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0).set_image(1920, 1080, 16, 50, (1, 1, 1)).build()
ctx = pt.Context(0)
ctx.upload(m)
z = np.load("gpurun_out/step_rays.npz")
o, d = z["o10"][:100], z["d10"][:100]
if sys.argv[1] == "count":
    hit = ctx.trace_batch(o, d)
    np.save("gpurun_out/step_counts.npy", hit["t"])
else:
    for rep in range(2):
        for i in range(len(o)):
            ctx.trace_batch(o[i:i + 1], d[i:i + 1])
