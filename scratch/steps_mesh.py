# debugging aid: per-ray step counts of the per-lane stream walk on the 1M-triangle mesh (library built with -DPT_DEBUG_STEPS)
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
import scenes_util as su
s = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0)
s.set_image(1920, 1080, 8, 50, (1, 1, 1)).build()
ctx = pt.Context(0)
ctx.set_option("trace_mode", 2)
ctx.upload(s)
cam = s.camera(0.0).as_array()
ro, rd = su.camera_rays(cam, 1920, 1080, 64 * 2000, 1)
h = ctx.trace_batch(ro, rd)
t = h["t"]; nodes = np.floor(t); leaves = np.round((t - nodes) * 1e4)
hit = h["prim"] >= 1
print("primary: nodes/ray mean %.1f max %d; leaves/ray mean %.1f max %d; mesh-hit fraction %.3f" % (nodes.mean(), nodes.max(), leaves.mean(), leaves.max(), hit.mean()))
print("  rays hitting the mesh: nodes mean %.1f leaves mean %.1f" % (nodes[hit].mean(), leaves[hit].mean()))
w = (nodes + leaves).reshape(-1, 64)
print("  per 64-ray batch: mean steps %.1f, mean of max %.1f" % (w.mean(), w.max(axis=1).mean()))
