# debugging aid: per-ray step counts of the per-lane stream walk (library built with -DPT_DEBUG_STEPS; PT_LIBDIR points at it)
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
import scenes_util as su
name = sys.argv[1] if len(sys.argv) > 1 else "SphereField"
s = pt.Scene(name).set_seeds(1, 1, 1).set_image(1920, 1080, 32, 50, (1, 1, 1)).build()
ctx = pt.Context(0)
ctx.set_option("trace_mode", 2)
ctx.upload(s)
cam = s.camera(0.0).as_array()
ro, rd = su.camera_rays(cam, 1920, 1080, 64 * 2000, 1)
h = ctx.trace_batch(ro, rd)
t = h["t"]
nodes = np.floor(t); leaves = np.round((t - nodes) * 1e4)
print(name, "prims", s.desc.n_prims, "nodes", s.desc.n_nodes)
print("primary (random pixels): nodes/ray mean %.1f max %d; leaves/ray mean %.1f max %d" % (nodes.mean(), nodes.max(), leaves.mean(), leaves.max()))
w = (nodes + leaves).reshape(-1, 64)
print("  per 64-ray batch: mean of sum-steps %.1f, mean of max %.1f" % (w.mean(), w.max(axis=1).mean()))
# coherent: tiles of 8x8 neighbouring pixels
rng = np.random.default_rng(2)
rd2 = rng.normal(size=ro.shape); ro2 = rng.uniform(-8, 8, ro.shape); ro2[:, 1] = rng.uniform(0.2, 3, len(ro))
h = ctx.trace_batch(ro2, rd2); t = h["t"]; nodes = np.floor(t); leaves = np.round((t - nodes) * 1e4)
print("incoherent interior rays: nodes/ray mean %.1f max %d; leaves/ray mean %.1f max %d" % (nodes.mean(), nodes.max(), leaves.mean(), leaves.max()))
w = (nodes + leaves).reshape(-1, 64); wn = nodes.reshape(-1, 64); wl = leaves.reshape(-1, 64)
print("  per 64-ray batch: mean steps %.1f, mean of max %.1f; max nodes %.1f max leaves %.1f" % (w.mean(), w.max(axis=1).mean(), wn.max(axis=1).mean(), wl.max(axis=1).mean()))
