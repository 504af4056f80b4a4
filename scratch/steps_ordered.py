# debugging aid: per-ray step counts of the ordered walk (library built with -DPT_DEBUG_STEPS; PT_LIBDIR points at it)
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
import scenes_util as su
name = sys.argv[1] if len(sys.argv) > 1 else "SphereField"
s = pt.Scene(name).set_seeds(1, 1, 1)
if name == "ModelTest":
    s.set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0)
s.set_image(1920, 1080, 32, 50, (1, 1, 1)).build()
ctx = pt.Context(0)
ctx.set_option("trace_mode", 3)
ctx.upload(s)
cam = s.camera(0.0).as_array()
def report(label, ro, rd):
    t = ctx.trace_batch(ro, rd)["t"]
    hit = ctx.trace_batch(ro, rd)["prim"] >= 0
    v = np.round(t).astype(np.int64)
    lit = v // 10**14; v %= 10**14
    literal = v // 10**9; v %= 10**9
    verify = v // 10**6; v %= 10**6
    leaf = v // 10**3; guide = v % 10**3
    print("%s: hit %.3f guide %.1f (max %d) leaf %.2f (max %d) verify %.1f literal-steps %.2f literal-rays %.5f" % (
        label, hit.mean(), guide.mean(), guide.max(), leaf.mean(), leaf.max(), verify.mean(), literal.mean(), lit.mean()))
    if hit.any():
        print("   hits only: guide %.1f leaf %.2f verify %.1f" % (guide[hit].mean(), leaf[hit].mean(), verify[hit].mean()))
ro, rd = su.camera_rays(cam, 1920, 1080, 64 * 2000, 1)
print(name, "prims", s.desc.n_prims, "nodes", s.desc.n_nodes)
report("primary", ro, rd)
h = ctx.trace_batch(ro, rd)
# second segment: from the hit points, random directions
ctx2 = None
rng = np.random.default_rng(2)
if name == "ModelTest":
    ro2 = rng.normal(size=ro.shape); ro2 = ro2 / np.linalg.norm(ro2, axis=1)[:, None] * 160.0 + np.array([0.0, 150.0, 0.0])
    rd2 = rng.normal(size=ro.shape)
else:
    rd2 = rng.normal(size=ro.shape); ro2 = rng.uniform(-8, 8, ro.shape); ro2[:, 1] = rng.uniform(0.2, 3, len(ro))
report("incoherent", ro2, rd2)
