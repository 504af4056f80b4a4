# debugging aid: the slowest rays of the ordered walk on the mesh scene (library built with -DPT_DEBUG_STEPS; PT_LIBDIR points at it)
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
s = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0)
s.set_image(1920, 1080, 8, 50, (1, 1, 1)).build()
ctx = pt.Context(0)
ctx.upload(s)
rng = np.random.default_rng(5)
n = 1 << 21
# rays leaving points just outside the mesh surface (radius ~150 around (0,150,0)) in random directions, like bounce rays
u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
ro = u * rng.uniform(150.0, 151.0, (n, 1)) + np.array([0.0, 150.0, 0.0])
rd = rng.normal(size=(n, 3))
h = ctx.trace_batch(ro, rd)
v = np.round(h["t"]).astype(np.int64)
lit = v // 10**14; v %= 10**14
literal = v // 10**9; v %= 10**9
verify = v // 10**6; v %= 10**6
leaf = v // 10**3; guide = v % 10**3
total = guide + leaf + verify + literal
print("rays", n, "hit", (h["prim"] >= 0).mean())
print("guide mean %.1f p99 %d p99.99 %d max %d" % (guide.mean(), np.percentile(guide, 99), np.percentile(guide, 99.99), guide.max()))
print("leaf  mean %.2f p99 %d p99.99 %d max %d" % (leaf.mean(), np.percentile(leaf, 99), np.percentile(leaf, 99.99), leaf.max()))
print("literal rays %d (%.2e), their literal steps mean %.0f max %d" % (lit.sum(), lit.mean(), literal[lit > 0].mean() if lit.any() else 0, literal.max()))
print("total steps mean %.1f p99.99 %d max %d" % (total.mean(), np.percentile(total, 99.99), total.max()))
