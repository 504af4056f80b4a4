# usage (GPU box): python scratch/step_stats.py gen && PT_LIBDIR=rust_pathtracer_amd/build/var_steps python scratch/step_stats.py count
# gen: rays of a diffuse random walk over the 1M-triangle mesh (bounces 0, 1, 4, 10) -> gpurun_out/step_rays.npz
# count: with a -DPT_DEBUG_STEPS build, the distribution of the ordered walk's step counts per ray
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import rust_pathtracer_amd as pt
m = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0).set_image(1920, 1080, 16, 50, (1, 1, 1)).build()
ctx = pt.Context(0)
for a in sys.argv[2:]:
    k, v = a.split("="); ctx.set_option(k, int(v))
ctx.upload(m)
if sys.argv[1] == "gen":
    cam = m.camera(0.0)
    rng = np.random.default_rng(7)
    w, h = 480, 270
    u, v = np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h)
    o0 = np.array(cam.origin); llc = np.array(cam.lower_left_corner); hz = np.array(cam.horizontal); vt = np.array(cam.vertical)
    d = llc[None, :] + u.reshape(-1, 1) * hz[None, :] + v.reshape(-1, 1) * vt[None, :] - o0[None, :]
    o = np.broadcast_to(o0, d.shape).copy()
    sets = {}
    for b in range(11):
        if b in (0, 1, 4, 10): sets[f"o{b}"] = o.copy(); sets[f"d{b}"] = d.copy()
        hit = ctx.trace_batch(o, d)
        ok = hit["prim"] >= 0
        p = o[ok] + hit["t"][ok, None] * d[ok]
        n = hit["normal"][ok]
        r = rng.normal(size=n.shape); r /= np.linalg.norm(r, axis=1, keepdims=True)
        o, d = p, n + r
        print("bounce", b, "rays", len(ok), "hits", int(ok.sum()), flush=True)
    np.savez("gpurun_out/step_rays.npz", **sets)
else:
    z = np.load("gpurun_out/step_rays.npz")
    for b in (0, 1, 4, 10):
        hit = ctx.trace_batch(z[f"o{b}"], z[f"d{b}"])
        c = hit["t"]
        lit = np.floor(c / 1e14); c = c - lit * 1e14
        litsteps = np.floor(c / 1e9); c = c - litsteps * 1e9
        ver = np.floor(c / 1e6); c = c - ver * 1e6
        leaf = np.floor(c / 1e3); guide = c - leaf * 1e3
        # (prim < 0 => t is reported as 0: misses carry no counts)
        okk = hit["prim"] >= 0
        def q(x): return "mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (x.mean(), *np.percentile(x, [50, 90, 99]), x.max())
        print(f"[{' '.join(sys.argv[2:]) or 'defaults'}] bounce {b}: {okk.sum()} hits of {len(okk)}; guide steps: {q(guide[okk])}; leaf: {q(leaf[okk])}; verify {q(ver[okk])}; literal rays {int(lit[okk].sum())}, literal steps {q(litsteps[okk])}", flush=True)
