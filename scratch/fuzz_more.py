# one-off: the ordered-walk fuzz of tests/test_gpu_parity.py over many more seeds
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import scenes_util as su
import oracle_api as orc
import rust_pathtracer_amd as pt
ctx = pt.Context(0)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n_objects = int(rng.choice([2, 3, 8, 30, 90, 200]))
    s, o = su.random_world(seed, n_objects=n_objects, with_mesh=bool(seed % 2), extent=float(rng.choice([2.0, 10.0, 300.0])))
    ctx.set_option("trace_mode", 3)
    ctx.upload(s)
    ro, rd = su.random_rays(seed + 100, 20000)
    inside_o = rng.uniform(-10, 10, (10000, 3))
    inside_d = rng.normal(size=(10000, 3)) * 10.0 ** rng.uniform(-3, 3, (10000, 1))
    p2, d2 = su.bounce_rays(o, ro, rd, seed + 200)
    ro, rd = np.concatenate([ro, inside_o, p2]), np.concatenate([rd, inside_d, d2])
    for t_min, t_max in [(0.001, np.inf), (-3.0, 3.0), (-np.inf, np.inf), (0.0, 1e-3)]:
        prim, t, normal, front = o.trace(ro, rd, t_min, t_max)
        h = ctx.trace_batch(ro, rd, t_min, t_max)
        hit = prim >= 0
        ok = np.array_equal(h["prim"], prim) and np.array_equal(h["t"][hit], t[hit], equal_nan=True)
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "objects", n_objects, "window", t_min, t_max, (h["prim"] != prim).sum(), flush=True)
    if seed % 10 == 0: print("seed", seed, "ok so far, mismatches", bad, flush=True)
print("done, mismatches", bad)
