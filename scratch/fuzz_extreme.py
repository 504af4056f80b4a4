# one-off: every walk against the oracle on random worlds with rays salted with extreme components (zeros, denormals, 1e300, inf)
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import scenes_util as su
import oracle_api as orc
import rust_pathtracer_amd as pt
ctx = pt.Context(0)
odd = np.array([0.0, -0.0, 5e-324, 1e-300, -1e-310, 2.0 ** -801, 2.0 ** -101, 2.0 ** -99, 2.0 ** 99, 2.0 ** 101, 1e200, -1e300, 2.0 ** 401, np.inf, -np.inf, 1.0, -2.5])
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n_objects = int(rng.choice([3, 10, 40]))
    s, o = su.random_world(seed, n_objects=n_objects, with_mesh=bool(seed % 2))
    n = 64 * 300
    ro, rd = su.random_rays(seed + 100, n)
    which = rng.random(n)
    rows = np.nonzero(which < 0.3)[0]
    ro[rows, rng.integers(0, 3, len(rows))] = rng.choice(odd[:13], len(rows))
    rows = np.nonzero((which >= 0.3) & (which < 0.6))[0]
    rd[rows, rng.integers(0, 3, len(rows))] = rng.choice(odd, len(rows))
    for mode in (0, 2, 3):
        ctx.set_option("trace_mode", mode)
        ctx.upload(s)
        for t_min, t_max in [(0.001, np.inf), (-1.0, 7.0), (0.0, np.inf)]:
            with np.errstate(all="ignore"):
                prim, t, normal, front = o.trace(ro, rd, t_min, t_max)
            h = ctx.trace_batch(ro, rd, t_min, t_max)
            hit = prim >= 0
            ok = np.array_equal(h["prim"], prim) and np.array_equal(h["t"][hit], t[hit], equal_nan=True)
            if not ok:
                bad += 1
                i = np.nonzero(h["prim"] != prim)[0]
                print("MISMATCH seed", seed, "mode", mode, "window", t_min, t_max, len(i), "first ray", ro[i[0]] if len(i) else None, rd[i[0]] if len(i) else None,
                      "oracle", prim[i[0]] if len(i) else None, "gpu", h["prim"][i[0]] if len(i) else None, flush=True)
    if seed % 5 == 0: print("seed", seed, "mismatches so far", bad, flush=True)
print("done, mismatches", bad)
