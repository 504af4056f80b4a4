import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import conftest  # noqa
import scenes_util as su
import test_gpu_parity as T
from rust_pathtracer_amd import api
import rust_pathtracer_amd as pt
np.set_printoptions(precision=17, linewidth=200)
s, o = T._box_world(False)
ctx = pt.Context(0)
ctx.set_option("trace_mode", 3)
ctx.upload(s)
rng = np.random.default_rng(11)
n = 64 * 400
ro = rng.uniform(-4, 4, (n, 3))
rd = rng.uniform(-3, 3, (n, 3)) - ro * 0.5
odd_o = np.array([0.0, -0.0, 5e-324, 1e-300, -1e-310, 2.0 ** -801, 2.0 ** -799, 1e200, -1e300, 2.0 ** 401, 0.25, -2.0, 0.5, 1.5])
odd_d = np.array([0.0, -0.0, 5e-324, 1e-300, -1e-200, 2.0 ** -151, 2.0 ** -149, 2.0 ** 149, 2.0 ** 151, 1e200, -1e300, np.inf, 1.0, -1.0])
q = n // 4
rows = np.arange(q)
ro[rows, rng.integers(0, 3, q)] = rng.choice(odd_o, q)
rows = np.arange(q) + q
rd[rows[: q // 2], rng.integers(0, 3, q // 2)] = rng.choice(odd_d, q // 2)
ro[rows[q // 2 :], rng.integers(0, 3, q - q // 2)] = rng.choice(odd_o[:9], q - q // 2)
salt = rng.random(n) < 0.025
salt[: 2 * q] = False
ro[salt, 0] = rng.choice(odd_o, salt.sum())
rd[salt, 1] = rng.choice(odd_d, salt.sum())
ro[2 * q : 2 * q + 2000, 1] = 0.0
rd[2 * q + 2000 : 2 * q + 3000, 2] = 0.0
for t_min, t_max in [(0.001, np.inf), (0.0, np.inf), (-1.0, 7.0)]:
    with np.errstate(all="ignore"):
        prim, t, normal, front = o.trace(ro, rd, t_min, t_max)
    h = ctx.trace_batch(ro, rd, t_min, t_max)
    bad = np.nonzero(h["prim"] != prim)[0]
    print("window", t_min, t_max, "differ", len(bad))
    for i in bad[:8]:
        print("  ray", i, "o", ro[i], "d", rd[i], "oracle", prim[i], t[i], "gpu", h["prim"][i], h["t"][i])
