# one-off: whole renders (ordered walk, tail launch from a random bounce, 1-3 lanes) against the oracle on random worlds
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import scenes_util as su
import oracle_api as orc
import rust_pathtracer_amd as pt
from rust_pathtracer_amd import api
ctx = pt.Context(0)
bad = 0
w, h, spp, depth = 48, 32, 4, 20
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    s, o = su.random_world(seed, n_objects=int(rng.choice([6, 40, 120])), with_mesh=bool(seed % 2))
    cam = api.camera_new((13.0, 6.0, 9.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, w / h, 0.05, 12.0)
    bg = (0.6, 0.7, 0.9)
    want, rays = o.render_accumulate(cam.as_array(), w, h, bg, 0, spp, depth, seed)
    for mode in (-1, 3):
        ctx.set_option("trace_mode", mode)
        ctx.set_option("tail_from", int(rng.choice([-1, 0, 2, 5, 4096])))
        ctx.set_option("lanes", int(rng.integers(1, 4)))
        ctx.set_option("deep_tree", int(rng.integers(-1, 2)))
        ctx.upload(s)
        ctx.accum_reset(w, h)
        st = ctx.render_accumulate(cam, w, h, bg, 0, spp, depth, seed)
        got = ctx.accum_download().astype(np.float64)
        err = np.abs(got - want).max()
        if st.rays != rays or err > 1e-4 * max(1.0, np.abs(want).max()):
            bad += 1
            print("MISMATCH seed", seed, "mode", mode, "rays", st.rays, rays, "err", err, flush=True)
    if seed % 5 == 0: print("seed", seed, "mismatches so far", bad, flush=True)
print("done, mismatches", bad)
