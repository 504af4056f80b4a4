"""Generates tests/golden/*.npz from the CPU oracle (oracle/oracle.cpp).

The reference (Rust) cannot be compiled or run in this image, and its only test holds six Vec3 known
answers, so these vectors are outputs of OUR restatement, committed to pin it against regressions and to
give the GPU tests fixed ray batches with expected {prim, t, normal, front_face}.  They are data only.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_api as orc  # noqa: E402

SCENES = ["SphereField", "TwoSpheres", "LightTest", "CornellBox", "TriangleTest"]
W, H, SPP, DEPTH, SEED = 32, 18, 2, 8, 7
BG = {"SphereField": (1, 1, 1), "TwoSpheres": (1, 1, 1), "LightTest": (0, 0, 0), "CornellBox": (0, 0, 0), "TriangleTest": (0, 0, 0)}


def rays_for(scene, cam, n, seed):
    rng = np.random.default_rng(seed)
    origin, llc, hor, ver = cam[0:3], cam[3:6], cam[6:9], cam[9:12]
    s, t = rng.random((n, 1)), rng.random((n, 1))
    d = llc + s * hor + t * ver - origin
    o = np.broadcast_to(origin, d.shape).copy()
    prim, tt, normal, _ = scene.trace(o, d)
    hit = prim >= 0
    p = o[hit] + tt[hit, None] * d[hit]
    u = rng.normal(size=p.shape)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    return np.concatenate([o, p]), np.concatenate([d, normal[hit] + u])


def main():
    out = {}
    for name in SCENES:
        sc = orc.OracleScene.named(name, 1, 1, 1)
        cam = orc.scene_camera(name, 0.25, W / H)
        o, d = rays_for(sc, cam, 192, 11)
        prim, t, normal, front = sc.trace(o, d)
        accum, rays = sc.render_accumulate(cam, W, H, BG[name], 0, SPP, DEPTH, SEED, threads=1)
        out[f"{name}_cam"] = cam
        out[f"{name}_o"] = o
        out[f"{name}_d"] = d
        out[f"{name}_prim"] = prim
        out[f"{name}_t"] = t
        out[f"{name}_normal"] = normal
        out[f"{name}_front"] = front
        out[f"{name}_accum"] = accum
        out[f"{name}_rays"] = np.array([rays], dtype=np.uint64)
        out[f"{name}_nprims"] = np.array([sc.num_prims])
    out["draws"] = orc.sample_draws(0x1234567890ABCDEF, 17, 3, 2, 16)
    out["params"] = np.array([W, H, SPP, DEPTH, SEED])
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_vectors.npz"))


if __name__ == "__main__":
    main()
