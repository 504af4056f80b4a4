"""The CPU oracle against everything that can pin it without running the (unbuildable) Rust reference:
the reference's own six known answers, published known-answer vectors of the generators it uses, analytic
hits derived by hand from scene.rs constants, brute force vs BVH, and the committed golden vectors."""
import math
import os

import numpy as np
import pytest

import scenes_util as su

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.npz")


# ---------------------------------------------------------------- the reference's only test: vec3.rs:256-274
def test_vec3_known_answers(oracle):
    import ctypes as C

    out = (C.c_double * 16)()
    oracle.lib().orc_vec3_kat(out)
    v = list(out)
    assert v[0:3] == [1.0, 1.0, 0.0]       # a + b                 vec3.rs:267
    assert v[3:6] == [1.0, -1.0, 0.0]      # a - b                 vec3.rs:268
    assert v[6:9] == [-1.0, -1.0, -1.0]    # -c                    vec3.rs:269
    assert v[15] == 2.0                    # (a + b).dot(c)        vec3.rs:270
    assert v[9:12] == [0.0, 0.0, 1.0]      # a.cross(b)            vec3.rs:271
    assert v[12:15] == [0.0, 1.0, 0.0]     # d.unit_vector()       vec3.rs:272


# ---------------------------------------------------------------- generators: published known answers
def test_philox4x32_10_random123_vectors(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert oracle.philox([0] * 4, [0] * 2) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_splitmix64_vectors(oracle):
    out = np.zeros(3, np.uint64)
    oracle.lib().orc_splitmix(0, 3, out.ctypes.data)
    assert [int(x) for x in out] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_sample_draws_layout(oracle):
    # draw k = 53 high bits of half k%2 of philox block k/2 with counter (pixel, sample, stream, k/2)
    seed, pixel, sample, stream = 0x1234567890ABCDEF, 17, 3, 2
    d = oracle.sample_draws(seed, pixel, sample, stream, 6)
    key = [seed & 0xFFFFFFFF, seed >> 32]
    for k in range(6):
        blk = oracle.philox([pixel, sample, stream, k // 2], key)
        bits = (blk[3] << 32 | blk[2]) if k & 1 else (blk[1] << 32 | blk[0])
        assert d[k] == (bits >> 11) * 2.0 ** -53
    assert ((d >= 0) & (d < 1)).all()


def test_perlin_table_is_a_permutation(oracle):
    t = np.zeros(256, np.uint8)
    oracle.lib().orc_perlin_table(1, t.ctypes.data)
    assert sorted(t.tolist()) == list(range(256))
    t2 = np.zeros(256, np.uint8)
    oracle.lib().orc_perlin_table(2, t2.ctypes.data)
    assert not np.array_equal(t, t2)


# ---------------------------------------------------------------- analytic hits in the Cornell box (scene.rs:367-464)
def test_cornell_analytic_hits(oracle):
    sc = oracle.OracleScene.named("CornellBox", 1, 1, 1)
    assert sc.num_prims == 20  # 6 rects + 2 boxes x 6 sides + 2 spheres
    o = np.array([[278, 278, -800], [278, 278, -800], [278, 500, -800], [-100, 278, -800]], dtype=float)
    d = np.array([[0, 0, 1], [-178, -278, 850], [0, 0, 1], [0, 0, 1]], dtype=float)
    prim, t, n, ff = sc.trace(o, d)
    # glass sphere c=(212.5,255,147.5) r=90 (scene.rs:452-456): t = 947.5 - sqrt(8100 - 65.5^2 - 23^2)
    assert prim[0] >= 0 and abs(t[0] - (947.5 - math.sqrt(3280.75))) < 1e-9 and ff[0] == 1
    # floor y=0 (scene.rs:412-420): exactly t=1, normal (0,1,0), front face
    assert t[1] == 1.0 and n[1].tolist() == [0.0, 1.0, 0.0] and ff[1] == 1
    # back wall z=555 (scene.rs:431-439): t=1355, normal (0,0,-1)
    assert t[2] == 1355.0 and n[2].tolist() == [0.0, 0.0, -1.0] and ff[2] == 1
    # outside the opening: nothing
    assert prim[3] == -1
    # the light (scene.rs:421-429) is one-sided: from below it is a front face with normal (0,-1,0)
    prim, t, n, ff = sc.trace(np.array([[278.0, 100.0, 279.5]]), np.array([[0.0, 1.0, 0.0]]))
    assert t[0] == 454.0 and n[0].tolist() == [0.0, -1.0, 0.0] and ff[0] == 1


def test_sphere_negative_radius_flips_normal(oracle):
    # geometry.rs:151: outward_normal = (p - c) / radius, so a negative radius points it inward (hollow glass)
    objs = [su._obj(0, 0, [0, 0, 0, -1.0])[1], su._obj(0, 0, [10, 0, 0, 1.0])[1]]
    sc = oracle.OracleScene.custom(objs, [su._mat(2, 0, 1.5)[1]], [su._tex(0)[1]])
    prim, t, n, ff = sc.trace(np.array([[0.0, 0.0, -5.0]]), np.array([[0.0, 0.0, 1.0]]))
    assert t[0] == 4.0 and ff[0] == 0 and n[0].tolist() == [0.0, 0.0, -1.0]


def test_triangle_is_single_sided_and_scale_dependent(oracle):
    # geometry.rs:526: det < 0.0001 -> miss (back faces and tiny/grazing triangles)
    tri = su._obj(6, 0, [0, 0, 0, 1, 0, 0, 0, 1, 0])[1]
    far = su._obj(0, 0, [100, 100, 100, 1.0])[1]
    sc = oracle.OracleScene.custom([tri, far], [su._mat(0, 0)[1]], [su._tex(0, color=(1, 1, 1))[1]])
    o = np.array([[0.25, 0.25, 1.0], [0.25, 0.25, -1.0]])
    d = np.array([[0.0, 0.0, -1.0], [0.0, 0.0, 1.0]])
    prim, t, n, ff = sc.trace(o, d)
    assert prim[0] == 0 and t[0] == 1.0 and ff[0] == 1 and n[0].tolist() == [0.0, 0.0, 1.0]
    assert prim[1] == -1  # from behind
    tiny = su._obj(6, 0, [0, 0, 0, 0.005, 0, 0, 0, 0.005, 0])[1]  # det = 2.5e-5 for |d|=1
    sc = oracle.OracleScene.custom([tiny, far], [su._mat(0, 0)[1]], [su._tex(0)[1]])
    prim, _, _, _ = sc.trace(np.array([[0.001, 0.001, 1.0]]), np.array([[0.0, 0.0, -1.0]]))
    assert prim[0] == -1
    prim, _, _, _ = sc.trace(np.array([[0.001, 0.001, 1.0]]), np.array([[0.0, 0.0, -10.0]]))  # longer d: det x10
    assert prim[0] == 0


def test_rectangle_edges_inclusive_and_ties_go_to_the_later_primitive(oracle):
    # geometry.rs:219: strict comparisons, so a hit exactly on the edge counts;
    # geometry.rs:143-146 / bvh.rs:113-120: t == t_max is accepted, so of two coincident surfaces the later one wins
    a = su._obj(2, 0, [0, 0, 0, 1, 0, 1, 1.0])[1]
    b = su._obj(2, 0, [0, 0, 0, 2, 0, 2, 1.0])[1]
    sc = oracle.OracleScene.custom([a, b], [su._mat(0, 0)[1]], [su._tex(0)[1]], bvh_seed=3)
    prim, t, _, _ = sc.trace(np.array([[1.0, 1.0, 0.5]]), np.array([[0.0, -1.0, 0.0]]))
    bprim, bt = sc.trace_brute(np.array([[1.0, 1.0, 0.5]]), np.array([[0.0, -1.0, 0.0]]))
    assert t[0] == 1.0 and prim[0] == 1 and bprim[0] == 1 and bt[0] == 1.0


# ---------------------------------------------------------------- brute force == BVH
@pytest.mark.parametrize("name", su.NAMED)
def test_bvh_equals_linear_scan_named(oracle, name):
    sc = oracle.OracleScene.named(name, 1, 1, 1)
    cam = oracle.scene_camera(name, 0.1, 16 / 9)
    o, d = su.camera_rays(cam, 64, 36, 20000, 5)
    p2, d2 = su.bounce_rays(sc, o, d, 6)
    o = np.concatenate([o, p2])
    d = np.concatenate([d, d2])
    prim, t, _, _ = sc.trace(o, d)
    bprim, bt = sc.trace_brute(o, d)
    assert (prim >= 0).mean() > 0.2
    assert np.array_equal(prim, bprim) and np.array_equal(t, bt)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_bvh_equals_linear_scan_random_worlds(oracle, seed):
    _, sc = su.random_world(seed, n_objects=40, with_mesh=(seed % 2 == 0))
    o, d = su.random_rays(seed + 100, 60000)
    prim, t, _, _ = sc.trace(o, d)
    bprim, bt = sc.trace_brute(o, d)
    assert (prim >= 0).mean() > 0.05
    assert np.array_equal(prim, bprim) and np.array_equal(t, bt)


# ---------------------------------------------------------------- materials (material.rs) and vec3 helpers
def test_reflect_refract_schlick(oracle):
    import ctypes as C

    L = oracle.lib()
    out = (C.c_double * 3)()
    s = 1 / math.sqrt(2)
    L.orc_reflect(oracle.d3([s, -s, 0]), oracle.d3([0, 1, 0]), out)
    assert np.allclose(list(out), [s, s, 0], atol=1e-15)
    L.orc_refract(oracle.d3([0, -1, 0]), oracle.d3([0, 1, 0]), 1 / 1.5, out)  # normal incidence: straight through
    assert np.allclose(list(out), [0, -1, 0], atol=1e-15)
    L.orc_refract(oracle.d3([s, -s, 0]), oracle.d3([0, 1, 0]), 1.0, out)  # equal indices: unchanged
    assert np.allclose(list(out), [s, -s, 0], atol=1e-15)
    assert abs(L.orc_reflectance(1.0, 1.5) - 0.04) < 1e-15  # ((1-1.5)/(1+1.5))^2
    assert abs(L.orc_reflectance(0.0, 1.5) - 1.0) < 1e-15


def test_scatter_rules(oracle):
    n = [0.0, 1.0, 0.0]
    # Lambertian (material.rs:41-55): direction = normal + unit vector, attenuation = albedo
    ok, d, att, em = oracle.scatter_probe(0, 0.0, [0.5, 0.6, 0.7], [0, 1, 0], [0.3, -1, 0.2], 1.0, n)
    assert ok and abs(np.linalg.norm(d - np.array(n)) - 1.0) < 1e-12 and att.tolist() == [0.5, 0.6, 0.7] and not em.any()
    # Metal, fuzz 0 (material.rs:77-93): mirror direction of the unit incoming direction
    ok, d, att, _ = oracle.scatter_probe(1, 0.0, [0.7, 0.6, 0.5], [0, 1, 0], [3.0, -4.0, 0.0], 1.0, n)
    assert ok and np.allclose(d, [0.6, 0.8, 0.0], atol=1e-15)
    # Metal hit from inside the surface plane: the normal is flipped to oppose the ray, reflection still leaves
    ok, d, _, _ = oracle.scatter_probe(1, 0.0, [1, 1, 1], [0, -1, 0], [0.0, 1.0, 0.0], 1.0, n)
    assert ok and np.allclose(d, [0, -1, 0])
    # Dielectric (material.rs:117-143): attenuation 1; total internal reflection when ratio * sin > 1
    ok, d, att, _ = oracle.scatter_probe(2, 1.5, [0, 0, 0], [0, -1, 0], [0.9, 0.2, 0.0], 1.0, n)  # from inside, grazing
    assert ok and att.tolist() == [1.0, 1.0, 1.0]
    u = np.array([0.9, 0.2, 0.0]) / np.linalg.norm([0.9, 0.2, 0.0])
    assert np.allclose(d, [u[0], -u[1], 0.0], atol=1e-15)  # reflected about the flipped normal
    # DiffuseLight (material.rs:158-166): emits on the front face only, never scatters
    ok, _, _, em = oracle.scatter_probe(3, 0.0, [4, 4, 4], [0, 1, 0], [0, -1, 0], 1.0, n)
    assert not ok and em.tolist() == [4.0, 4.0, 4.0]
    ok, _, _, em = oracle.scatter_probe(3, 0.0, [4, 4, 4], [0, -1, 0], [0, 1, 0], 1.0, n)
    assert not ok and not em.any()


def test_textures(oracle):
    # checker (texture.rs:36-45): sines < 0 -> odd (0.2,0.3,0.1) else even (0.9,...)
    pts = np.array([[0.1, 0.1, 0.1], [-0.1, 0.1, 0.1], [0.1, 0.1, 0.1 + math.pi / 10], [0.0, 5.0, 7.0]])
    c = oracle.texture_value(0, 0.0, 0, pts)
    assert c[0].tolist() == [0.9, 0.9, 0.9] and c[1].tolist() == [0.2, 0.3, 0.1] and c[2].tolist() == [0.2, 0.3, 0.1]
    assert c[3].tolist() == [0.9, 0.9, 0.9]  # sin(0) = 0 is not < 0
    # marble (texture.rs:75-80): grey, 0.5 * (1 - sin(...)) in [0, 1]
    rng = np.random.default_rng(0)
    pts = rng.uniform(-20, 20, (2000, 3))
    m = oracle.texture_value(1, 4.0, 1, pts)
    assert (m[:, 0] == m[:, 1]).all() and (m[:, 1] == m[:, 2]).all()
    assert m.min() >= 0.0 and m.max() <= 1.0 and m.std() > 0.2
    # the noise itself: zero on the integer lattice, bounded
    import ctypes as C

    assert oracle.lib().orc_perlin_get(1, oracle.d3([3.0, -2.0, 7.0])) == 0.0
    vals = [oracle.lib().orc_perlin_get(1, oracle.d3(p)) for p in pts[:500]]
    assert max(abs(v) for v in vals) <= 1.0 and np.std(vals) > 0.1


def test_color_to_u8_saturates_like_rust(oracle):
    f = oracle.lib().orc_to_u8  # vec3.rs:278-290: (c * 255.999) as u8
    assert f(0.0) == 0 and f(1.0) == 255 and f(0.5) == 127 and f(2.0) == 255 and f(-1.0) == 0
    assert f(float("nan")) == 0 and f(float("inf")) == 255
    assert f(1.0 / 255.999) == 1 and f(0.999 / 255.999) == 0


# ---------------------------------------------------------------- renderer semantics (renderer.rs / ray.rs)
def test_render_depth_and_background(oracle):
    sc = oracle.OracleScene.named("TwoSpheres", 1, 1, 1)
    cam = oracle.scene_camera("TwoSpheres", 0.0, 2.0)
    a, rays = sc.render_accumulate(cam, 8, 4, (1, 1, 1), 0, 2, 0, 1, threads=1)
    assert rays == 0 and not a.any()  # ray.rs:21-23: depth 0 is black, no hit query
    a, rays = sc.render_accumulate(cam, 8, 4, (1, 1, 1), 0, 1, 1, 1, threads=1)
    assert rays == 32  # one query per primary ray; hits return black (emitted + att * black)
    assert set(np.unique(a)) <= {0.0, 1.0}
    # misses return the background untouched
    sc = oracle.OracleScene.named("CornellBox", 1, 1, 1)
    camc = oracle.camera_new((278, 278, -800), (278, 278, -2000), (0, 1, 0), 40.0, 1.0, 0.0, 10.0)  # looking away
    a, rays = sc.render_accumulate(camc, 4, 4, (0.25, 0.5, 0.75), 0, 3, 5, 1, threads=1)
    assert rays == 48 and np.allclose(a, np.array([0.75, 1.5, 2.25]))


def test_render_is_thread_and_range_invariant(oracle):
    sc = oracle.OracleScene.named("CornellBox", 1, 1, 1)
    cam = oracle.scene_camera("CornellBox", 0.0, 2.0)
    a1, r1 = sc.render_accumulate(cam, 16, 8, (0, 0, 0), 0, 6, 10, 9, threads=1)
    a4, r4 = sc.render_accumulate(cam, 16, 8, (0, 0, 0), 0, 6, 10, 9, threads=4)
    assert r1 == r4 and np.array_equal(a1, a4)
    lo, rl = sc.render_accumulate(cam, 16, 8, (0, 0, 0), 0, 2, 10, 9, threads=2)
    hi, rh = sc.render_accumulate(cam, 16, 8, (0, 0, 0), 2, 4, 10, 9, threads=2)
    assert rl + rh == r1 and np.allclose(lo + hi, a1, rtol=1e-13, atol=1e-13)


# ---------------------------------------------------------------- committed golden vectors
@pytest.mark.parametrize("name", su.NAMED)
def test_oracle_matches_golden(oracle, name):
    g = np.load(GOLDEN)
    w, h, spp, depth, seed = [int(x) for x in g["params"]]
    sc = oracle.OracleScene.named(name, 1, 1, 1)
    assert sc.num_prims == int(g[f"{name}_nprims"][0])
    assert np.array_equal(oracle.scene_camera(name, 0.25, w / h), g[f"{name}_cam"])
    prim, t, normal, front = sc.trace(g[f"{name}_o"], g[f"{name}_d"])
    assert np.array_equal(prim, g[f"{name}_prim"]) and np.array_equal(t, g[f"{name}_t"])
    assert np.array_equal(normal, g[f"{name}_normal"]) and np.array_equal(front, g[f"{name}_front"])
    bg = (1, 1, 1) if name in ("SphereField", "TwoSpheres") else (0, 0, 0)
    accum, rays = sc.render_accumulate(g[f"{name}_cam"], w, h, bg, 0, spp, depth, seed)
    assert rays == int(g[f"{name}_rays"][0])
    # libm's sin may differ by an ulp between builds: the texture values, hence the sums, are pinned to 1e-12
    assert np.allclose(accum, g[f"{name}_accum"], rtol=1e-12, atol=1e-12)


def test_golden_draws(oracle):
    g = np.load(GOLDEN)
    assert np.array_equal(oracle.sample_draws(0x1234567890ABCDEF, 17, 3, 2, 16), g["draws"])
