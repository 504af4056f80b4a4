"""GPU tests added in round 2: the multi-GPU exchange through the C ABI (pt_accum_reduce), rays whose origin is far from
the scene relative to its primitives (the ordered walk's admission rule), the OBJ/MTL path end to end on the device,
failure behaviour of the frame loop, and the per-kernel busy times of PtStats."""
import os

import numpy as np
import pytest

import scenes_util as su
from rust_pathtracer_amd import api
from test_gpu_parity import assert_hits_equal, assert_images_close, render_pair

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- pt_accum_reduce
def test_accum_reduce_one_context_is_the_identity(pt):
    """n = 1 still goes through librccl (dlopen, ncclCommInitAll on one device, ncclReduce in place on the context's
    stream): the buffer must come back bit for bit, twice (the communicator is cached), and resolve as before."""
    w, h, spp, depth = 48, 32, 4, 12
    s = pt.Scene("CornellBox").set_image(w, h, spp, depth, (0, 0, 0)).build()
    ctx = pt.Context(0)
    try:
        ctx.upload(s)
        ctx.accum_reset(w, h)
        ctx.render_accumulate(s.camera(0.0), w, h, (0, 0, 0), 0, spp, depth, 7)
        before = ctx.accum_download()
        img = ctx.resolve(spp)
        assert before.any()
        pt.accum_reduce([ctx])
        assert np.array_equal(ctx.accum_download(), before)
        pt.accum_reduce([ctx])
        assert np.array_equal(ctx.accum_download(), before)
        assert np.array_equal(ctx.resolve(spp), img)
        # sample ranges rendered one after the other into the same buffer, then the (one-rank) reduce: what rank 0 of an
        # N-GPU frame does, with N = 1
        ctx.accum_reset(w, h)
        for b, n in [(0, 1), (1, 3)]:
            ctx.render_accumulate(s.camera(0.0), w, h, (0, 0, 0), b, n, depth, 7)
        pt.accum_reduce([ctx])
        assert np.allclose(ctx.accum_download(), before, rtol=1e-5, atol=1e-5)
    finally:
        ctx.close()


def test_accum_reduce_refuses_bad_arguments(pt):
    a, b = pt.Context(0), pt.Context(0)
    try:
        with pytest.raises(pt.PtError, match="before pt_accum_reset"):
            pt.accum_reduce([a])
        a.accum_reset(16, 8)
        b.accum_reset(16, 8)
        with pytest.raises(pt.PtError, match="one context per GPU"):
            pt.accum_reduce([a, b])  # two contexts on one device
        with pytest.raises(pt.PtError, match="one context per GPU"):
            pt.accum_reduce([a, a])
        b.accum_reset(8, 8)
        with pytest.raises(pt.PtError):
            pt.accum_reduce([a, b])
    finally:
        a.close()
        b.close()


# ---------------------------------------------------------------- far origins (the ordered walk's admission rule)
def _far_world():
    """Small and large primitives of every kind around the origin: what a ray from far away cancels against."""
    texs = [su._tex(0, color=(0.7, 0.7, 0.7))]
    mats = [su._mat(0, 0)]
    objs = [
        su._obj(0, 0, [0.0, 0.0, 0.0, 0.5]),          # the advisor's case: radius 0.5 at the origin
        su._obj(0, 0, [3.0, 0.25, 1.0, 0.05]),        # a tiny sphere
        su._obj(0, 0, [-4.0, 1.0, 2.0, -1.5]),        # hollow
        su._obj(0, 0, [0.0, -1000.0, 0.0, 1000.0]),   # the ground
        su._obj(6, 0, [1.0, 0.5, -2.0, 2.0, 0.5, -2.0, 1.0, 1.5, -2.0]),
        su._obj(6, 0, [-2.0, 0.1, 3.0, -1.99, 0.1, 3.0, -2.0, 0.11, 3.0]),  # a tiny triangle
        su._obj(1, 0, [-1.0, 2.0, 4.0, 1.0, 3.0, 4.0, 1.0]),
        su._obj(2, 0, [5.0, 0.75, -1.0, 6.0, 0.75, 1.0, 1.0]),
        su._obj(3, 0, [-6.0, 0.0, -1.0, -6.0, 2.0, 1.0, -1.0]),
        su._obj(4, 0, [2.0, 0.0, 2.0, 2.5, 0.5, 2.5]),
    ]
    rng = np.random.default_rng(77)
    for _ in range(40):
        c = rng.uniform(-8, 8, 3)
        c[1] = abs(c[1])
        objs.append(su._obj(0, 0, [*c, rng.uniform(0.1, 0.6)]))
    ps = su.pt.world_create([o[0] for o in objs], [m[0] for m in mats], [t[0] for t in texs], bvh_seed=5)
    os_ = su.orc.OracleScene.custom([o[1] for o in objs], [m[1] for m in mats], [t[1] for t in texs], [], [], bvh_seed=5)
    targets = np.array([[0.0, 0.0, 0.0], [3.0, 0.25, 1.0], [-4.0, 1.0, 2.0], [1.3, 0.8, -2.0], [-1.997, 0.103, 3.0], [0.0, 2.5, 4.0],
                        [5.5, 0.75, 0.0], [-6.0, 1.0, 0.0], [2.25, 0.25, 2.25]])
    return ps, os_, targets


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_far_origins_match_the_oracle(gpu_ctx, oracle, mode):
    """Origins 1e3 .. 1e25 away from the scene, aimed at primitives and just beside them, with small nonzero off-axis
    direction components (seen from 1e9 away a sphere of radius 0.5 'hits' rays that pass whole units beside
    it -- the discriminant cancels -- and the reference, which tests it whenever its parent's box is crossed, takes that
    hit).  The ordered walk must send such rays down the literal walk; every mode must equal the oracle bit for bit."""
    s, o, targets = _far_world()
    gpu_ctx.set_option("trace_mode", mode)
    try:
        gpu_ctx.upload(s)
    finally:
        gpu_ctx.set_option("trace_mode", -1)
    rng = np.random.default_rng(5)
    n = 64 * 600
    tgt = targets[rng.integers(0, len(targets), n)]
    beside = rng.random(n) < 0.6
    tgt = tgt + np.where(beside[:, None], rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 1, (n, 1)), 0.0)
    axis = rng.integers(0, 3, n)
    sign = rng.choice([-1.0, 1.0], n)
    dist = 10.0 ** rng.uniform(3, 25, n)
    d = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-14, -1, (n, 1))  # small off-axis components, never zero
    d[np.arange(n), axis] = sign
    ro = tgt - d * dist[:, None]
    rd = d * 10.0 ** rng.uniform(-2, 2, (n, 1))
    # the advisor's example verbatim, and neighbours
    ro[:4] = [[0.6, 0.0, -1e9], [0.6, 0.0, -1e9], [0.0, 0.6, -1e12], [2.0, 0.0, -1e9]]
    rd[:4] = [[1e-12, 1e-12, 1.0], [0.0, 1e-12, 1.0], [1e-12, 1e-12, 1.0], [1e-12, 1e-12, 1.0]]
    with np.errstate(all="ignore"):
        prim, t, normal, front = o.trace(ro, rd)
    assert 0.05 < (prim >= 0).mean() < 0.999
    assert_hits_equal(gpu_ctx.trace_batch(ro, rd), prim, t, normal, front)
    with np.errstate(all="ignore"):
        prim, t, normal, front = o.trace(ro[:8000], rd[:8000], -1e30, 1e30)
    assert_hits_equal(gpu_ctx.trace_batch(ro[:8000], rd[:8000], -1e30, 1e30), prim, t, normal, front)


def test_ordered_walk_still_takes_ordinary_rays(gpu_ctx, pt):
    """The admission rule must not send a scene's ordinary rays to the literal walk: SphereField rendered on the ordered
    walk has to stay several times faster than on the literal per-lane walk (it is 3x at the time of writing)."""
    w, h, spp, depth = 320, 180, 8, 20
    s = pt.Scene("SphereField").set_image(w, h, spp, depth, (1, 1, 1)).build()
    cam = s.camera(0.1)
    ms = {}
    rays = {}
    for mode in (2, 3):
        gpu_ctx.set_option("trace_mode", mode)
        try:
            gpu_ctx.upload(s)
        finally:
            gpu_ctx.set_option("trace_mode", -1)
        gpu_ctx.accum_reset(w, h)
        gpu_ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, spp, depth, 1)  # warm-up (allocations)
        gpu_ctx.accum_reset(w, h)
        st = gpu_ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, spp, depth, 1)
        ms[mode], rays[mode] = st.ms_trace, st.rays
    assert rays[2] == rays[3]
    assert ms[3] < 0.7 * ms[2], f"ordered walk {ms[3]:.2f} ms vs literal {ms[2]:.2f} ms: rays are being sent to the literal walk"


# ---------------------------------------------------------------- OBJ / MTL through the device (obj_model.rs:19-111)
OBJ_TEXT = """# two objects, quads, an n-gon, negative indices, three materials, per-vertex normals on one of them
mtllib model.mtl
o pyramid
v 0 0 0
v 2 0 0
v 2 0 2
v 0 0 2
v 1 2 1
vn 0 -1 0
vn 0.7 0.7 0
usemtl glass
f 1//1 2//1 3//1 4//1
usemtl mirror
f 1 5 2
f 2 5 3
usemtl paint
f 3 5 4
f -2 -1 -5
o slab
v -3 0.5 -1
v -1 0.5 -1
v -1 0.5 1
v -2 0.5 2
v -3 0.5 1
g top
f 6 10 9 8 7
"""
MTL_TEXT = """newmtl glass
illum 7
Ni 1.5
Kd 1 1 1
newmtl mirror
illum 5
Kd 0.8 0.8 0.9
Ns 50
newmtl paint
illum 2
Kd 0.7 0.2 0.2
"""


def test_obj_file_world_matches_oracle_on_the_gpu(gpu_ctx, pt, oracle, tmp_path):
    """ModelTestScene { path } (scene.rs:597-658): OBJ + MTL on disk -> the product's reader -> triangles, MTL materials and
    the nested BvhNode -> flattened tables on the device.  pt_trace_batch and a small render against the oracle fed the
    same tobj-shaped arrays (the reader itself is pinned on the CPU in test_host.py; tobj parity is unpinned)."""
    (tmp_path / "model.obj").write_text(OBJ_TEXT)
    (tmp_path / "model.mtl").write_text(MTL_TEXT)
    w, h, spp, depth, bg = 48, 48, 4, 20, (1, 1, 1)
    s = pt.Scene("ModelTest").set_seeds(1, 2, 1).set_model_path(str(tmp_path / "model.obj")).set_image(w, h, spp, depth, bg).build()
    meshes, mtls = s.meshes(), s.mtls()
    assert len(meshes) >= 4 and len(mtls) == 3
    assert sum(m.n_indices for m in meshes) // 3 == 2 + 1 + 1 + 1 + 1 + 3  # quad, 4 triangles, pentagon fan
    om, ol = su.mesh_to_oracle(meshes, mtls)
    o = su.orc.OracleScene.named("ModelTest", 1, 2, 1, om, ol)
    gpu_ctx.upload(s)
    cam = s.camera(0.0).as_array()
    ro, rd = su.camera_rays(cam, w, h, 30000, 3)
    p2, d2 = su.bounce_rays(o, ro, rd, 4)
    rng = np.random.default_rng(9)
    top_o = rng.uniform(-4, 4, (20000, 3))
    top_o[:, 1] = 6.0
    top_d = np.stack([rng.uniform(-0.3, 0.3, 20000), -np.ones(20000), rng.uniform(-0.3, 0.3, 20000)], axis=1)
    ro, rd = np.concatenate([ro, p2, top_o, top_o * [1, -1, 1]]), np.concatenate([rd, d2, top_d, -top_d])  # from above, and from below
    prim, t, normal, front = o.trace(ro, rd)
    n_tris = sum(m.n_indices for m in meshes) // 3
    assert len(np.unique(prim[prim >= 0])) >= n_tris - 1  # nearly every triangle and the ground are hit by some ray
    assert_hits_equal(gpu_ctx.trace_batch(ro, rd), prim, t, normal, front)
    stats, got, want, rays = render_pair(gpu_ctx, o, s.camera(0.0), w, h, bg, 0, spp, depth, 5)
    assert stats.rays == rays
    assert_images_close(got, want, spp)
    # the same world on the ordered walk (what a real, large OBJ takes)
    gpu_ctx.set_option("trace_mode", 3)
    try:
        gpu_ctx.upload(s)
    finally:
        gpu_ctx.set_option("trace_mode", -1)
    assert_hits_equal(gpu_ctx.trace_batch(ro, rd), prim, t, normal, front)


# ---------------------------------------------------------------- failure behaviour, statistics
def test_frame_loop_failure_raises_instead_of_terminating(pt, tmp_path):
    """a failure while a frame's PNG is still being written by the second thread (here: the directory cannot be
    created, so frame 1's write fails and the loop throws after frame 2's render) must surface as PtError, not unwind past
    a joinable std::thread (std::terminate)."""
    blocker = tmp_path / "file"
    blocker.write_text("not a directory")
    s = pt.Scene("SphereField").set_seeds(3, 3, 3).set_image(32, 16, 2, 4, (1, 1, 1)).build()
    with pytest.raises(pt.PtError):
        s.render_frames(str(blocker / "sub"), seed=3, first_frame=0, frame_stride=100)
    # and the scene is still usable afterwards
    info = s.render_frames(str(tmp_path), seed=3, first_frame=0, frame_stride=200)
    assert info["frames_written"] == 2


def test_frame_loop_reports_progress(pt, tmp_path):
    """main.rs:42-49: progress per frame -- here a callback (frames done, frames to do, frame index, seconds)."""
    import subprocess

    s = pt.Scene("SphereField").set_seeds(3, 3, 3).set_image(32, 16, 2, 4, (1, 1, 1)).build()
    seen = []
    info = s.render_frames(str(tmp_path), seed=3, first_frame=5, frame_stride=100, progress=lambda *a: seen.append(a))
    assert info["frames_written"] == 3
    assert [(d, n, f) for d, n, f, _ in seen] == [(1, 3, 5), (2, 3, 105), (3, 3, 205)] and all(sec > 0 for *_, sec in seen)
    seen.clear()
    s.render_frames(str(tmp_path), seed=3, first_frame=0, frame_stride=10, max_frames=2, progress=lambda *a: seen.append(a))
    assert [(d, n, f) for d, n, f, _ in seen] == [(1, 2, 0), (2, 2, 10)]
    exe = os.path.join(os.path.dirname(api.__file__), "bin", "ptrender")
    args = [exe, "SphereField", "--width", "32", "--height", "16", "--spp", "2", "--depth", "4", "--out", str(tmp_path / "cli"), "--frame-stride", "150"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "[1/2] frame 0000" in r.stderr and "[2/2] frame 0150" in r.stderr, r.stderr
    r = subprocess.run(args + ["--quiet"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "frame 0000" not in r.stderr


def test_busy_times_are_unions_of_launch_intervals(pt):
    w, h, spp, depth = 256, 144, 12, 12
    s = pt.Scene("CornellBox").set_image(w, h, spp, depth, (0, 0, 0)).build()
    ctx = pt.Context(0)
    try:
        ctx.upload(s)
        cam = s.camera(0.0)
        for lanes in (1, 3):
            ctx.set_option("lanes", lanes)
            ctx.accum_reset(w, h)
            ctx.render_accumulate(cam, w, h, (0, 0, 0), 0, spp, depth, 1)
            ctx.accum_reset(w, h)
            st = ctx.render_accumulate(cam, w, h, (0, 0, 0), 0, spp, depth, 1)
            assert 0 < st.ms_busy_trace <= st.ms_trace * 1.0001 and 0 < st.ms_busy_shade <= st.ms_shade * 1.0001
            assert st.ms_busy_all <= st.ms_total * 1.0001
            assert st.ms_busy_all >= max(st.ms_busy_trace, st.ms_busy_shade)
            if lanes == 1:  # one stream: launches cannot overlap, union = sum
                assert abs(st.ms_busy_trace - st.ms_trace) < 1e-3 and abs(st.ms_busy_shade - st.ms_shade) < 1e-3
        # profile = 0: same image, no kernel times
        ctx.set_option("lanes", 0)
        ctx.accum_reset(w, h)
        ctx.render_accumulate(cam, w, h, (0, 0, 0), 0, spp, depth, 1)
        a = ctx.accum_download()
        ctx.set_option("profile", 0)
        ctx.accum_reset(w, h)
        st = ctx.render_accumulate(cam, w, h, (0, 0, 0), 0, spp, depth, 1)
        assert np.array_equal(ctx.accum_download(), a)
        assert st.rays > 0 and st.ms_trace == 0 and st.ms_busy_all == 0 and st.ms_total > 0
    finally:
        ctx.close()


# ---------------------------------------------------------------- the guide tree built on the device
def test_device_guide_tree_builder(gpu_ctx, pt):
    """guide_build.hip (Morton sort + Karras hierarchy + bottom-up refit), verified on the host with the same invariants as the
    host builder's test: every box a leaf exactly once, inside every box stored above it -- scattered boxes, flat ones, hundreds
    of identical ones (equal Morton keys), a long thin row, huge next to tiny, two, a million."""
    rng = np.random.default_rng(4)

    def boxes(lo, size):
        return np.concatenate([lo, lo + size], axis=1)

    n = 5000
    cases = {
        "scattered": boxes(rng.uniform(-100, 100, (n, 3)), rng.uniform(0.01, 5, (n, 3))),
        "flat": boxes(rng.uniform(-10, 10, (n, 3)), rng.uniform(0, 2, (n, 3)) * (rng.integers(0, 3, (n, 1)) != np.arange(3))),
        "identical": np.tile([[1.0, 2.0, 3.0, 1.5, 2.5, 3.5]], (777, 1)),
        "row": boxes(np.stack([np.arange(n) * 1.0, np.zeros(n), np.zeros(n)], axis=1), np.full((n, 3), 0.5)),
        "scales": boxes(rng.uniform(-1, 1, (n, 3)) * 10.0 ** rng.integers(-6, 6, (n, 1)), 10.0 ** rng.uniform(-8, 5, (n, 3))),
        "two": np.array([[0, 0, 0, 1, 1, 1], [0, 0, 0, 1, 1, 1.0]]),
        "three": np.array([[0, 0, 0, 1, 1, 1], [5, 0, 0, 6, 1, 1.0], [0, 7, 0, 1, 8, 1.0]]),
        "million": boxes(rng.uniform(-300, 300, (1_000_000, 3)), rng.uniform(0.1, 1.0, (1_000_000, 3))),
    }
    for name, b in cases.items():
        nodes, depth = gpu_ctx.selftest_guide_tree_device(b)
        assert nodes == len(b) - 1, name
        assert depth <= 97, (name, depth)  # 64 key bits + 32 position bits
    assert gpu_ctx.selftest_guide_tree_device(cases["scattered"])[1] <= 40
    for bad in (np.array([[0, 0, 0, 1, 1, np.inf], [0, 0, 0, 1, 1, 1.0]]), np.array([[0, 0, 0, 1, 1, 1.0]]),
                np.array([[0, 0, 0, 1, 1, 1.0], [2, 0, 0, 1, 1, 1.0]])):
        with pytest.raises(pt.PtError):
            gpu_ctx.selftest_guide_tree_device(bad)


@pytest.mark.parametrize("world", ["SphereField", "mesh", "random"])
def test_device_built_guide_tree_gives_identical_results(pt, oracle, world):
    """Which builder made the guide tree must not show in any result: hits equal the oracle's bit for bit, and a render's
    accumulation buffer equals the host-built tree's bit for bit."""
    w, h, spp, depth, bg = 64, 36, 4, 20, (1, 1, 1)
    if world == "SphereField":
        s, o = su.named_pair("SphereField", 1, 1, 1, image=(w, h, spp, depth, bg))
        cam = s.camera(0.3)
    elif world == "mesh":
        s, o = su.model_pair(120, 61, 1.3, (0.0, 1.5, 0.0), image=(w, h, spp, depth, bg))
        cam = s.camera(0.0)
    else:
        s, o = su.random_world(31, n_objects=120, with_mesh=True)
        cam = api.camera_new((13.0, 6.0, 9.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, w / h, 0.0, 10.0)
    ro, rd = su.camera_rays(cam.as_array(), w, h, 40000, 3)
    p2, d2 = su.bounce_rays(o, ro, rd, 4)
    ro, rd = np.concatenate([ro, p2]), np.concatenate([rd, d2])
    prim, t, normal, front = o.trace(ro, rd)
    ctx = pt.Context(0)
    try:
        ctx.set_option("trace_mode", 3)
        out = {}
        for build in (0, 1):
            ctx.set_option("guide_build", build)
            ctx.upload(s)
            assert_hits_equal(ctx.trace_batch(ro, rd), prim, t, normal, front)
            ctx.accum_reset(w, h)
            st = ctx.render_accumulate(cam, w, h, bg, 0, spp, depth, 9)
            out[build] = (st.rays, ctx.accum_download())
        assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    finally:
        ctx.close()


def test_split_references_and_dealt_tail_do_not_change_results(pt, oracle):
    """guide_split references a thin diagonal triangle once per piece (the slivers around a mesh's poles) and tail_deal
    re-deals the rays before the per-wave tail: hits stay bit-equal to the oracle's -- also for rays aimed at the poles,
    where a ray meets the same triangle through several boxes -- and the accumulation buffer does not change by a bit."""
    w, h, spp, depth, bg = 64, 36, 4, 20, (1, 1, 1)
    s, o = su.model_pair(120, 61, 1.3, (0.0, 1.5, 0.0), image=(w, h, spp, depth, bg))
    cam = s.camera(0.0)
    ro, rd = su.camera_rays(cam.as_array(), w, h, 20000, 3)
    p2, d2 = su.bounce_rays(o, ro, rd, 4)
    rng = np.random.default_rng(5)
    # at the two poles of the mesh: x = z = center, from above / below / the side, through points within a sliver's length
    r_top, r_bottom = 1.3 * (1.0 + 0.07), 1.3 * (1.0 - 0.07)  # scene.cpp make_synthetic_mesh: radius * (1 + 0.07 cos(9 theta)) on the axis
    poles = np.array([[0.0, 1.5 + r_top, 0.0], [0.0, 1.5 - r_bottom, 0.0]])
    n = 6000
    target = poles[rng.integers(0, 2, n)] + rng.normal(size=(n, 3)) * np.array([0.02, 0.002, 0.02])
    origin = target + rng.normal(size=(n, 3)) * np.array([1.0, 0.3, 1.0]) + np.array([0.0, 0.0, 0.0])
    ro = np.concatenate([ro, p2, origin])
    rd = np.concatenate([rd, d2, target - origin])
    prim, t, normal, front = o.trace(ro, rd)
    assert (prim[-n:] >= 0).mean() > 0.3  # the pole rays do hit the mesh
    ctx = pt.Context(0)
    try:
        ctx.set_option("trace_mode", 3)
        ctx.set_option("tail_from", 2)
        out = []
        for split, deal, build in ((0, 0, 0), (4, 1, 0), (1, 1, 0), (4, 0, 1), (1000, 1, 0)):
            ctx.set_option("guide_split", split)
            ctx.set_option("tail_deal", deal)
            ctx.set_option("tail_waves_div", 4 if split == 1 else 1)  # (dealt over a quarter of the waves, once)
            ctx.set_option("guide_build", build)
            ctx.upload(s)
            assert_hits_equal(ctx.trace_batch(ro, rd), prim, t, normal, front)
            ctx.accum_reset(w, h)
            st = ctx.render_accumulate(cam, w, h, bg, 0, spp, depth, 9)
            out.append((st.rays, ctx.accum_download()))
        for rays, img in out[1:]:
            assert rays == out[0][0] and np.array_equal(img, out[0][1])
    finally:
        ctx.close()


def test_dealt_tail_with_fewer_waves_never_overruns_its_regions(pt):
    """tail_waves_div deals the tail's rays over fewer waves, i.e. into larger regions: in a closed box, where nearly every path
    is still alive at bounce 1, the regions of a third / a seventh / a 64th of the waves would not fit the streams, and the
    kernel must fall back to all waves.  Same accumulation buffer as without the deal, bit for bit."""
    w, h, spp, depth, bg = 96, 64, 8, 12, (0, 0, 0)
    s = pt.Scene("CornellBox").set_seeds(1, 1, 1).set_image(w, h, spp, depth, bg).build()
    cam = s.camera(0.0)
    ctx = pt.Context(0)
    try:
        ctx.set_option("trace_mode", 3)
        ctx.set_option("tail_from", 1)
        ctx.upload(s)
        out = []
        for deal, div in ((0, 1), (1, 1), (1, 2), (1, 3), (1, 7), (1, 64)):
            ctx.set_option("tail_deal", deal)
            ctx.set_option("tail_waves_div", div)
            ctx.accum_reset(w, h)
            st = ctx.render_accumulate(cam, w, h, bg, 0, spp, depth, 3)
            out.append((st.rays, ctx.accum_download()))
        assert out[0][0] > w * h * spp * 3  # paths do live on
        for rays, img in out[1:]:
            assert rays == out[0][0] and np.array_equal(img, out[0][1])
    finally:
        ctx.close()
