"""BASELINE.json's configurations at their full sizes on the GPU: exact hit-query counts and f32-tolerance images
against the oracle at 1-2 spp (the oracle finishes those in seconds on the box's cores), plus properties that do
not depend on the size: determinism, sample-range additivity (what multi-GPU sharding relies on), monotone
per-bounce ray counts."""
import numpy as np
import pytest

import scenes_util as su

pytestmark = pytest.mark.gpu


def _close(got, want):
    tol = 1e-4 * max(1.0, np.abs(want).max())  # f32 radiometry vs the oracle's f64
    assert np.abs(got - want).max() <= tol


def test_config2_cornell_1080p(gpu_ctx, oracle):
    w, h, depth, seed = 1920, 1080, 50, 1
    s, o = su.named_pair("CornellBox", 1, 1, 1, image=(w, h, 256, depth, (0, 0, 0)))
    gpu_ctx.upload(s)
    cam = s.camera(0.0)
    gpu_ctx.accum_reset(w, h)
    st = gpu_ctx.render_accumulate(cam, w, h, (0, 0, 0), 0, 2, depth, seed)
    got = gpu_ctx.accum_download().astype(np.float64)
    want, rays = o.render_accumulate(cam.as_array(), w, h, (0, 0, 0), 0, 2, depth, seed)
    assert st.rays == rays and st.paths == w * h * 2
    _close(got, want)
    # scene.rs S3: at 16:9 the rays outside the box opening miss everything -> black side bands
    assert not got[:, :200].any() and not got[:, -200:].any() and got[:, 800:1100].mean() > 0.1
    # more samples: per-bounce ray counts never grow, the run is reproducible, sample ranges add up
    gpu_ctx.accum_reset(w, h)
    st8 = gpu_ctx.render_accumulate(cam, w, h, (0, 0, 0), 0, 8, depth, seed)
    a8 = gpu_ctx.accum_download()
    rpb = list(st8.rays_per_bounce)[:depth]
    assert rpb[0] == w * h * 8 and all(x >= y for x, y in zip(rpb, rpb[1:])) and sum(rpb) == st8.rays
    gpu_ctx.accum_reset(w, h)
    st8b = gpu_ctx.render_accumulate(cam, w, h, (0, 0, 0), 0, 8, depth, seed)
    assert st8b.rays == st8.rays and np.array_equal(gpu_ctx.accum_download(), a8)
    gpu_ctx.accum_reset(w, h)
    ra = gpu_ctx.render_accumulate(cam, w, h, (0, 0, 0), 0, 3, depth, seed).rays
    rb = gpu_ctx.render_accumulate(cam, w, h, (0, 0, 0), 3, 5, depth, seed).rays
    assert ra + rb == st8.rays and np.allclose(gpu_ctx.accum_download(), a8, rtol=1e-5, atol=1e-5)


def test_config3_million_triangles(gpu_ctx, oracle, pt):
    # ModelTest with a synthetic closed mesh of 1,000,000 triangles, world x100
    s = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0)
    w, h, depth = 480, 270, 50
    s.set_image(w, h, 64, depth, (1, 1, 1)).build()
    assert s.desc.n_prims == 1_000_001 and s.desc.n_nodes == 1_000_000
    om, ol = su.mesh_to_oracle(s.meshes(), s.mtls())
    ground = su._obj(0, 0, [0.0, -1000.0 * 100.0, 0.0, 1000.0 * 100.0])[1]
    texs = [su._tex(0, color=(0.2, 0.3, 0.1))[1], su._tex(0, color=(0.9, 0.9, 0.9))[1], su._tex(1, odd=0, even=1)[1]]
    o = oracle.OracleScene.custom([ground, su._obj(7, 0, [])[1]], [su._mat(0, 2)[1]], texs, om, ol, bvh_seed=1)
    assert o.num_prims == 1_000_001
    gpu_ctx.upload(s)
    cam = s.camera(0.0)
    ro, rd = su.camera_rays(cam.as_array(), w, h, 60000, 1)
    p2, d2 = su.bounce_rays(o, ro, rd, 2)
    ro, rd = np.concatenate([ro, p2]), np.concatenate([rd, d2])
    prim, t, normal, front = o.trace(ro, rd)
    hits = gpu_ctx.trace_batch(ro, rd)
    assert (prim >= 1).mean() > 0.05  # the mesh subtends ~12 degrees of the 60 x 92 degree view
    assert np.array_equal(hits["prim"], prim)
    hh = prim >= 0
    assert np.array_equal(hits["t"][hh], t[hh]) and np.array_equal(hits["normal"][hh], normal[hh])
    gpu_ctx.accum_reset(w, h)
    st = gpu_ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, 2, depth, 3)
    want, rays = o.render_accumulate(cam.as_array(), w, h, (1, 1, 1), 0, 2, depth, 3)
    assert st.rays == rays
    _close(gpu_ctx.accum_download().astype(np.float64), want)
    # At BASELINE.json's full image size (1920x1080), where the oracle would take minutes: what must not depend on how the frame
    # is scheduled -- the tail's deal, split references in the guide tree, who built the tree -- does not: same ray count per
    # bounce and the same accumulation buffer bit for bit; and the two halves of the sample range sum to the whole.
    W, H = 1920, 1080
    ctx = pt.Context(0)
    try:
        out = []
        for split, deal, build in ((4, 1, 0), (0, 0, 0), (4, 1, 1)):
            ctx.set_option("guide_split", split)
            ctx.set_option("tail_deal", deal)
            ctx.set_option("guide_build", build)
            ctx.upload(s)
            ctx.accum_reset(W, H)
            st = ctx.render_accumulate(cam, W, H, (1, 1, 1), 0, 2, depth, 3)
            out.append(([int(v) for v in st.rays_per_bounce[:50]], ctx.accum_download()))
        for per_bounce, img in out[1:]:
            assert per_bounce == out[0][0] and np.array_equal(img, out[0][1])
        assert out[0][0][0] == W * H * 2 and out[0][0][49] > 0  # every primary ray is a segment; some paths live all 50 bounces
        ctx.accum_reset(W, H)
        ra = ctx.render_accumulate(cam, W, H, (1, 1, 1), 0, 1, depth, 3).rays
        rb = ctx.render_accumulate(cam, W, H, (1, 1, 1), 1, 1, depth, 3).rays
        assert ra + rb == sum(out[0][0]) and np.allclose(ctx.accum_download(), out[0][1], rtol=1e-5, atol=1e-5)
    finally:
        ctx.close()


def test_config4_lighttest_4k_sample_sharding(gpu_ctx, oracle):
    w, h, depth, seed = 3840, 2160, 50, 1
    s, o = su.named_pair("LightTest", 1, 1, 1, image=(w, h, 1024, depth, (0, 0, 0)))
    gpu_ctx.upload(s)
    cam = s.camera(0.0)
    gpu_ctx.accum_reset(w, h)
    st = gpu_ctx.render_accumulate(cam, w, h, (0, 0, 0), 0, 1, depth, seed)
    got = gpu_ctx.accum_download().astype(np.float64)
    want, rays = o.render_accumulate(cam.as_array(), w, h, (0, 0, 0), 0, 1, depth, seed)
    assert st.rays == rays
    _close(got, want)
    # 8 "GPUs" x 1 sample each on ranges [g, g+1) must sum to the single-device 8-sample buffer (<= 1e-5 relative)
    gpu_ctx.accum_reset(w, h)
    whole = gpu_ctx.render_accumulate(cam, w, h, (0, 0, 0), 0, 8, depth, seed).rays
    a_whole = gpu_ctx.accum_download()
    parts = np.zeros_like(a_whole)
    rays_parts = 0
    for g in range(8):
        gpu_ctx.accum_reset(w, h)
        rays_parts += gpu_ctx.render_accumulate(cam, w, h, (0, 0, 0), g, 1, depth, seed).rays
        parts += gpu_ctx.accum_download()
    assert rays_parts == whole
    assert np.allclose(parts, a_whole, rtol=1e-5, atol=1e-6)


def test_config5_spherefield_frames(gpu_ctx, oracle):
    w, h, depth = 1920, 1080, 50
    s, o = su.named_pair("SphereField", 1, 1, 1, image=(w, h, 128, depth, (1, 1, 1)))
    gpu_ctx.upload(s)  # the world is built once; only the camera moves (main.rs:23,51-64)
    for frame in (0, 37):
        cam = s.camera(frame / 120.0)
        assert np.array_equal(cam.as_array(), oracle.scene_camera("SphereField", frame / 120.0, w / h))
        gpu_ctx.accum_reset(w, h)
        st = gpu_ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, 1, depth, 5)
        want, rays = o.render_accumulate(cam.as_array(), w, h, (1, 1, 1), 0, 1, depth, 5)
        assert st.rays == rays
        _close(gpu_ctx.accum_download().astype(np.float64), want)
