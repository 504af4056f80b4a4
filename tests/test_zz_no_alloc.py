"""pt_trace_batch_static (include/ptgpu.h): the hit query for callers that cannot allocate device memory.  Called through
the C ABI and compared with pt_trace_batch -- the entry point it is a variant of -- bit for bit, on two calls in a row, the
second with a smaller input (the library's scratch array keeps the first call's contents), on the lockstep walk and on the
ordered walk, from two contexts sharing the GPU's one array, and at the size limit."""
import numpy as np
import pytest

import scenes_util as su

pytestmark = pytest.mark.gpu


def _same(a, b):
    return a.tobytes() == b.tobytes()  # whole 40-byte records: primitive, front_face, t bits, normal bits


@pytest.mark.parametrize("name", ["CornellBox", "SphereField"])
def test_static_trace_equals_allocating_trace(pt, name):
    s = pt.Scene(name).set_seeds(1, 1, 1).build()
    ctx = pt.Context(0)
    try:
        ctx.upload(s)
        cam = s.camera(0.2).as_array()
        ro, rd = su.camera_rays(cam, 64, 36, 50000, 3)
        want = ctx.trace_batch(ro, rd)
        assert (want["prim"] >= 0).mean() > 0.3
        got = ctx.trace_batch_static(ro, rd)  # first call: 50000 rays
        assert _same(got, want)
        ro2, rd2 = su.camera_rays(cam, 64, 36, 777, 4)  # second call, smaller, other rays: nothing of the first may show
        want2 = ctx.trace_batch(ro2, rd2)
        got2 = ctx.trace_batch_static(ro2, rd2)
        assert _same(got2, want2)
        assert _same(ctx.trace_batch_static(ro2[:1], rd2[:1]), want2[:1])
        assert _same(ctx.trace_batch_static(ro, rd, 0.5, 900.0), ctx.trace_batch(ro, rd, 0.5, 900.0))
        assert len(ctx.trace_batch_static(np.zeros((0, 3)), np.zeros((0, 3)))) == 0
    finally:
        ctx.close()


def test_static_trace_limit_and_shared_array(pt):
    n_max = 65536  # PT_TRACE_STATIC_MAX_RAYS
    s = pt.Scene("TriangleTest").set_seeds(1, 1, 1).build()
    a, b = pt.Context(0), pt.Context(0)
    try:
        a.upload(s)
        b.upload(pt.Scene("LightTest").set_seeds(1, 1, 1).build())
        ro, rd = su.camera_rays(s.camera(0.0).as_array(), 64, 64, n_max + 1, 5)
        with pytest.raises(pt.PtError, match="PT_TRACE_STATIC_MAX_RAYS"):
            a.trace_batch_static(ro, rd)
        assert _same(a.trace_batch_static(ro[:n_max], rd[:n_max]), a.trace_batch(ro[:n_max], rd[:n_max]))  # exactly the limit
        # two contexts on one GPU share the GPU's one scratch array; used one after the other they do not disturb each other
        wa, wb = a.trace_batch(ro[:3000], rd[:3000]), b.trace_batch(ro[:2000], rd[:2000])
        assert _same(a.trace_batch_static(ro[:3000], rd[:3000]), wa)
        assert _same(b.trace_batch_static(ro[:2000], rd[:2000]), wb)
        assert _same(a.trace_batch_static(ro[:3000], rd[:3000]), wa)
        with pytest.raises(pt.PtError, match="no scene"):
            pt.Context(0).trace_batch_static(ro[:4], rd[:4])
    finally:
        a.close()
        b.close()
