"""The N > 1 path on CPU: two processes, gloo, the same sharding + reduce code bench.py runs over RCCL.
The renderer plugged in here is the oracle (tests may use it); what is under test is the split and the exchange."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

W, H, SPP, DEPTH, SEED = 24, 16, 7, 12, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import oracle_api as orc
    from rust_pathtracer_amd.distributed import render_animation_sharded, render_static_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = orc.OracleScene.named("CornellBox", 1, 1, 1)
        cam = orc.scene_camera("CornellBox", 0.0, W / H)
        buf = torch.zeros((H, W, 3), dtype=torch.float32)
        rays = torch.zeros(1, dtype=torch.int64)

        def accumulate(begin, count):
            a, r = sc.render_accumulate(cam, W, H, (0, 0, 0), begin, count, DEPTH, SEED, threads=1)
            buf.add_(torch.from_numpy(a.astype(np.float32)))
            rays.add_(r)

        rng = render_static_sharded(accumulate, buf, SPP, rank, world, dist)
        dist.all_reduce(rays)
        frames = render_animation_sharded(lambda i: i * i, 9, rank, world)
        np.save(os.path.join(out_dir, f"frames{rank}.npy"), np.array(sorted(frames.items())))
        np.save(os.path.join(out_dir, f"range{rank}.npy"), np.array(rng))
        if rank == 0:
            np.save(os.path.join(out_dir, "sum.npy"), buf.numpy())
            np.save(os.path.join(out_dir, "rays.npy"), rays.numpy())
    finally:
        dist.destroy_process_group()


def test_sample_sharding_and_reduce_world_size_2(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sc = oracle.OracleScene.named("CornellBox", 1, 1, 1)
    cam = oracle.scene_camera("CornellBox", 0.0, W / H)
    want, rays = sc.render_accumulate(cam, W, H, (0, 0, 0), 0, SPP, DEPTH, SEED, threads=1)
    got = np.load(tmp_path / "sum.npy")
    assert int(np.load(tmp_path / "rays.npy")[0]) == rays  # the union of the ranks' samples is the single-process set
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5)     # equal up to the f32 reduce order
    r0, r1 = np.load(tmp_path / "range0.npy"), np.load(tmp_path / "range1.npy")
    assert r0.tolist() == [0, 4] and r1.tolist() == [4, 3]
    f0, f1 = np.load(tmp_path / "frames0.npy"), np.load(tmp_path / "frames1.npy")
    assert f0[:, 0].tolist() == [0, 2, 4, 6, 8] and f1[:, 0].tolist() == [1, 3, 5, 7]
    assert (f0[:, 1] == f0[:, 0] ** 2).all()
