"""Multi-GPU glue: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI) for the one exchange.

A static image is sharded by samples: every rank holds the whole scene, renders its own sample range of every
pixel into a width*height*3 f32 buffer, and the buffers are summed to rank 0 with ONE reduce (24.9 MB at 1080p,
99.5 MB at 4K -- about a millisecond over xGMI against hundreds of milliseconds of rendering).  An animation is
sharded by frames and needs no collective.  The functions take the renderer as a callable so the same logic is
exercised by the CPU tests (gloo, world_size 2) and by bench.py on GPUs.
"""
from __future__ import annotations

from typing import Callable

from .sharding import frames_for_rank, sample_range_for_rank


def render_static_sharded(accumulate: Callable[[int, int], None], buffer, total_spp: int, rank: int, world_size: int, dist=None):
    """Renders this rank's samples with accumulate(sample_begin, sample_count) -- which adds into `buffer`, a torch
    tensor on this rank's device -- then sum-reduces `buffer` to rank 0.  Returns (sample_begin, sample_count)."""
    begin, count = sample_range_for_rank(total_spp, rank, world_size)
    if count > 0:
        accumulate(begin, count)
    if world_size > 1:
        reduce_to_root(buffer, dist)
    return begin, count


def reduce_to_root(buffer, dist):
    """Sum-reduces `buffer` to rank 0: ONE collective.  Backend "nccl" (= RCCL) reduces device tensors in place over xGMI; gloo
    reduces host tensors only, so a device tensor goes through the host there (CPU tests, and rehearsals on a box with one GPU)."""
    if buffer.is_cuda and dist.get_backend() == "gloo":
        host = buffer.cpu()
        dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
        if dist.get_rank() == 0:
            buffer.copy_(host)
    else:
        dist.reduce(buffer, dst=0, op=dist.ReduceOp.SUM)


def render_animation_sharded(render_frame: Callable[[int], object], n_frames: int, rank: int, world_size: int) -> dict:
    """Renders frames i = rank (mod world_size); returns {frame index: result of render_frame(i)}."""
    return {i: render_frame(i) for i in frames_for_rank(n_frames, rank, world_size)}
