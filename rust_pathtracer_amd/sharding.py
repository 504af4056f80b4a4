"""How work is split over GPUs (one process per GPU).

Samples of a pixel are i.i.d. (renderer.rs:21-28) and frames are independent given the shared world
(main.rs:51-64), so both shard with no data exchange during rendering:

* a static image: rank r of R renders sample indices [begin, begin+count) of EVERY pixel into its own
  width*height*3 f32 sum buffer; the counter-based sampler is keyed by the absolute sample index, so the
  union over ranks is exactly the set of samples a single GPU would draw.  One sum-reduce of the buffers
  (RCCL over xGMI via torch.distributed) to rank 0 follows, then rank 0 resolves.
* an animation: frame i goes to rank i mod R; no collective at all.
"""
from __future__ import annotations


def sample_range_for_rank(total_spp: int, rank: int, world_size: int) -> tuple[int, int]:
    """(begin, count) of the sample indices rank `rank` renders; ranges are disjoint and cover [0, total_spp)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    if total_spp < 0:
        raise ValueError("negative spp")
    base, extra = divmod(total_spp, world_size)
    begin = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return begin, count


def frames_for_rank(n_frames: int, rank: int, world_size: int) -> list[int]:
    """Frame indices rank `rank` renders: i with i mod world_size == rank."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    return list(range(rank, n_frames, world_size))
