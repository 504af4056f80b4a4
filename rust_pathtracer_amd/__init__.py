"""MI355X-native path-tracing backend with the capabilities of ArcticXWolf/rust-pathtracer.

The product is two native libraries built by ``make -C rust_pathtracer_amd``:

* ``lib/libptgpu.so``  -- hand-written HIP kernels for gfx950 behind the C ABI of ``include/ptgpu.h``
* ``lib/libpthost.so`` -- C++17 mirror of the reference's public surface (``include/pthost.h``)

This package is a thin ctypes binding over them (plumbing for tests, bench and torch.distributed); it
contains no rendering code of its own and no CPU fallback: importing :mod:`rust_pathtracer_amd.api`
raises if the libraries are missing.
"""
from .api import (  # noqa: F401
    Context,
    PtCamera,
    PtHitOut,
    PtMaterial,
    PtNode,
    PtPrim,
    PtSceneDesc,
    PtStats,
    PtTexture,
    PtError,
    Scene,
    accum_reduce,
    camera_new,
    write_png,
    selftest_guide_tree,
    selftest_split_triangle,
    device_count,
    libptgpu,
    libpthost,
    world_create,
)
from .sharding import sample_range_for_rank, frames_for_rank  # noqa: F401
