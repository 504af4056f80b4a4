#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and s/frame, CornellBox 1920x1080, 256 spp, depth 50 (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one whole frame: accumulate 256 samples per pixel at depth 50 through the wavefront pipeline
(raygen -> [trace -> shade/compact] x depth -> fold), sum-reduce the per-GPU accumulation buffers to rank 0
(N > 1), resolve to RGB8.  The scene is built and uploaded before the timed region (the reference builds its
world once per run, main.rs:23); no input crosses PCIe inside it -- rays are generated on the device -- and the
6.2 MB RGB8 frame is copied back by resolve, as the reference returns a Vec<u8>.
N > 1: one process per GPU, samples sharded (strong scaling: the frame is fixed), RCCL reduce via torch.distributed.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_TRACE = 60       # per ray: read o,d 48 B + write t,prim 12 B              
BYTES_SHADE_IN = 76    # per ray: ray 48 + hit 12 + state 16
BYTES_SHADE_OUT = 64   # per surviving ray: next ray 48 + state 16
BYTES_SEGMENT = 200    # whole pipeline per ray segment with separate trace and shade kernels
BYTES_PATH_IN = 64     # fused path kernel, per ray: ray 48 + state 16 (the hit record stays in registers)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="CornellBox")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--paths-per-pass", type=int, default=0, help="0 = library default")
    ap.add_argument("--repack-every", type=int, default=-1, help="-1 = library default")
    ap.add_argument("--repack-first", type=int, default=-1)
    ap.add_argument("--lanes", type=int, default=0, help="0 = library default")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (repeatable)")
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="0 = library default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    return ap.parse_args()


BACKGROUNDS = {"CornellBox": (0.0, 0.0, 0.0), "TriangleTest": (0.0, 0.0, 0.0), "LightTest": (0.0, 0.0, 0.0),
               "TwoSpheres": (1.0, 1.0, 1.0), "SphereField": (1.0, 1.0, 1.0), "ModelTest": (1.0, 1.0, 1.0)}


def cpu_baseline(args, target_seconds):
    """The oracle (a C++ port of the reference's CPU renderer, NOT the Rust binary) timed on this box's host cores
    on a bounded sample of the same workload: same scene, depth and aspect at 480x270, spp chosen for ~15 s."""
    import oracle_api as orc

    threads = min(16, os.cpu_count() or 1)  # the GPU box's CPU share for one GPU
    w, h = 480, 270
    sc = orc.OracleScene.named(args.scene, 1, 1, 1)
    cam = orc.scene_camera(args.scene, 0.0, w / h)
    bg = BACKGROUNDS[args.scene]
    t0 = time.perf_counter()
    _, rays1 = sc.render_accumulate(cam, w, h, bg, 0, 1, args.depth, args.seed, threads)
    dt1 = time.perf_counter() - t0
    spp = int(max(1, min(4096, target_seconds / max(dt1, 1e-3))))
    t0 = time.perf_counter()
    _, rays = sc.render_accumulate(cam, w, h, bg, 0, spp, args.depth, args.seed, threads)
    dt = time.perf_counter() - t0
    mrays = rays / dt / 1e6
    return {
        "value": round(mrays, 3),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{args.scene} {w}x{h}, {spp} spp, depth {args.depth}: {rays} rays in {dt:.2f} s "
                  f"(C++ restatement of rust-pathtracer, not the Rust binary)",
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with python -m torch.distributed.run --nproc-per-node N")

    import numpy as np
    import torch

    import rust_pathtracer_amd as pt
    from rust_pathtracer_amd import sample_range_for_rank
    from rust_pathtracer_amd.distributed import render_static_sharded

    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    w, h, spp, depth = args.width, args.height, args.spp, args.depth
    bg = BACKGROUNDS[args.scene]
    scene = pt.Scene(args.scene).set_seeds(1, 1, 1).set_image(w, h, spp, depth, bg).build()
    cam = scene.camera(0.0)
    ctx = pt.Context(local_rank)
    if args.paths_per_pass:
        ctx.set_option("paths_per_pass", args.paths_per_pass)
    if args.blocks_per_cu:
        ctx.set_option("blocks_per_cu", args.blocks_per_cu)
    if args.lanes:
        ctx.set_option("lanes", args.lanes)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    if args.repack_every >= 0:
        ctx.set_option("repack_every", args.repack_every)
    if args.repack_first >= 0:
        ctx.set_option("repack_first", args.repack_first)
    ctx.upload(scene)
    ctx.accum_reset(w, h)
    accum = None
    if world > 1:
        accum = torch.zeros((h, w, 3), dtype=torch.float32, device=f"cuda:{local_rank}")
        torch.cuda.synchronize()
        ctx.accum_bind(accum.data_ptr())
    s_begin, s_count = sample_range_for_rank(spp, rank, world)

    def step():
        if world > 1:
            accum.zero_()
            torch.cuda.synchronize()
        else:
            ctx.accum_reset(w, h)
        box = [pt.PtStats()]

        def accumulate(begin, count):
            box[0] = ctx.render_accumulate(cam, w, h, bg, begin, count, depth, args.seed)

        render_static_sharded(accumulate, accum, spp, rank, world, dist)  # this rank's samples, then ONE reduce to rank 0
        st = box[0]
        if world > 1:
            torch.cuda.synchronize()
        img = ctx.resolve(spp) if rank == 0 else None
        return st, img

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays = 0
    ms_trace = ms_shade = ms_raygen = ms_fold = ms_repack = ms_path = 0.0
    n_trace = n_shade = n_path = 0
    shade_in = shade_out = 0
    img = None
    for _ in range(args.steps):
        st, img = step()
        rays += st.rays
        ms_trace += st.ms_trace
        ms_shade += st.ms_shade
        ms_raygen += st.ms_raygen
        ms_fold += st.ms_fold
        ms_repack += st.ms_repack
        ms_path += st.ms_path
        n_path += st.launches_path
        n_trace += st.launches_trace
        n_shade += st.launches_shade
        rpb = list(st.rays_per_bounce)
        shade_in += st.rays
        shade_out += st.rays - rpb[0] if st.rays else 0  # rays entering bounces >= 1 are shade's survivors
    fence()
    elapsed = time.perf_counter() - t0

    # one more, untimed, frame on a single stream: the same kernels without the overlap between lanes, for an
    # isolated per-kernel roofline next to the co-run one of the timed region
    iso = None
    if world == 1 and not args.lanes:
        ctx.set_option("lanes", 1)
        iso, _ = step()
        ctx.set_option("lanes", 0)  # the library default (automatic)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total = int(r.item())
    else:
        rays_total = rays

    if rank == 0:
        headline = (args.scene, w, h, spp, depth) == ("CornellBox", 1920, 1080, 256, 50)
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays_total / elapsed / 1e6
        # roofline of the dominant kernel on rank 0, from HIP-event times on the library's stream
        if ms_path > 0:
            # the fused kernel keeps the hit record in registers: it has to move 64 B per ray in and 64 B per survivor out
            kname, kms, kbytes, klaunches = "path_kernel", ms_path, shade_in * BYTES_PATH_IN + shade_out * BYTES_SHADE_OUT, n_path
        # which of the two is the dominant kernel: by their times alone on the chip when that frame was measured (with several
        # sub-passes sharing the chip a launch's duration also counts the time it waited for its share), else by the timed region
        elif (iso.ms_trace >= iso.ms_shade) if iso is not None else (ms_trace >= ms_shade):
            kname, kms, kbytes, klaunches = "trace_kernel", ms_trace, rays * BYTES_TRACE, n_trace
        else:
            kname, kms, kbytes, klaunches = "shade_kernel", ms_shade, shade_in * BYTES_SHADE_IN + shade_out * BYTES_SHADE_OUT, n_shade
        achieved = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        # HBM bytes from the PMC counters (profiles/pmc_summary.json, collected by profiles/collect_pmc.sh on this
        # scene at 16 spp and corrected as MI355X_MICROARCH.md prescribes): stored per ray, scaled to this launch size
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc) and args.scene == "CornellBox":
            try:
                per_ray = json.load(open(pmc)).get(kname, {}).get("hbm_bytes_per_ray")
                if per_ray is not None:
                    traffic = int(per_ray * rays / max(1, klaunches))
            except Exception:
                traffic = None
        isolated = None
        if iso is not None:
            if kname == "path_kernel":
                ims, il = iso.ms_path, iso.launches_path
                ibytes = iso.rays * BYTES_PATH_IN + (iso.rays - list(iso.rays_per_bounce)[0]) * BYTES_SHADE_OUT
            elif kname == "trace_kernel":
                ims, ibytes, il = iso.ms_trace, iso.rays * BYTES_TRACE, iso.launches_trace
            else:
                ims = iso.ms_shade
                ibytes = iso.rays * BYTES_SHADE_IN + (iso.rays - list(iso.rays_per_bounce)[0]) * BYTES_SHADE_OUT
                il = iso.launches_shade
            ia = ibytes / (ims * 1e-3) / 1e9 if ims > 0 else 0.0
            isolated = {"achieved": round(ia, 2), "frac": round(ia / HBM_PEAK_GBS, 5), "avg_launch_ms": round(ims / max(1, il), 5),
                        "kernel_ms_per_step": {"raygen": round(iso.ms_raygen, 3), "trace": round(iso.ms_trace, 3),
                                               "shade": round(iso.ms_shade, 3), "fold": round(iso.ms_fold, 3), "path": round(iso.ms_path, 3)},
                        "note": "one untimed frame with lanes=1 (no overlap between sub-passes)"}
        out = {
            "metric": f"Mrays/s (ray segments = hit queries per second), {args.scene} {h}p {spp}spp; ms_per_step = s/frame*1e3",
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64 geometry / f32 radiometry",
            "data": f"synthetic (scene.rs {args.scene} constants, seeded BVH, device-generated rays)",
            "config": {"workload": f"{args.scene} {w}x{h} {spp}spp depth {depth}" + (" (BASELINE.json configs[1])" if headline else ""),
                       "parallelism": f"spp-sharded x{world}" if world > 1 else "single GPU",
                       "rays_per_frame": rays_total // args.steps},
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "avg_launch_ms": round(kms / max(1, klaunches), 5),
                "bytes_per_launch": int(kbytes / max(1, klaunches)),
                "pipeline_frac": round(mrays * 1e6 * BYTES_SEGMENT / world / (HBM_PEAK_GBS * 1e9), 5),
                "isolated": isolated,
            },
            "kernel_ms_per_step": {"raygen": round(ms_raygen / args.steps, 3), "trace": round(ms_trace / args.steps, 3),
                                   "shade": round(ms_shade / args.steps, 3), "fold": round(ms_fold / args.steps, 3),
                                   "repack": round(ms_repack / args.steps, 3), "path": round(ms_path / args.steps, 3)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
