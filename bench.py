#!/usr/bin/env python3
"""Benchmark of the render path: Mrays/s and s/frame on the configurations BASELINE.json names.

  python bench.py [--config C] [--gpus N] [--steps K] [--warmup W]            (N GPUs driven from ONE process)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N ...                                                (one process per GPU)

--config (default 2, the headline):
  1  TwoSpheres 400x225, 100 spp, depth 50 on the CPU renderer (the oracle: plumbing, no GPU); image written
  2  CornellBox 1920x1080, 256 spp, depth 50
  3  ModelTest with a synthetic closed mesh of 1,000,000 triangles (world x100), 1920x1080, 64 spp, depth 50
  4  LightTest 3840x2160, 1024 spp, depth 50; samples sharded over the GPUs, one RCCL reduce per frame
  5  SphereField animation, 120 frames at 1920x1080, 128 spp, depth 50; frame i on GPU i mod N, world rebuilt and
     uploaded before every frame (timed apart)
--scene/--width/--height/--spp/--depth/--frames override the configuration's values.

A step = one whole frame (configs 2-4: accumulate this GPU's sample range through raygen -> [trace -> shade] x depth ->
fold, sum-reduce the per-GPU accumulation buffers to GPU 0 when N > 1, resolve to RGB8) or one whole animation (config 5).
The scene is built and uploaded before the timed region (the reference builds its world once per run, main.rs:23) except
where config 5 asks for the rebuild; no input crosses PCIe inside the timed region -- rays are generated on the device --
and the RGB8 frame is copied back by resolve, as the reference returns a Vec<u8>.

N > 1 without torchrun: this one process creates a context per GPU, drives each from its own host thread (the C ABI's
contexts are independent; ctypes releases the GIL during a call) and sums the buffers with pt_accum_reduce (RCCL over
xGMI).  Under torchrun every rank drives one GPU and the reduce is torch.distributed's (the same RCCL).  Either way
strong scaling: the frame is fixed, per-GPU work shrinks with N.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_TRACE = 60       # per ray: read o,d 48 B + write t,prim 12 B              
BYTES_SHADE_IN = 76    # per ray: ray 48 + hit 12 + state 16
BYTES_SHADE_OUT = 64   # per surviving ray: next ray 48 + state 16
BYTES_SEGMENT = 200    # whole pipeline per ray segment with separate trace and shade kernels
BYTES_PATH_IN = 64     # fused / tail kernel, per ray: ray 48 + state 16 (the hit record stays in registers)
# Instruction-issue roofline (profiles/r01_instruction_rates.txt, measured on this chip): one wave64 f64 or integer-multiply
# VALU instruction issues per 4.2 cycles per SIMD at full occupancy (78.6 TFLOP/s f64 = 4 cycles), 4 SIMDs per CU.
CYCLES_PER_VALU = 4.2
SIMDS_PER_CU = 4

CONFIGS = {
    1: dict(scene="TwoSpheres", width=400, height=225, spp=100, depth=50, cpu_only=True),
    2: dict(scene="CornellBox", width=1920, height=1080, spp=256, depth=50),
    3: dict(scene="ModelTest", width=1920, height=1080, spp=64, depth=50, synthetic=(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0)),
    4: dict(scene="LightTest", width=3840, height=2160, spp=1024, depth=50),
    5: dict(scene="SphereField", width=1920, height=1080, spp=128, depth=50, frames=120, rebuild=True),
}
BACKGROUNDS = {"CornellBox": (0.0, 0.0, 0.0), "TriangleTest": (0.0, 0.0, 0.0), "LightTest": (0.0, 0.0, 0.0),
               "TwoSpheres": (1.0, 1.0, 1.0), "SphereField": (1.0, 1.0, 1.0), "ModelTest": (1.0, 1.0, 1.0)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--frames", type=int, default=-1, help="animation: frames per step (config 5: 120); 0 = a static image")
    ap.add_argument("--no-rebuild", action="store_true", help="config 5: build and upload the world once, as the reference does")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--paths-per-pass", type=int, default=0, help="0 = library default")
    ap.add_argument("--repack-every", type=int, default=-1, help="-1 = library default")
    ap.add_argument("--repack-first", type=int, default=-1)
    ap.add_argument("--lanes", type=int, default=0, help="0 = library default")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (repeatable)")
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="0 = library default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated", action="store_true", help="skip the extra untimed lanes=1 frame")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--out-image", default=None, help="write the last frame as a PNG (config 1 always writes one)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="one process per GPU: torch.distributed backend of the reduce (nccl = RCCL over xGMI).  gloo + --devices 0,0 rehearses the "
                    "per-process code path on a box with one GPU (RCCL refuses two ranks on one device); the line then says so")
    ap.add_argument("--devices", default=None, help="comma-separated device index per worker (default 0..N-1).  Listing a device twice "
                    "rehearses the N-worker code path on a box with fewer GPUs: RCCL cannot reduce between two contexts of one device, "
                    "so the buffers are then summed through the host and the line says so")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    overridden = False
    for key, val in (("scene", args.scene), ("width", args.width), ("height", args.height), ("spp", args.spp), ("depth", args.depth)):
        if val:
            overridden = overridden or cfg[key] != val
            cfg[key] = val
    if args.scene and args.scene != CONFIGS[args.config]["scene"]:
        cfg.pop("synthetic", None)
        cfg.pop("frames", None)
        cfg.pop("rebuild", None)
    if args.frames >= 0:
        overridden = overridden or cfg.get("frames", 0) != args.frames
        cfg["frames"] = args.frames
    if args.no_rebuild:
        cfg["rebuild"] = False
    cfg["background"] = BACKGROUNDS[cfg["scene"]]
    cfg["label"] = (f"{cfg['scene']} {cfg['width']}x{cfg['height']} {cfg['spp']}spp depth {cfg['depth']}"
                    + (f", {cfg['frames']} frames" if cfg.get("frames") else "")
                    + (", 1M-triangle synthetic mesh" if cfg.get("synthetic") else "")
                    + ("" if overridden else f" (BASELINE.json configs[{args.config - 1}])"))
    return args, cfg


# ---------------------------------------------------------------------------------------------------------- CPU baseline
def oracle_scene_for(cfg, product_scene=None):
    """The oracle's world for a configuration (tests/oracle_api.py; CPU restatement of the reference, test infrastructure)."""
    import oracle_api as orc

    if cfg.get("synthetic"):
        import scenes_util as su  # the scaled mesh world is a custom world on the oracle's side (as in tests/test_gpu_fullsize.py)
        import rust_pathtracer_amd as pt

        n_lon, n_lat, radius, center, scale = cfg["synthetic"]
        s = product_scene or pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(n_lon, n_lat, radius, center, scale).build()
        om, ol = su.mesh_to_oracle(s.meshes(), s.mtls())
        ground = su._obj(0, 0, [0.0, -1000.0 * scale, 0.0, 1000.0 * scale])[1]
        texs = [su._tex(0, color=(0.2, 0.3, 0.1))[1], su._tex(0, color=(0.9, 0.9, 0.9))[1], su._tex(1, odd=0, even=1)[1]]
        o = orc.OracleScene.custom([ground, su._obj(7, 0, [])[1]], [su._mat(0, 2)[1]], texs, om, ol, bvh_seed=1)
        o._keep_scene = s
        return o, True
    return orc.OracleScene.named(cfg["scene"], 1, 1, 1), False


def cpu_baseline(cfg, seed, target_seconds, rays_per_frame, product_scene=None):
    """The oracle (a C++ port of the reference's CPU renderer, NOT the Rust binary) timed on this box's host cores on a
    bounded sample of the same workload: same scene, depth and aspect at 480x270, spp chosen for ~target_seconds; plus the
    single-thread rate on a third of that time, and the frame time both extrapolate to."""
    import oracle_api as orc

    threads = min(16, os.cpu_count() or 1)  # the GPU box's CPU share for one GPU (the box caps worker pools at 16)
    w, h = 480, 270
    sc, custom = oracle_scene_for(cfg, product_scene)
    if custom:
        import rust_pathtracer_amd as pt

        tmp = pt.Scene("ModelTest").set_seeds(1, 1, 1)
        tmp.set_model_synthetic(12, 7, cfg["synthetic"][2], cfg["synthetic"][3], cfg["synthetic"][4])  # only for the camera at this aspect
        tmp.set_image(w, h, 1, cfg["depth"], cfg["background"]).build()
        cam = tmp.camera(0.0).as_array()
    else:
        cam = orc.scene_camera(cfg["scene"], 0.0, w / h)
    bg, depth = cfg["background"], cfg["depth"]
    t0 = time.perf_counter()
    _, rays1 = sc.render_accumulate(cam, w, h, bg, 0, 1, depth, seed, threads)
    dt1 = time.perf_counter() - t0
    spp = int(max(1, min(4096, target_seconds / max(dt1, 1e-3))))
    t0 = time.perf_counter()
    _, rays = sc.render_accumulate(cam, w, h, bg, 0, spp, depth, seed, threads)
    dt = time.perf_counter() - t0
    mrays = rays / dt / 1e6
    spp1 = max(1, spp // (3 * threads))
    t0 = time.perf_counter()
    _, rays_1t = sc.render_accumulate(cam, w, h, bg, 0, spp1, depth, seed, 1)
    dt_1t = time.perf_counter() - t0
    mrays_1t = rays_1t / dt_1t / 1e6
    return {
        "value": round(mrays, 3),
        "unit": "Mrays/s",
        "cores": threads,
        "cores_available": os.cpu_count(),
        "kind": "port",
        "sample": f"{cfg['scene']} {w}x{h}, {spp} spp, depth {depth}: {rays} rays in {dt:.2f} s "
                  f"(C++ restatement of rust-pathtracer, not the Rust binary)",
        "single_thread_mrays": round(mrays_1t, 3),
        "single_thread_sample": f"{spp1} spp: {rays_1t} rays in {dt_1t:.2f} s on 1 thread",
        "extrapolated_s_per_frame": round(rays_per_frame / (mrays * 1e6), 1) if rays_per_frame else None,
        "extrapolated_s_per_frame_single_thread": round(rays_per_frame / (mrays_1t * 1e6), 1) if rays_per_frame else None,
    }


def run_config1(args, cfg):
    """BASELINE.json configs[0]: the CPU renderer alone (the oracle), image written, ray count reported."""
    import numpy as np
    import oracle_api as orc
    import rust_pathtracer_amd as pt  # only for the PNG writer (host library; no GPU is touched)

    threads = min(16, os.cpu_count() or 1)
    w, h, spp, depth, bg = cfg["width"], cfg["height"], cfg["spp"], cfg["depth"], cfg["background"]
    sc = orc.OracleScene.named(cfg["scene"], 1, 1, 1)
    cam = orc.scene_camera(cfg["scene"], 0.0, w / h)
    t0 = time.perf_counter()
    acc, rays = sc.render_accumulate(cam, w, h, bg, 0, spp, depth, args.seed, threads)
    dt = time.perf_counter() - t0
    img = orc.resolve(acc, spp)
    out = args.out_image or os.path.join(ROOT, "gpurun_out", "config1_twospheres.png")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    pt.write_png(out, np.ascontiguousarray(img))
    mrays = rays / dt / 1e6
    line = {
        "metric": f"Mrays/s (ray segments = hit queries per second), {cfg['scene']} {h}p {spp}spp on the CPU renderer; ms_per_step = s/frame*1e3",
        "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": 0, "steps": 1, "warmup": 0, "ms_per_step": round(dt * 1e3, 1),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": f"synthetic (scene.rs {cfg['scene']} constants, seeded BVH and Perlin table)",
        "config": {"workload": cfg["label"], "parallelism": f"CPU, {threads} threads", "rays_per_frame": int(rays), "image": out},
        "roofline": None,
        "cpu_baseline": {"value": round(mrays, 3), "unit": "Mrays/s", "cores": threads, "cores_available": os.cpu_count(), "kind": "port",
                         "sample": f"the whole configuration: {rays} rays in {dt:.2f} s (C++ restatement of rust-pathtracer, not the Rust binary)"},
    }
    print(json.dumps(line))


# ---------------------------------------------------------------------------------------------------------- one GPU's work
class Worker:
    """One GPU: its context, the uploaded scene, and this GPU's share of a step."""

    def __init__(self, pt, args, cfg, device, rank, world):
        self.pt, self.cfg, self.device, self.rank, self.world, self.seed = pt, cfg, device, rank, world, args.seed
        self.ctx = pt.Context(device)
        if args.paths_per_pass:
            self.ctx.set_option("paths_per_pass", args.paths_per_pass)
        if args.blocks_per_cu:
            self.ctx.set_option("blocks_per_cu", args.blocks_per_cu)
        if args.lanes:
            self.ctx.set_option("lanes", args.lanes)
        for kv in args.opt:
            k, v = kv.split("=")
            self.ctx.set_option(k, int(v))
        if args.repack_every >= 0:
            self.ctx.set_option("repack_every", args.repack_every)
        if args.repack_first >= 0:
            self.ctx.set_option("repack_first", args.repack_first)
        self.scene = self.make_scene().build()
        self.ctx.upload(self.scene)
        self.ctx.accum_reset(cfg["width"], cfg["height"])
        self.stats = []          # PtStats of this step's render calls
        self.img = None
        self.seconds_build = self.seconds_upload = 0.0

    def make_scene(self):
        c = self.cfg
        s = self.pt.Scene(c["scene"]).set_seeds(1, 1, 1)
        if c.get("synthetic"):
            s.set_model_synthetic(*c["synthetic"])
        return s.set_image(c["width"], c["height"], c["spp"], c["depth"], c["background"])

    def render_samples(self, begin, count, t=0.0):
        c = self.cfg
        st = self.ctx.render_accumulate(self.scene.camera(t), c["width"], c["height"], c["background"], begin, count, c["depth"], self.seed)
        self.stats.append(st)

    def static_share(self):
        """This GPU's sample range of the one frame (accumulation buffer zeroed first)."""
        c = self.cfg
        self.stats = []
        self.ctx.accum_reset(c["width"], c["height"])
        begin, count = self.pt.sample_range_for_rank(c["spp"], self.rank, self.world)
        if count:
            self.render_samples(begin, count)

    def animation_share(self):
        """This GPU's frames of the animation: frame i on GPU i mod N (no collective); each frame resolved to RGB8."""
        c = self.cfg
        self.stats = []
        self.seconds_build = self.seconds_upload = 0.0
        n = c["frames"]
        for i in self.pt.frames_for_rank(n, self.rank, self.world):
            if c.get("rebuild"):  # BASELINE.json configs[4]: BvhNode::new and the upload again, same seeds => the same world
                t0 = time.perf_counter()
                self.scene = self.make_scene().build()
                t1 = time.perf_counter()
                self.ctx.upload(self.scene)
                self.seconds_build += t1 - t0
                self.seconds_upload += time.perf_counter() - t1
            self.ctx.accum_reset(c["width"], c["height"])
            self.render_samples(0, c["spp"], i / n)  # main.rs:52: t = frame / frames
            self.img = self.ctx.resolve(c["spp"])

    def close(self):
        self.ctx.close()


def sum_stats(stats_list):
    """Sums the PtStats of several render calls.  Every call's rays are split between the kernels that traced them: bounces
    [0, tail_from) ran as one trace and one shade launch each, bounces [tail_from, depth) inside the fused path / tail launch
    (PtStats.tail_from; = depth when no such launch ran)."""
    keys = ("rays", "ms_total", "ms_raygen", "ms_trace", "ms_shade", "ms_fold", "ms_repack", "ms_path", "ms_busy_trace", "ms_busy_shade",
            "ms_busy_path", "ms_busy_all", "launches_trace", "launches_shade", "launches_path")
    out = {k: 0 for k in keys}
    out.update(rays_wavefront=0, survivors_wavefront=0, rays_tail=0, survivors_tail=0, tail_from=None)
    for st in stats_list:
        for k in keys:
            out[k] += getattr(st, k)
        rpb = [int(x) for x in st.rays_per_bounce]
        tf = min(int(st.tail_from), len(rpb) - 1)  # (bounces >= 63 are folded into the last slot: with depth <= 63 nothing is)
        out["rays_wavefront"] += sum(rpb[:tf])                # rays entering the separate trace + shade launches
        out["survivors_wavefront"] += sum(rpb[1:tf + 1])      # what their shade launches wrote: the rays entering bounces 1..tail_from
        out["rays_tail"] += sum(rpb[tf:])                     # rays entering a bounce inside the tail launch
        out["survivors_tail"] += sum(rpb[tf + 1:])            # what the tail launch wrote back between its bounces
        out["tail_from"] = tf if out["tail_from"] is None else min(out["tail_from"], tf)
    return out


def kernel_entries(tot, pmc, rays_per_s_whole, n_cus, clock_hz):
    """Per-kernel roofline entries from the summed PtStats of the timed region (rank / GPU 0): algorithmic bytes of the rays each
    kernel really processed over the sum of its launch durations."""
    rw, sw, rt, stl = tot["rays_wavefront"], tot["survivors_wavefront"], tot["rays_tail"], tot["survivors_tail"]
    table = [("trace_kernel", tot["ms_trace"], tot["ms_busy_trace"], tot["launches_trace"], rw * BYTES_TRACE, rw),
             ("shade_kernel", tot["ms_shade"], tot["ms_busy_shade"], tot["launches_shade"], rw * BYTES_SHADE_IN + sw * BYTES_SHADE_OUT, rw),
             # fused path / tail launch: per ray entering one of its bounces 64 B in (ray 48 + state 16; the hit record stays in
             # registers), per survivor 64 B out
             ("path_kernel", tot["ms_path"], tot["ms_busy_path"], tot["launches_path"], rt * BYTES_PATH_IN + stl * BYTES_SHADE_OUT, rt)]
    out = []
    for name, ms_sum, ms_busy, launches, nbytes, rays in table:
        if launches == 0 or ms_sum <= 0:
            continue
        achieved = nbytes / (ms_sum * 1e-3) / 1e9
        e = {"kernel": name, "launches": launches, "rays": int(rays), "launch_ms_sum": round(ms_sum, 3), "busy_ms": round(ms_busy, 3),
             "avg_launch_ms": round(ms_sum / launches, 5), "bytes_per_launch": int(nbytes / launches), "achieved": round(achieved, 2),
             "frac": round(achieved / HBM_PEAK_GBS, 5),
             "achieved_while_busy": round(nbytes / (ms_busy * 1e-3) / 1e9, 2) if ms_busy > 0 else None}
        p = pmc.get(name)
        if p and p.get("hbm_bytes_per_ray") is not None:
            e["traffic"] = int(p["hbm_bytes_per_ray"] * rays / launches)
            e["traffic_source"] = pmc.get("_source", "stored PMC")
        out.append(e)
    return out


def issue_roofline(pmc, tot, mrays_per_gpu, n_cus, clock_hz):
    """The bound the pipeline actually runs against: VALU instruction issue.  Wave-instructions per 64 ray segments from the PMC
    counters (SQ_INSTS_VALU / rays * 64 of each kernel, weighted by the rays that kernel traced in THIS run: trace + shade for the
    bounces launched separately, the tail launch for the rest), CYCLES_PER_VALU cycles each per SIMD.  Raygen, fold and resolve
    are not in it (a few per cent of a frame's instructions)."""
    try:
        rw, rt = tot["rays_wavefront"], tot["rays_tail"]
        valu = (pmc["trace_kernel"]["valu_per_64_rays"] + pmc["shade_kernel"]["valu_per_64_rays"]) * rw
        if rt:
            valu += pmc["path_kernel"]["valu_per_64_rays"] * rt
        valu /= (rw + rt)
    except (KeyError, TypeError, ZeroDivisionError):
        return None
    peak = 64.0 * n_cus * SIMDS_PER_CU * clock_hz / (valu * CYCLES_PER_VALU) / 1e6  # Mrays/s
    return {"bound": "valu_issue", "achieved": round(mrays_per_gpu, 1), "peak": round(peak, 1), "unit": "Mrays/s",
            "frac": round(mrays_per_gpu / peak, 4), "valu_per_64_segments": round(valu, 1), "cycles_per_valu": CYCLES_PER_VALU,
            "simds": n_cus * SIMDS_PER_CU, "clock_mhz": round(clock_hz / 1e6),
            "source": pmc.get("_source", "stored PMC") + "; issue rate from profiles/r01_instruction_rates.txt"}


def load_pmc(args_config, overridden):
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if overridden or not os.path.exists(path):
        return {}
    try:
        d = json.load(open(path))
    except Exception:
        return {}
    return d.get(f"config{args_config}", d if args_config == 2 and "trace_kernel" in d else {})  # (round 1's file was config 2 alone)


# ---------------------------------------------------------------------------------------------------------- main
def main():
    args, cfg = parse_args()
    if cfg.get("cpu_only"):
        return run_config1(args, cfg)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    per_process = env_world > 1          # launched by torch.distributed.run: one process per GPU
    if per_process and env_world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env_world}")
    world = args.gpus
    in_process = 1 if per_process else world  # GPUs this process drives

    import numpy as np
    import torch

    import rust_pathtracer_amd as pt
    from rust_pathtracer_amd.distributed import render_static_sharded

    dist = None
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(world))
    if len(devices) != world:
        raise SystemExit(f"--devices lists {len(devices)} devices for --gpus {world}")
    if per_process:
        import torch.distributed as dist

        local_rank = devices[rank] if args.devices else local_rank  # (the device this rank drives)
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if not per_process:
        if max(devices) >= pt.device_count():
            raise SystemExit(f"--gpus {world} needs devices {devices} but only {pt.device_count()} visible")
    rehearsal = len(set(devices)) < len(devices)

    animation = bool(cfg.get("frames"))
    w, h, spp, depth = cfg["width"], cfg["height"], cfg["spp"], cfg["depth"]
    if per_process:
        workers = [Worker(pt, args, cfg, local_rank, rank, world)]
    else:
        workers = [Worker(pt, args, cfg, devices[g], g, world) for g in range(world)]
    root = workers[0]
    pool = ThreadPoolExecutor(max_workers=in_process) if in_process > 1 else None
    accum = None
    if per_process and not animation:
        accum = torch.zeros((h, w, 3), dtype=torch.float32, device=f"cuda:{local_rank}")
        torch.cuda.synchronize()
        root.ctx.accum_bind(accum.data_ptr())

    def everyone(fn_name):
        if pool is None:
            getattr(root, fn_name)()
        else:
            for f in [pool.submit(getattr(wk, fn_name)) for wk in workers]:
                f.result()

    def step():
        if animation:
            everyone("animation_share")
            return
        if per_process:
            accum.zero_()  # (the bound tensor is the accumulation buffer; accum_reset would clear the internal one)
            torch.cuda.synchronize()
            root.stats = []
            # this rank's sample range, then ONE reduce to rank 0 (RCCL over xGMI, on torch's stream): the function that
            # tests/test_distributed_cpu.py runs over gloo
            render_static_sharded(root.render_samples, accum, spp, rank, world, dist)
            torch.cuda.synchronize()
        else:
            everyone("static_share")
            if world > 1 and not rehearsal:
                pt.accum_reduce([wk.ctx for wk in workers])  # RCCL over xGMI, on the contexts' streams; returns when drained
            elif world > 1:  # rehearsal on shared devices only: through the host
                total = workers[0].ctx.accum_download()
                for wk in workers[1:]:
                    total = total + wk.ctx.accum_download()
                root.ctx.accum_upload(total)
        if rank == 0:
            root.img = root.ctx.resolve(spp)

    def fence():
        for g in ([local_rank] if per_process else sorted(set(devices))):
            torch.cuda.synchronize(g)
        if per_process:
            dist.barrier()
            torch.cuda.synchronize()

    # Kernel timing (two HIP events around every launch) costs ~1 ms per frame: nothing against a 137 ms frame, 6 % of an 8-GPU
    # share of it.  On one GPU every timed step is profiled; on N > 1 only GPU 0's first timed step is (still inside the timed
    # region), the rest of the run carries no events.
    profile_all = world == 1
    if not profile_all:
        for wk in workers:
            wk.ctx.set_option("profile", 0)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats_root, rays_local = [], 0
    seconds_build = seconds_upload = 0.0
    profiled_steps = 0
    for k in range(args.steps):
        profiled = profile_all or (k == 0 and rank == 0)
        if not profile_all:
            root.ctx.set_option("profile", 1 if profiled else 0)
        step()
        if profiled:
            stats_root += root.stats
            profiled_steps += 1
        rays_local += sum(st.rays for wk in workers for st in wk.stats)
        seconds_build += max(wk.seconds_build for wk in workers)
        seconds_upload += max(wk.seconds_upload for wk in workers)
    fence()
    elapsed = time.perf_counter() - t0

    # one more, untimed, frame on a single stream: the same kernels without the overlap between lanes, for an isolated
    # per-kernel roofline next to the co-run one of the timed region
    iso = None
    if world == 1 and not args.lanes and not args.no_isolated and not animation:
        root.ctx.set_option("lanes", 1)
        root.static_share()
        iso = sum_stats(root.stats)
        root.ctx.set_option("lanes", 0)  # the library default (automatic)
    profiled_steps = max(1, profiled_steps)

    if per_process:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays_local], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total = int(r.item())
    else:
        rays_total = rays_local

    if rank == 0:
        props = torch.cuda.get_device_properties(local_rank if per_process else devices[0])
        n_cus = props.multi_processor_count
        clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays_total / elapsed / 1e6
        frames_per_step = cfg["frames"] if animation else 1
        overridden = "BASELINE.json" not in cfg["label"]
        pmc = load_pmc(args.config, overridden)
        tot = sum_stats(stats_root)
        kernels = kernel_entries(tot, pmc, mrays * 1e6, n_cus, clock_hz)
        # the dominant kernel: the one with the larger total launch time IN THE TIMED REGION
        dominant = max(kernels, key=lambda e: e["launch_ms_sum"]) if kernels else None
        roofline = None
        if dominant is not None:
            with_bytes = dominant if "achieved" in dominant else max((e for e in kernels if "achieved" in e), key=lambda e: e["launch_ms_sum"])
            roofline = {"bound": "hbm", "kernel": with_bytes["kernel"], "achieved": with_bytes["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": with_bytes["frac"], "traffic": with_bytes.get("traffic"), "traffic_source": with_bytes.get("traffic_source"),
                        "avg_launch_ms": with_bytes["avg_launch_ms"], "bytes_per_launch": with_bytes["bytes_per_launch"],
                        "dominant_by": "largest sum of launch durations in the timed region (GPU 0)",
                        "profiled_steps": profiled_steps,
                        "tail_from": tot["tail_from"],  # bounces before it: trace_kernel + shade_kernel; from it on: path_kernel
                        "kernels": kernels,
                        "pipeline_frac": round(mrays * 1e6 * BYTES_SEGMENT / world / (HBM_PEAK_GBS * 1e9), 5)}
            if iso is not None:
                roofline["isolated"] = {"note": "one untimed frame with lanes=1 (no overlap between sub-passes)",
                                        "kernels": kernel_entries(iso, pmc, 0, n_cus, clock_hz)}
        out = {
            "metric": f"Mrays/s (ray segments = hit queries per second), {cfg['scene']} {h}p {spp}spp; ms_per_step = s/frame*1e3"
                      + (f" x {frames_per_step} frames" if animation else ""),
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
            "data": f"synthetic (scene.rs {cfg['scene']} constants, seeded BVH, device-generated rays)",
            "config": {"workload": cfg["label"],
                       "parallelism": ("single GPU" if world == 1 else
                                       (f"frames mod {world}" if animation else f"spp-sharded x{world}, one RCCL reduce per frame")
                                       + (", one process per GPU" + ("" if animation else f" (torch.distributed reduce, backend {args.dist_backend})") if per_process
                                          else ", one process, one host thread per GPU" + ("" if animation else " (pt_accum_reduce)"))
                                       + (" -- REHEARSAL: workers share a GPU, buffers summed through the host, not a multi-GPU measurement" if rehearsal else "")
                                       + (" -- gloo: the reduce goes through the host, not xGMI" if per_process and args.dist_backend != "nccl" and not rehearsal else "")),
                       "rays_per_frame": rays_total // (args.steps * frames_per_step)},
            "roofline": roofline,
            "issue_roofline": issue_roofline(pmc, tot, mrays / world, n_cus, clock_hz),
            # union of each kernel's launch intervals on GPU 0 (with several lanes launches overlap: the sums of launch
            # durations, roofline.kernels[].launch_ms_sum, can exceed the frame)
            "kernel_busy_ms_per_step": {"trace": round(tot["ms_busy_trace"] / profiled_steps, 3), "shade": round(tot["ms_busy_shade"] / profiled_steps, 3),
                                        "path": round(tot["ms_busy_path"] / profiled_steps, 3), "any_kernel": round(tot["ms_busy_all"] / profiled_steps, 3),
                                        "raygen_sum": round(tot["ms_raygen"] / profiled_steps, 3), "fold_sum": round(tot["ms_fold"] / profiled_steps, 3),
                                        "render_call": round(tot["ms_total"] / profiled_steps, 3)},
        }
        if root.stats:
            out["rays_per_bounce_last_call"] = [int(x) for x in list(root.stats[-1].rays_per_bounce)[:min(depth, 64)]]
        if animation:
            out["ms_per_frame"] = round(ms_per_step / frames_per_step, 3)
            out["frames_per_s"] = round(frames_per_step * args.steps / elapsed, 3)
            out["rebuild_ms_per_frame"] = {"build": round(seconds_build / args.steps / max(1, len(pt.frames_for_rank(cfg["frames"], 0, world))) * 1e3, 3),
                                           "upload": round(seconds_upload / args.steps / max(1, len(pt.frames_for_rank(cfg["frames"], 0, world))) * 1e3, 3),
                                           "note": "BvhNode::new + flatten / record stream + guide tree + copies, per frame on the slowest GPU; inside ms_per_step"
                                                   if cfg.get("rebuild") else "world built and uploaded once (not rebuilt)"}
        if args.out_image and root.img is not None:
            pt.write_png(args.out_image, root.img)
        if not args.no_cpu_baseline:
            # the same host cores whatever N is: with N > 1 a shorter sample (the line must still finish within minutes)
            seconds = args.cpu_seconds if world == 1 else min(args.cpu_seconds, 5.0)
            out["cpu_baseline"] = cpu_baseline(cfg, args.seed, seconds, rays_total // (args.steps * frames_per_step), root.scene)
        print(json.dumps(out))
    for wk in workers:
        wk.close()
    if pool is not None:
        pool.shutdown()
    if per_process:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
