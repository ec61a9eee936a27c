#!/usr/bin/env python3
"""Headline benchmark: rays/s of the 512^2 x 64-sample tri-plane render on MI355X (BASELINE.json).

A step = one pass of the hot path over one batch of 4 synthetic views per GPU (BASELINE config 2):
plane statistics + affines + NCHW->gather-layout pack (a4), then the fused render kernel (a2, a5-a12)
with in-kernel Philox jitter.  Inputs (raw planes, cameras, decoder weights) are resident in HBM
before the timed region.  With N>1 GPUs every rank renders its own 4 views per step (weak scaling,
no data-path collective) and the 3-channel raw frames are all-gathered over RCCL, as the
batch-of-views path does.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VIEWS_PER_GPU, R, D, PLANE = 4, 512, 64, 256
BYTES_PER_RAY_S1 = D * 1 * 1536 + 196        # S=1 (single-gather identity) -> 98 500 B/ray
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8 TB/s spec
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # written by tools/pmc.sh (rocprofv3 --pmc passes)


def measured_traffic():
    """HBM bytes per render_kernel launch from the committed rocprofv3 PMC passes of this same command
    (tools/pmc.sh): FETCH_SIZE*1024*2 (gfx950 counts 128-B requests at 64 B) + WRITE_SIZE*1024.  PMC counters
    cannot be collected from inside the timed process, so the figure comes from the profile file."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            d = json.load(f)
        return float(d["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def synth_inputs(torch, dev, seed):
    """Synthetic inputs of the config-2 shape: planes ~ N(0,1) with a per-channel mean/std spread
    (stands in for the random-init backbone output), random-init decoder (randn weights, zero bias),
    cameras on the gen_samples.py:166 yaw set at pitch -0.2, radius 2.7, pivot (0,0,0.2), fov 18.837."""
    from nerffaceediting_amd.camera_utils import FOV_to_intrinsics, LookAtPoseSampler
    g = torch.Generator(device="cpu").manual_seed(seed)
    planes = torch.randn(VIEWS_PER_GPU, 96, PLANE, PLANE, generator=g)
    planes = planes * torch.exp(0.5 * torch.randn(1, 96, 1, 1, generator=g)) + 0.7 * torch.randn(1, 96, 1, 1, generator=g)
    rng = np.random.RandomState(seed)                # FullyConnectedLayer init: randn weights, zero bias (networks_stylegan2.py:108-109)
    shapes = {"geo_net.0.weight": (64, 32), "geo_net.0.bias": (64,), "geo_net.2.weight": (16, 64), "geo_net.2.bias": (16,),
              "app_net.0.weight": (64, 32), "app_net.0.bias": (64,), "app_net.2.weight": (32, 64), "app_net.2.bias": (32,)}
    dec = {k: (rng.randn(*shp) if k.endswith("weight") else np.zeros(shp)).astype(np.float32) for k, shp in shapes.items()}
    yaws = [0.4, 0.0, -0.4, 0.2]
    pivot = torch.tensor([0.0, 0.0, 0.2])
    c2w = torch.cat([LookAtPoseSampler.sample(math.pi / 2 + y, math.pi / 2 - 0.2, pivot, radius=2.7) for y in yaws], 0).numpy()
    K = np.tile(FOV_to_intrinsics(18.837).numpy()[None], (VIEWS_PER_GPU, 1, 1))
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    return planes.to(dev), {k: to(v) for k, v in dec.items()}, dec, to(c2w), to(K), planes.numpy(), c2w, K


def cpu_baseline(planes_np, dec_np, c2w, K, opts, seed):
    """oracle/render_oracle.c (the C port of the reference algorithm, OpenMP over rays) rebuilt for this
    host and timed on a bounded sample of the same workload: view 0 of the batch, all 512^2 rays x 64
    samples, same Philox jitter, every host thread OpenMP gives us."""
    from oracle import c_oracle
    from oracle import render_oracle as orc
    c_oracle.build(native=True)
    norm, denorm, _, _ = orc.synthesis_planes(planes_np[:1])
    o, d = orc.ray_sampler(c2w[:1], K[:1], R)
    u = orc.philox_uniform(R * R, D, seed, 0)[None]
    threads = c_oracle.max_threads()
    t0 = time.perf_counter()
    c_oracle.render(norm, denorm, dec_np, o, d, opts, u, threads=0)
    dt = time.perf_counter() - t0
    return {"value": R * R / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"1 of the {VIEWS_PER_GPU} views: {R * R} rays x {D} samples, oracle/render_oracle.c "
                      f"(gcc -O3 -march=native -fopenmp, {threads} threads), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from nerffaceediting_amd import ops, sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback path exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    seed = 1000 + rank
    planes, dec_t, dec_np, c2w_t, K_t, planes_np, c2w, K = synth_inputs(torch, dev, seed)
    opts = dict(depth_resolution=D, depth_resolution_importance=0, ray_start=2.25, ray_end=3.3, box_warp=1,
                disparity_space_sampling=False, clamp_mode="softplus")
    names = ["geo_net.0.weight", "geo_net.0.bias", "geo_net.2.weight", "geo_net.2.bias",
             "app_net.0.weight", "app_net.0.bias", "app_net.2.weight", "app_net.2.bias"]
    dec_packed = ops.decoder_pack(*[dec_t[k] for k in names])
    M = R * R
    n_total = world * VIEWS_PER_GPU
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i, timed):
        mean, std = ops.plane_stats(planes)                     # a4
        aff = ops.make_affine(mean, std)
        packed = ops.plane_pack(planes)
        if timed:
            ev[i][0].record()
        rgb, seg, depth, wsum = ops.render(packed, packed, dec_packed, opts, cam2world=c2w_t, intrinsics=K_t,
                                           resolution=R, affines=aff, seed=seed + i, channels_first=True)
        if timed:
            ev[i][1].record()
        if world > 1:                                           # frames of every rank, in view order
            frames = rgb[:, :3].reshape(VIEWS_PER_GPU, 3, R, R)
            sharding.all_gather_frames(frames, n_total)
        return rgb

    for i in range(args.warmup):
        step(i, False)

    def barrier():
        if world > 1:
            dist.barrier()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
        rays_per_step = n_total * M
        value = rays_per_step * args.steps / dt
        launch_bytes = VIEWS_PER_GPU * M * BYTES_PER_RAY_S1
        achieved = launch_bytes / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "rays/s, 512^2 x 64-sample tri-plane render", "value": value, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: render core (a2,a4-a12), 4 views/GPU/step, 512^2 rays x 64 "
                                   "stratified samples, 256^2x96 planes, fp32, Philox jitter",
                       "views_per_step": n_total, "views_per_s": n_total * args.steps / dt,
                       "rays_per_view": M, "depth_samples": D, "parallelism": f"views-dp{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(),
                         "kernel": "nfe::render_kernel<false,false>", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": launch_bytes,
                         "note": "achieved = logical gather bytes (S=1: 98500 B/ray) / kernel time; traffic = HBM bytes per "
                                 "launch from profiles/r01_pmc_traffic.json (planes stay L2/Infinity-Cache resident, so "
                                 "traffic << algorithmic bytes); the kernel is TA/VALU-bound, see DESIGN.md §6"},
        }
        if not args.no_cpu_baseline and world == 1:          # reported at N=1 only (host cores are shared by all ranks)
            out["cpu_baseline"] = cpu_baseline(planes_np, dec_np, c2w, K, opts, seed)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
