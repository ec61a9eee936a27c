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
    python bench.py --gpus N ...            # no launcher: starts its own N ranks (children, before any GPU call), like
                                            # the reference's train.py:98-103 spawn; rank 0's line is forwarded
Prints ONE JSON line on rank 0.

The default line also carries `strong_scaling`: the 512-frame orbit job of BASELINE config 4 (full synthesis, frames sharded
over the ranks, chunked uint8 all-gather overlapped with rendering), run once after the timed region, so a 1/2/4/8-GPU sweep
of the default command measures both the weak-scaling render metric (`value`) and the strong-scaling views/s.

Other workloads of SURVEY.md section 8(d) (never the default; same JSON contract, lines kept in profiles/):
    --workload full     config 3: full synthesis (a1-a14), 8 views/GPU/step, 512^2 x 64 render, bf16 convs -> views/s
    --workload orbit    config 4: 512 (latent, camera) pairs sharded over the ranks (strong scaling), one all-gather
    --workload twopass  config 5: render core, D=96 + 96 importance samples, dual plane sets, 512^2 -> rays/s
    --workload ffhq     the reference's own inference configuration (train.py:306-307): full synthesis, 128^2 x (48+48) render,
                        fp32-grade (split-bf16) convs, 4 views/GPU/step -> views/s
    --workload editstep plane-editing step: 128^2 x (48+48) dual-plane render + nfe_render_backward -> rays/s
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

import bench_roofline as rl

VIEWS_PER_GPU, R, D, PLANE = 4, 512, 64, 256
BYTES_PER_RAY_S1 = D * 1 * 1536 + 196        # S=1 (single-gather identity) -> 98 500 B/ray
HBM_PEAK_GBS = rl.HBM_PEAK_GBS               # MI355X_MICROARCH.md: 8 TB/s spec
# Per-launch hardware counters of every kernel a roofline block names: rocprofv3 --pmc passes of THESE commands on the shipped build
# (tools/pmc.sh -> tools/pmc_summary.py -> issue_floor.json, committed under profiles/; PMC cannot be collected inside the timed
# process).  All roofline arithmetic lives in bench_roofline.py: one formula per number, the same for every workload.
PMC = {"render": "r06_issue_floor.json", "render_fp32": "r06_issue_floor_fp32.json",
       "twopass_final": "r06_issue_floor_twopass_final.json", "twopass_sigma": "r06_issue_floor_twopass_sigma.json",
       "twopass_importance": "r06_issue_floor_twopass_importance.json"}                 # tools/r06_profile.sh
# split-bf16 decoder: 3 MFMAs per product, and the geometry head's second layer runs a 32-row M block for 16 rows (8 192 padded of the
# 7 168 algorithmic MACs per sample): matrix work issued per algorithmic flop
DECODER_MFMA_WORK = 3.0 * 8192.0 / 7168.0


def backward_roofline(bwd_ms, samples, logical_gbs):
    """Roofline block of the edit step.  The dominant kernel by HBM traffic is the accumulate pass of the binned scatter: it reads every
    256-byte feature-gradient row once per plane (3 x 1.6 GB per 4 views), its records and index list; with its tile in registers
    (round 3) it runs at the read ceiling of the machine (tools/microbench/read_bw.hip: 6.3 TB/s for a read-only stream).  The
    committed counter file (tools/pmc.sh, PMC_KERNEL=bwd_accumulate_reg) gives that kernel's traffic and duration for the same launch."""
    import json
    import os
    rec = None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_backward_counters.json")
    if os.path.exists(path):
        rec = json.load(open(path))
    ach = rec["hbm_bytes_per_launch"] / rec["avg_ns_profiled"] if rec else None            # GB/s
    alg = samples * (3 * 256 + 3 * 28)                                                      # rows + records + index, per launch
    return {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS if rec else None,
            "traffic": rec["hbm_bytes_per_launch"] if rec else None, "algorithmic_bytes_per_launch": alg,
            "kernel": "nfe::bwd_accumulate_reg_kernel (of: color_dot_kernel, bwd_ray_kernel, bwd_decoder_kernel, bwd_bin_fill_kernel, "
                      "bwd_accumulate_reg_kernel)",
            # the longest kernel of the backward is not HBM-bound: its counter-measured unit fractions (same file, same box, same command)
            "decoder_kernel": ({k: rec["decoder_kernel"][k] for k in ("kernel", "avg_ns_trace", "single_wave_kernel_avg_ns_trace", "valu_active", "ta_busy",
                                                                      "mfma_busy", "lds_issue", "hbm_bytes_per_launch")} if rec else None),
            "kernel_ms": rec["avg_ns_profiled"] * 1e-6 if rec else None, "backward_ms": bwd_ms,
            "hbm_bytes_per_sample_model": 1240, "hbm_model_gbs": samples * 1240 / (bwd_ms * 1e-3) / 1e9,
            "logical_gather_scatter_gbs": logical_gbs,
            "note": "achieved / traffic / kernel_ms: the accumulate pass of one 4-view launch from profiles/r06_backward_counters.json "
                    "(FETCH_SIZE corrected as the guide prescribes + WRITE_SIZE; duration = rocprofv3 kernel trace of the both-sets mode, which "
                    "agrees with the duration under the counters to a few per cent), not re-measured by this run; peak = the 8 TB/s HBM figure (a "
                    "read-only stream reaches 6.3 TB/s on this machine, tools/microbench/read_bw.hip, so the kernel is at 0.87 of what reads "
                    "can get); backward_ms is this run's HIP-event time of the whole backward.  Per sample "
                    "the decoder-backward kernel (wave-specialised since round 5: decoder_kernel holds its busy fractions - vector ALU, texture addresser, matrix pipe - none of them a bound) writes a 256-byte feature-gradient row and three 32-byte records, the fill pass sorts a 4-byte "
                    "index per record, the accumulate pass (one wave owns an 8x8 texel tile in registers) reads index, record and row once per "
                    "plane; the forward keeps the decoders' per-sample outputs (192 B per sample) so that no sample is re-evaluated.  logical "
                    "gather + scatter bytes / time is quoted for reference only (planes and gradients are cache resident)"}


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


GFLOP_DENSE_PER_VIEW = 2 * (46.55 + 98.00)     # backbone + SuperresolutionHybrid8XDC MACs x 2
MFMA_BF16_PEAK_TFLOPS = 2500.0                # MI355X_MICROARCH.md: dense bf16


def full_generator(torch, dev, D, Di, conv_math):
    """Full-width TriPlaneGenerator (train.py FFHQ config, SURVEY section 8c recipe), random init."""
    from nerffaceediting_amd.training.triplane import TriPlaneGenerator
    rk = dict(superresolution_module="training.superresolution.SuperresolutionHybrid8XDC", sr_antialias=True,
              c_gen_conditioning_zero=False, c_scale=1, superresolution_noise_mode="none", depth_resolution=D,
              depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1, disparity_space_sampling=False,
              clamp_mode="softplus", decoder_lr_mul=1)
    torch.manual_seed(0)
    G = TriPlaneGenerator(512, 25, 512, 512, 3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2), rendering_kwargs=rk,
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"),
                          channel_base=32768, channel_max=512, fused_modconv_default="inference_only", num_fp16_res=0,
                          conv_clamp=None).to(dev).eval().requires_grad_(False)
    G.backbone.synthesis.conv_math = conv_math
    G.superresolution.conv_math = conv_math
    return G


def timed_steps(args, torch, dist, world, step, n_streams=1):
    """W untimed + exactly K timed steps between barrier + synchronize; MAX over ranks.  n_streams > 1: consecutive steps are
    issued on alternating HIP streams (every step still runs completely inside the timed region: both synchronize calls wait
    for all streams), so the latency-bound render of one batch overlaps the convolutions of the next."""
    if n_streams > 1:
        ring = [torch.cuda.Stream() for _ in range(n_streams)]
        for s_ in ring:
            s_.wait_stream(torch.cuda.current_stream())
        inner = step

        def step(i):                      # noqa: F811
            with torch.cuda.stream(ring[i % n_streams]):
                return inner(i)
    for i in range(args.warmup):
        step(i)
    t_pre = time.perf_counter()                # untimed clock-settling pre-roll (see main())
    while args.warmup > 0 and time.perf_counter() - t_pre < getattr(args, "preroll_s", 0.0):
        for i in range(4):
            step(i % args.warmup)
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def extra_workload(args, torch, dist, dev, rank, world):
    from nerffaceediting_amd import apps, ops, sharding
    base = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
            "data": "synthetic", "cpu_baseline": None}
    if args.workload == "twopass":           # config 5: D = Di = 96 (projector.py:33-34), norm planes != normalised(denorm planes)
        Dc = 96
        seed = 1000 + rank
        planes, dec_t, _, c2w_t, K_t, _, _, _ = synth_inputs(torch, dev, seed)
        mean, std = ops.plane_stats(planes)
        gs, gb, as_, ab = ops.make_affine(mean, std, mean.roll(1, 0).contiguous(), std.roll(1, 0).contiguous())   # appearance of the next view
        norm = ops.plane_pack(ops.plane_affine(planes, gs, gb))
        denorm = ops.plane_pack(ops.plane_affine(planes, as_, ab))
        names = ["geo_net.0.weight", "geo_net.0.bias", "geo_net.2.weight", "geo_net.2.bias",
                 "app_net.0.weight", "app_net.0.bias", "app_net.2.weight", "app_net.2.bias"]
        dec_packed = ops.decoder_pack(*[dec_t[k] for k in names])
        opts = dict(depth_resolution=Dc, depth_resolution_importance=Dc, ray_start=2.25, ray_end=3.3, box_warp=1,
                    disparity_space_sampling=False, clamp_mode="softplus")
        ev = {}

        def step(i):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.render(norm, denorm, dec_packed, opts, cam2world=c2w_t, intrinsics=K_t, resolution=R, seed=seed + i, channels_first=True)
            b.record()
            ev[i] = (a, b)
        dt = timed_steps(args, torch, dist, world, step)
        ms = sum(ev[args.warmup + i][0].elapsed_time(ev[args.warmup + i][1]) for i in range(args.steps)) / args.steps
        n_total, M = world * VIEWS_PER_GPU, R * R
        bytes_ray = (Dc + Dc) * 2 * 1536 + Dc * 1536 + 196      # final pass: 192 samples x 2 plane sets; coarse pass: 96 x geometry set
        ach = VIEWS_PER_GPU * M * bytes_ray / (ms * 1e-3) / 1e9
        # counter-derived roofline of the three kernels of the step (bench_roofline.kernel_block).  This workload has no in-kernel clock
        # stamps, so every busy fraction is the ratio of two counters of the SAME profiled launch (busy cycles / GRBM_GUI_ACTIVE): a
        # property of the kernel that no clock difference between the profile box and this one can move (round 6: pricing the profiled
        # busy cycles against this run's time x the profiled clock printed 0.79 on a slow box and 0.91 on a fast one for the same 0.83).
        # `kernel_ms_this_run` = the kernel's share of the step under the profiler x this run's step time.
        samples = {"final": VIEWS_PER_GPU * M * 2 * Dc, "sigma": VIEWS_PER_GPU * M * Dc}
        recs = {"final": PMC["twopass_final"], "sigma": PMC["twopass_sigma"], "importance": PMC["twopass_importance"]}
        prof = {k: rl.load(v) for k, v in recs.items()}
        kern = None
        if all(prof.values()):
            tot = sum(p["avg_ns_profiled"] for p in prof.values())
            kern = {"final": rl.kernel_block(recs["final"], flops=samples["final"] * rl.FLOPS_PER_SAMPLE,
                                             mfma_type="bf16", mfma_work_multiplier=DECODER_MFMA_WORK,
                                             gather_bytes=samples["final"] * 2 * rl.GATHER_BYTES_PER_SAMPLE_SET),
                    "sigma": rl.kernel_block(recs["sigma"], gather_bytes=samples["sigma"] * rl.GATHER_BYTES_PER_SAMPLE_SET),
                    "importance": rl.kernel_block(recs["importance"])}
            for k in kern:
                kern[k]["kernel_ms_this_run"] = ms * prof[k]["avg_ns_profiled"] / tot
        dom = kern["final"] if kern else None
        return dict(base, metric="rays/s, 512^2 x (96+96)-sample two-pass dual-plane render", value=n_total * M * args.steps / dt,
                    unit="rays/s", ms_per_step=dt / args.steps * 1e3, scaling="weak", dtype="f32",
                    config={"workload": "BASELINE config 5: render core through the (norm, denorm) entry, appearance statistics swapped "
                                        "between views, 4 views/GPU/step, 512^2 rays, 96 coarse + 96 importance samples",
                            "views_per_step": n_total, "parallelism": f"views-dp{world}"},
                    roofline=dict(rl.headline_fields(dom), kernel=dom["kernel"] if dom else "nfe::render_ws_kernel<4,2,DUAL> (final pass)",
                                  kernel_ms=dom["kernel_ms_this_run"] if dom else None, step_ms=ms, kernels=kern, logical_gather_gbs=ach,
                                  note="dominant kernel = the final pass over the 192 merged samples with both plane sets; bound / frac = the "
                                       "largest COUNTER-MEASURED busy fraction of that kernel (bench_roofline.py: `fractions`; `models` are "
                                       "instruction-count diagnostics and never the bound); `kernels` lists the same record for the sigma-only "
                                       "coarse pass and importance_kernel; every fraction = busy cycles / elapsed cycles of the SAME profiled launch "
                                       "(the committed rocprofv3 --pmc record of that kernel: clock-independent), `kernel_ms` / `kernel_ms_this_run` = the "
                                       "kernel's share of the step under the profiler x this run's step time; logical_gather_gbs = SURVEY "
                                       "8(d) gather bytes of all three passes / their time, not a physical rate (planes are cache resident)"))

    if args.workload == "editstep":          # forward + backward of the renderer w.r.t. both plane sets, FFHQ rendering config
        Re, Dc = 128, 48
        seed = 1000 + rank
        planes, dec_t, _, c2w_t, K_t, _, _, _ = synth_inputs(torch, dev, seed)
        mean, std = ops.plane_stats(planes)
        gs, gb, as_, ab = ops.make_affine(mean, std, mean.roll(1, 0).contiguous(), std.roll(1, 0).contiguous())
        norm_nchw = ops.plane_affine(planes, gs, gb)       # what the optimiser holds: NCHW planes, a new version every step
        denorm_nchw = ops.plane_affine(planes, as_, ab)
        names = ["geo_net.0.weight", "geo_net.0.bias", "geo_net.2.weight", "geo_net.2.bias",
                 "app_net.0.weight", "app_net.0.bias", "app_net.2.weight", "app_net.2.bias"]
        heads = [dec_t[k] for k in names]
        dec_packed = ops.decoder_pack(*heads)
        opts = dict(depth_resolution=Dc, depth_resolution_importance=Dc, ray_start=2.25, ray_end=3.3, box_warp=1,
                    disparity_space_sampling=False, clamp_mode="softplus")
        Me = Re * Re
        g = torch.Generator(device="cpu").manual_seed(seed)
        cots = tuple(torch.randn(VIEWS_PER_GPU, Me, c, generator=g).to(dev) for c in (32, 15, 1, 1))
        ev = {}

        def step(i):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            # an optimiser step changed the planes: the gather-layout copies are re-made inside the step, as
            # DisentangledImportanceRenderer._packed does on a new tensor version (2 x 25 MB x views, no host sync)
            norm, denorm = ops.plane_pack(norm_nchw), ops.plane_pack(denorm_nchw)
            # as the renderer's autograd function calls it: the forward keeps the decoders' per-sample outputs for the backward
            out = ops.render(norm, denorm, dec_packed, opts, cam2world=c2w_t, intrinsics=K_t, resolution=Re, seed=seed + i, taps=True,
                             sample_colors=True)
            e[1].record()
            ops.render_backward(norm, denorm, heads, 1.0, opts, out[4]["depths_all"], cots, cam2world=c2w_t, intrinsics=K_t, resolution=Re,
                                sample_colors=out[4]["sample_colors"], sample_colors_resolution=out[4]["sample_colors_resolution"])
            e[2].record()
            ev[i] = e
        dt = timed_steps(args, torch, dist, world, step)
        timed = [ev[args.warmup + i] for i in range(args.steps)]
        fwd_ms = sum(e[0].elapsed_time(e[1]) for e in timed) / args.steps
        bwd_ms = sum(e[1].elapsed_time(e[2]) for e in timed) / args.steps
        n_total = world * VIEWS_PER_GPU
        S2 = 2 * Dc
        bytes_sample = 2 * 1536 + 192 + 2 * 1536            # one gather pass over two plane sets + the kept decoder outputs + one scatter into two sets
        ach = VIEWS_PER_GPU * Me * S2 * bytes_sample / (bwd_ms * 1e-3) / 1e9
        return dict(base, metric="rays/s, plane-editing step: 128^2 x (48+48) dual-plane render forward + backward w.r.t. both plane sets",
                    value=n_total * Me * args.steps / dt, unit="rays/s", ms_per_step=dt / args.steps * 1e3, scaling="weak", dtype="f32",
                    config={"workload": "SURVEY 8(f)4 backward pass: 4 views/GPU/step, 128^2 rays, 48 + 48 samples, norm/denorm plane sets with "
                                        "swapped statistics, random cotangents for rgb/seg/depth/wsum, gradients w.r.t. both plane sets; forward_ms "
                                        "includes the per-step NCHW -> gather-layout re-pack of both plane sets",
                            "views_per_step": n_total, "forward_ms": fwd_ms, "backward_ms": bwd_ms, "parallelism": f"views-dp{world}"},
                    roofline=backward_roofline(bwd_ms, VIEWS_PER_GPU * Me * S2, ach))

    ffhq = args.workload == "ffhq"
    conv_math = ("bf16x3" if ffhq else "bf16") if getattr(args, "conv_math", "auto") == "auto" else args.conv_math
    G = full_generator(torch, dev, 48 if ffhq else D, 48 if ffhq else 0, conv_math)
    if args.workload in ("full", "ffhq"):    # config 3, or the FFHQ inference configuration
        NV = 4 if ffhq else 8
        Rn = 128 if ffhq else R
        g = torch.Generator(device="cpu").manual_seed(1000 + rank)
        z = torch.randn(NV, 512, generator=g).to(dev)
        c = apps.orbit_cameras(NV, dev)
        ws = G.mapping(z, c, truncation_psi=0.7, truncation_cutoff=14)
        ev = {}

        def step(i):
            # stage events inside the timed region: [0] backbone [1] stats + render [2] SR [3]
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            wsi = G.mapping(z, c, truncation_psi=0.7, truncation_cutoff=14)
            e[0].record()
            G.stage_events = e
            out = G.synthesis(wsi, c, neural_rendering_resolution=Rn, noise_mode="const")
            e[3].record()
            ev[i] = e
            return out
        dt = timed_steps(args, torch, dist, world, step, n_streams=args.streams)
        # stage times: three more steps on ONE stream, outside the timed region (steps of the timed region overlap on the
        # stream ring, which stretches every stage by the other streams' work)
        torch.cuda.synchronize()
        solo = [args.warmup + args.steps + i for i in range(3)]
        for i in solo:
            step(i)
        torch.cuda.synchronize()
        G.stage_events = None
        timed = [ev[i] for i in solo]
        avg = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in timed) / len(timed)
        syn_ms = avg(0, 3)
        stage = {"backbone": avg(0, 1), "stats_render": avg(1, 2), "sr": avg(2, 3)}
        dense_ms = stage["backbone"] + stage["sr"]
        n_total = world * NV
        flops = GFLOP_DENSE_PER_VIEW * NV * 1e9
        ach = flops / (dense_ms * 1e-3) / 1e12
        what = ("FFHQ inference configuration (train.py:306-307): a1-a14, 4 views/GPU/step, neural render 128^2 x (48+48) -> "
                "SuperresolutionHybrid8XDC to 512^2, %s MFMA convs, fp32 render" % MATH_NAME[conv_math]) if ffhq else (
                "BASELINE config 3: a1-a14, 8 views/GPU/step, neural render 512^2 x 64 -> antialias resize -> "
                "SuperresolutionHybrid8XDC to 512^2, %s MFMA convs (fp32 accumulate), fp32 render" % MATH_NAME[conv_math])
        return dict(base, metric="512^2 views/s, full synthesis (mapping + backbone + %s render + SR)" % ("128^2 x (48+48)" if ffhq else "512^2 x 64"),
                    value=n_total * args.steps / dt, unit="views/s", ms_per_step=dt / args.steps * 1e3, scaling="weak",
                    dtype=conv_math,
                    config={"workload": what,
                            "views_per_step": n_total, "synthesis_ms": syn_ms, "stage_ms": stage, "streams": args.streams,
                            "stage_ms_note": "HIP-event times of three extra steps issued on one stream after the timed region (inside it "
                                             "consecutive steps overlap on the stream ring); synthesis_ms is their sum",
                            "parallelism": f"views-dp{world}"},
                    roofline={"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": None, "kernel": "nfe::conv3_kernel<*> + upfir/torgb (backbone + SR stages)",
                              "kernel_ms": dense_ms,
                              "kernels": dense_kernel_pmc(conv_math), "conv_math": conv_math,
                              "render_stage": render_kernel_block() if not ffhq else None,
                              "note": "289.1 GFLOP per view (SURVEY 8d) / single-stream time of the backbone + SR stages (stage_ms); "
                                      "split-bf16 issues 3 MFMAs per product, so the matrix pipe does 3x these flops in that mode.  `kernels`: "
                                      "matrix-pipe busy fraction of each conv kernel variant = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel "
                                      "cycles) from the committed rocprofv3 --pmc passes of this command (the dense kernels are unchanged since); "
                                      "`render_stage`: the ceilings of the render kernel that takes the stats_render stage (config 3 only)"})

    out = orbit_job(args, torch, dist, dev, rank, world, frames=args.orbit_frames, G=G, steps=args.steps, warmup=args.warmup)
    return dict(base, metric="512^2 views/s, 512-frame orbit (gen_videos camera path)", value=out["views_per_s"], unit="views/s",
                ms_per_step=out["seconds_per_pass"] * 1e3, scaling="strong", dtype="bf16",
                config={"workload": out["workload"], "frames": out["frames"], "frames_per_rank": out["frames_per_rank"],
                        "chunk": out["chunk"], "parallelism": f"frames-dp{world}"},
                roofline=orbit_roofline(out))


# matrix-pipe busy fraction of the conv kernel variants = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) over all launches of a
# `tools/time_full.py` run under rocprofv3 --pmc, written by tools/dense_pmc_table.py into the committed JSON next to the raw summary
DENSE_PMC_FILE = "r06_dense_kernels.json"


def dense_kernel_pmc(conv_math):
    t = rl.load(DENSE_PMC_FILE)
    if t is None:
        return None
    key = {"bf16x3": "bf16x3", "bf16": "bf16", "fp16": "fp16"}.get(conv_math, conv_math)
    return dict(t.get(key, {}), table="profiles/" + DENSE_PMC_FILE)


MATH_NAME = {"bf16x3": "split-bf16 (fp32-grade, 3 MFMAs per product)", "bf16": "bf16", "fp16": "fp16-operand (the reference's GPU arithmetic, 1 MFMA per product)"}


def render_kernel_block(kernel_ms=None, clock_ghz=None):
    """bench_roofline record of the headline render kernel (4 views x 512^2 x 64): the ONE place that says which counter file, census
    entries and algorithmic work belong to it, so that every workload prints the same numbers for it."""
    c = rl.load(PMC["render"])
    if c is None:
        return None
    samples = VIEWS_PER_GPU * R * R * D
    return rl.kernel_block(PMC["render"], kernel_ms=kernel_ms, clock_ghz=clock_ghz, census_parts=rl.render_census_parts(c["kernel"]),
                           flops=samples * rl.FLOPS_PER_SAMPLE, mfma_type="bf16", mfma_work_multiplier=DECODER_MFMA_WORK,
                           gather_bytes=VIEWS_PER_GPU * R * R * BYTES_PER_RAY_S1)


def orbit_roofline(out):
    """The orbit job is 77 % render kernel (8-view launches of the headline kernel, 1.56 ms per 512^2 x 64 view): its roofline block is
    that kernel's, from the same committed counter record as the headline line; the dense stages' aggregate stays beside it."""
    r = render_kernel_block()
    if r is None:
        return {"bound": None, "frac": None, "achieved": None, "peak": None, "unit": None, "traffic": None, "kernel": None, "kernel_ms": None}
    per_view_ms = r["kernel_ms_profiled"] / VIEWS_PER_GPU
    return dict(rl.headline_fields(r), **{"kernel": r["kernel"], "kernel_ms": r["kernel_ms_profiled"],
            "fractions": r["fractions"], "models": r["models"], "algorithmic": r.get("algorithmic"), "render_share_of_pass": per_view_ms * 1e-3 * out["frames_per_rank"] / out["seconds_per_pass"],
            "dense_tflops": out["dense_tflops"], "dense_frac_of_bf16_peak": out["dense_tflops"] / MFMA_BF16_PEAK_TFLOPS,
            "note": "dominant kernel = the render kernel (render_share_of_pass of this rank's wall time, from its profiled 4-view launch "
                    "time); fractions = busy cycles / kernel cycles of the committed rocprofv3 --pmc record of the headline command "
                    "(%s: the same record and the same formulas as the default bench line, at the profiled launch's own time and clock), "
                    "not re-measured by this run; dense_tflops = this rank's conv flops / wall time" % r["counters_file"]})


def orbit_job(args, torch, dist, dev, rank, world, frames=512, G=None, steps=1, warmup=0, chunk=8, return_frames=False):
    """BASELINE config 4, the strong-scaling job: `frames` (per-frame ws, camera) pairs on the gen_videos.py:128-133 camera
    path, full synthesis at 512^2 x 64 with bf16 convs, frames cut into contiguous blocks per rank and rendered `chunk` at a
    time; the uint8 frames ([chunk,512,512,3], what the video writer consumes, gen_videos.py:147-151) of chunk k are
    all-gathered over RCCL while chunk k+1 renders (sharding.ChunkedFrameGather).  Timed like the main loop: barrier +
    synchronize on both sides, MAX over ranks."""
    from nerffaceediting_amd import apps, sharding
    coll = world > 1 or bool(getattr(args, "force_collective", False))      # one-rank RCCL group: the exchange still runs
    if G is None:
        G = full_generator(torch, dev, D, 0, "bf16")
    G.neural_rendering_resolution = R
    V = int(frames)
    c_all = apps.orbit_cameras(V, dev)
    a, b = sharding.shard_range(V, rank, world)
    ws_local = torch.stack([torch.from_numpy(np.random.RandomState(f).randn(14, 512).astype(np.float32)) for f in range(a, max(b, a + 1))]).to(dev)

    def chunk_frames(s_, e_):
        if e_ <= s_:
            return torch.zeros((0, G.img_resolution, G.img_resolution, 3), dtype=torch.uint8, device=dev)
        img = G.synthesis(ws_local[s_ - a:e_ - a].contiguous(), c_all[s_:e_].contiguous(), noise_mode="const")["image"]
        return apps.to_uint8(img)

    def one_pass():
        gat = sharding.ChunkedFrameGather(V, chunk, (G.img_resolution, G.img_resolution, 3), torch.uint8, dev, force_collective=coll)
        ring = apps.StreamRing(dev, getattr(args, "streams", 3))       # chunks rotate over the HIP streams
        for k in range(gat.rounds()):
            sl = gat.local_slice(k)
            gat.submit(k, ring.take(*ring.run(lambda: chunk_frames(*sl))))
        return gat.finish()
    if b > a:                                 # weight packing, allocator (an empty shard - more ranks than frames - renders nothing)
        nw = min(chunk, b - a)
        G.synthesis(ws_local[:nw].contiguous(), c_all[a:a + nw].contiguous(), noise_mode="const")
    for _ in range(warmup):
        one_pass()
    if coll:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(steps, 1)):
        out = one_pass()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0         # this rank's own block + its share of the exchanges, before waiting for the others
    if coll:
        dist.barrier()
    dt = time.perf_counter() - t0
    blocks = [{"rank": rank, "frames": [a, b], "seconds_per_pass_before_barrier": dt_local / max(steps, 1)}]
    if coll:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, blocks[0])
        blocks = gathered
    assert out.shape == (V, G.img_resolution, G.img_resolution, 3)
    per_pass = dt / max(steps, 1)
    if return_frames:
        return out
    return {"rank_blocks": blocks, "workload": f"BASELINE config 4: {V} (per-frame ws, camera) pairs, gen_videos.py:128-133 path, full synthesis 512^2 x 64 + SR, bf16 "
                        f"convs, contiguous blocks per rank, uint8 frames all-gathered per {chunk}-frame chunk under the next chunk's render",
            "frames": V, "frames_per_rank": b - a, "chunk": chunk, "n_gpus": world, "seconds_per_pass": per_pass, "views_per_s": V / per_pass,
            "scaling": "strong", "dense_tflops": GFLOP_DENSE_PER_VIEW * (b - a) * 1e9 / per_pass / 1e12}


def distributed_info(dist, world):
    """What the process group itself reports (not what the command line asked for)."""
    if world > 1 or dist.is_initialized():      # a one-rank group exists only under --force-collective
        assert dist.is_initialized() and dist.get_world_size() == world, (dist.get_world_size(), world)
        return {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                "launcher": os.environ.get("NFE_LAUNCHER", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "none (single rank)")}
    return {"backend": None, "world_size": 1, "launcher": None}


def exchange_check(args, torch, dist, rank, world):
    """--workload exchange: the launch + frame-exchange plumbing of the config-4 job WITHOUT rendering — every rank fills its
    block of a synthetic frame list (frame f = all bytes f % 251) and the blocks travel through sharding.ChunkedFrameGather
    exactly as orbit_job's frames do.  Runs on the gloo backend on CPU tensors (tests/test_sharding_cpu.py starts it through the
    self-launch path) or on RCCL with device tensors.  Not a benchmark: it times nothing that BASELINE.json names."""
    from nerffaceediting_amd import launch, sharding
    if args.backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    else:
        dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        launch.init_process_group(dist, args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
    info = distributed_info(dist, world)
    V, chunk, shape = int(args.orbit_frames), 3, (8, 8, 3)
    t0 = time.perf_counter()
    for _ in range(max(args.steps, 1)):
        gat = sharding.ChunkedFrameGather(V, chunk, shape, torch.uint8, dev)
        for k in range(gat.rounds()):
            s_, e_ = gat.local_slice(k)
            fr = torch.stack([torch.full(shape, f % 251, dtype=torch.uint8, device=dev) for f in range(s_, e_)]) if e_ > s_ \
                else torch.zeros((0,) + shape, dtype=torch.uint8, device=dev)
            gat.submit(k, fr)
        out = gat.finish()
    dt = time.perf_counter() - t0
    want = (torch.arange(V) % 251).to(torch.uint8).to(dev).view(V, 1, 1, 1).expand(V, *shape)
    ok = bool(torch.equal(out, want))
    if world > 1:
        flag = torch.tensor([int(ok)], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(f"rank {rank}: gathered frames differ from the expected frame list")
    return {"metric": "frame-exchange plumbing check (no rendering)", "value": V * max(args.steps, 1) / dt, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic", "config": {"workload": f"exchange check: {V} 8x8x3 uint8 frames, "
            f"chunk {chunk}, ChunkedFrameGather over {info['backend']}", "frames": V}, "distributed": info, "frames_ok": ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 0.6 s of timed launches at the headline shape (round 4's default 20 = 0.13 s)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["render", "full", "ffhq", "orbit", "twopass", "editstep", "exchange"], default="render")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend; 'nccl' is RCCL over xGMI.  'gloo' exists for --workload exchange, the CPU check of "
                         "the launch + frame-exchange plumbing (it renders nothing and is not a benchmark)")
    ap.add_argument("--preroll-s", type=float, default=1.0,
                    help="untimed clock-settling pre-roll: the step is repeated for at least this long right before the timed region")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the full-synthesis workloads alternate their batches on")
    ap.add_argument("--orbit-frames", type=int, default=512, help="frames of the strong-scaling orbit job (BASELINE config 4)")
    ap.add_argument("--no-strong-scaling", action="store_true", help="skip the config-4 orbit job reported beside the default line")
    ap.add_argument("--conv-math", choices=["auto", "bf16x3", "bf16", "fp16"], default="auto",
                    help="operand arithmetic of the convolutions in --workload full / ffhq / orbit: auto = the workload's own (bf16 for config "
                         "3 and the orbit, split-bf16 for ffhq); fp16 = one v_mfma_f32_32x32x16_f16 per product, the reference's GPU arithmetic for "
                         "its fp16 layers (train.py:183, networks_stylegan2.py:421-423)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: still create the (one-rank) RCCL process group and run every frame exchange through it instead of "
                         "short-circuiting - all of the multi-GPU data path a 1-GPU box can execute (train.py:37-43 precedent)")
    args = ap.parse_args()

    from nerffaceediting_amd import launch
    if args.gpus > 1 and not launch.launched_by_a_launcher():
        # No launcher around us: start one fresh process per GPU (train.py:98-103 does the same with mp.spawn).  Nothing in
        # this process has touched the GPU (torch is not even imported yet) and it never replaces itself: the ranks are
        # children, their exit codes decide ours.
        rc, _ = launch.spawn_ranks(__file__, sys.argv[1:], args.gpus)
        sys.exit(rc)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: pass --gpus equal to the number of ranks the launcher started")
    if args.workload == "exchange":
        out = exchange_check(args, torch, dist, rank, world)
        if rank == 0:
            print(json.dumps(out))
        return
    if args.backend != "nccl":
        raise SystemExit("only --workload exchange runs on the gloo backend; every rendering workload needs a GPU and RCCL")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback path exists)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    from nerffaceediting_amd import ops, sharding
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll = world > 1 or args.force_collective
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                           # no launcher: a one-rank rendezvous of our own
            os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        launch.init_process_group(dist, "nccl", device_id=dev)
    dist_info = distributed_info(dist, world)

    if args.workload != "render":
        out = extra_workload(args, torch, dist, dev, rank, world)
        out["distributed"] = dist_info
        if rank == 0:
            print(json.dumps(out))
        if coll:
            dist.destroy_process_group()
        return

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
    probes = torch.zeros(max(args.steps, 1), 4, dtype=torch.int64, device=dev)      # in-run shader clock of every timed launch

    pending = []                                                # outstanding frame exchanges (work, gathered frames)

    def drain():
        while pending:
            pending.pop(0)[0].wait()

    def step(i, timed, decoder_math="bf16x3", events=None):
        events = ev if events is None else events
        mean, std = ops.plane_stats(planes)                     # a4
        aff = ops.make_affine(mean, std)
        packed = ops.plane_pack(planes)
        if timed:
            events[i][0].record()
        rgb, seg, depth, wsum = ops.render(packed, packed, dec_packed, opts, cam2world=c2w_t, intrinsics=K_t,
                                           resolution=R, affines=aff, seed=seed + i, channels_first=True,
                                           clock_probe=probes[i] if timed and decoder_math == "bf16x3" else None,
                                           decoder_math=decoder_math)
        if timed:
            events[i][1].record()
        if coll:                                                # frames of every rank, in view order; the exchange of step i
            frames = rgb[:, :3].reshape(VIEWS_PER_GPU, 3, R, R)  # runs under the render of step i+1 (two in flight at most)
            if len(pending) >= 2:
                pending.pop(0)[0].wait()
            pending.append(sharding.all_gather_frames_async(frames, n_total, force=coll))
        return rgb

    for i in range(args.warmup):
        step(i, False)
    # clock-settling pre-roll (untimed): the same step back to back for >= preroll_s, so the timed region starts at the
    # clock the chip holds under this load, not at the idle boost clock
    t_pre, n_pre = time.perf_counter(), 0
    while time.perf_counter() - t_pre < args.preroll_s:
        for i in range(8):
            step(i, False)
        torch.cuda.synchronize()
        n_pre += 8
    drain()

    def barrier():
        if coll:
            dist.barrier()
    def timed_region(decoder_math, events):
        """EXACTLY args.steps steps between barrier + synchronize on both sides; the MAX over ranks."""
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, True, decoder_math, events)
        drain()                                                 # every frame exchange of the timed steps has completed
        torch.cuda.synchronize(); barrier()
        dt_ = time.perf_counter() - t0
        if coll:
            t = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_

    dt = timed_region("bf16x3", ev)                             # the headline: `value`, `ms_per_step`

    # ---- the SAME timed region once more with the exact-fp32 MFMA decoder (round 6: the same steps, the same bracketing and the
    # same averaging as the headline, not a minimum of three launches): `value_fp32_exact`, `ms_per_step_fp32_exact`
    fe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for i in range(min(args.warmup, 3)):
        step(i, False, "fp32")
    drain()
    dt_fp32 = timed_region("fp32", fe)
    fp32_ms = sum(a_.elapsed_time(b_) for a_, b_ in fe) / max(args.steps, 1)
    # ---- outside the timed regions: the strong-scaling job of config 4
    strong = None if args.no_strong_scaling else orbit_job(args, torch, dist, dev, rank, world, frames=args.orbit_frames)

    if rank == 0:
        kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
        rays_per_step = n_total * M
        value = rays_per_step * args.steps / dt
        launch_bytes = VIEWS_PER_GPU * M * BYTES_PER_RAY_S1
        pr = probes.cpu().numpy().astype(np.float64)
        ok = (pr[:, 3] > pr[:, 1]) & (pr[:, 2] > pr[:, 0])
        clock_ghz = float(np.median((pr[ok, 2] - pr[ok, 0]) / (pr[ok, 3] - pr[ok, 1]) * 0.1)) if ok.any() else None   # 100 MHz reference
        blk = render_kernel_block(kernel_ms=kern_ms, clock_ghz=clock_ghz)
        samples = VIEWS_PER_GPU * M * D
        fp32_blk = rl.kernel_block(PMC["render_fp32"], kernel_ms=fp32_ms, flops=samples * rl.FLOPS_PER_SAMPLE, mfma_type="f32",
                                   mfma_work_multiplier=8192.0 / 7168.0, gather_bytes=launch_bytes)
        out = {
            "metric": "rays/s, 512^2 x 64-sample tri-plane render", "value": value, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the same steps with decoder_math="fp32" (v_mfma_f32_32x32x2_f32: the reference's own fp32 multiplier arithmetic,
            # training/triplane.py:249-270), timed and averaged exactly like `value` / `ms_per_step`
            "value_fp32_exact": rays_per_step * args.steps / dt_fp32, "ms_per_step_fp32_exact": dt_fp32 / args.steps * 1e3,
            "config": {"workload": "BASELINE config 2: render core (a2,a4-a12), 4 views/GPU/step, 512^2 rays x 64 "
                                   "stratified samples, 256^2x96 planes, fp32 in/out, Philox jitter; decoder_math=bf16x3 "
                                   "(fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate: 2-6e-6 "
                                   "max-abs vs the reference; the exact-fp32 MFMA mode runs the same timed region after it: "
                                   "value_fp32_exact / ms_per_step_fp32_exact)",
                       "decoder_math": "bf16x3", "views_per_step": n_total, "views_per_s": n_total * args.steps / dt,
                       "rays_per_view": M, "depth_samples": D, "parallelism": f"views-dp{world}",
                       "preroll_steps": n_pre, "preroll_s": args.preroll_s},
            "distributed": dist_info,
            "roofline": dict(rl.headline_fields(blk), **{
                "kernel": blk["kernel"] if blk else "nfe::render_ws_kernel<4,2,SQUARE=1,GENERIC=0>", "kernel_ms": kern_ms,
                "kernel_mcycles": blk["kernel_mcycles"] if blk else None,   # shader cycles per launch (x 1e6): the box-independent figure (clocks differ by +-4 % between boxes)
                "fractions": blk["fractions"] if blk else None, "models": blk["models"] if blk else None,
                "algorithmic": blk.get("algorithmic") if blk else None,
                "kernel_ms_fp32_exact": fp32_ms, "fp32_exact": fp32_blk,
                "algorithmic_bytes_per_launch": launch_bytes, "detail": blk,
                "note": "bound / frac = the LARGEST COUNTER-MEASURED busy fraction of a hardware unit (`fractions`: texture addresser, L1 "
                        "request rate, matrix pipe, LDS issue, true HBM traffic - busy cycles of the committed rocprofv3 --pmc record of "
                        "this command / (unit count x kernel cycles), kernel cycles = this run's HIP-event time x the clock the kernel "
                        "itself read).  `models` are diagnostics from instruction counts x micro-benchmarked SIMD cycles per class and are "
                        "never the bound: valu_pipe, simd_no_overlap (round 4's simd_pipes: vector and matrix time summed), "
                        "simd_overlap_aware = max(valu + 8 n_mfma, 32 n_mfma) and the counter mfma_coexec_share beside them.  "
                        "`algorithmic`: SURVEY 8(d)'s 0.94 GFLOP/kray against the bf16 MFMA peak at the split mode's 3.43x issued work, "
                        "and its 98 500 B/ray against the aggregate L1 bandwidth (256 CUs x 64 B/clk) and, as north_star quotes it, "
                        "against 8 TB/s (not physical: the planes are cache resident, HBM carries `traffic`).  Formulas: bench_roofline.py"}),
        }
        if strong is not None:
            out["strong_scaling"] = strong
        if not args.no_cpu_baseline and world == 1:          # reported at N=1 only (host cores are shared by all ranks)
            out["cpu_baseline"] = cpu_baseline(planes_np, dec_np, c2w, K, opts, seed)
        print(json.dumps(out))
    if coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
