#!/usr/bin/env python3
"""Back-to-back synthesis batches: eager launches on 1 / 3 streams against hipGraph replays (graphs.GraphedSynthesis, one captured graph
per stream) on 1 / 2 / 3 streams.  Does removing the host from the launch path - and the dispatch gaps between the ~110 kernels of one
synthesis - buy throughput once batches already overlap on a stream ring?  Usage: tools/time_graph_streams.py [N] [R] [D] [Di] [math]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerffaceediting_amd import apps
from nerffaceediting_amd.graphs import GraphedSynthesis
from nerffaceediting_amd.training.triplane import TriPlaneGenerator

N, R, D, Di = (int(a) for a in (sys.argv[1:5] + ["4", "128", "48", "48"][len(sys.argv) - 1:])[:4])
math = sys.argv[5] if len(sys.argv) > 5 else "bf16x3"
rk = dict(superresolution_module="training.superresolution.SuperresolutionHybrid8XDC", sr_antialias=True, c_gen_conditioning_zero=False,
          c_scale=1, superresolution_noise_mode="none", depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3,
          box_warp=1, disparity_space_sampling=False, clamp_mode="softplus", decoder_lr_mul=1)
torch.manual_seed(0)
G = TriPlaneGenerator(512, 25, 512, 512, 3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2), rendering_kwargs=rk,
                      sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"), channel_base=32768,
                      channel_max=512, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None)
dev = torch.device("cuda:0")
G = G.to(dev).eval().requires_grad_(False)
G.backbone.synthesis.conv_math = math
G.superresolution.conv_math = math
c = apps.orbit_cameras(max(N, 2), dev)[:N]
z = torch.randn(N, 512, device=dev)
ws = G.mapping(z, c, truncation_psi=0.7, truncation_cutoff=14)
for _ in range(3):
    G.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const")
torch.cuda.synchronize()
STEPS = 36


def run(ns, call):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            call(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        with torch.cuda.stream(streams[i % ns]):
            call(i % ns)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS


graphs = [GraphedSynthesis(G, batch=N, neural_rendering_resolution=R, noise_mode="const") for _ in range(3)]
ref = G.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const")["image_raw"].clone()
for rep in range(2):
    for ns in (1, 3):
        dt = run(ns, lambda k: G.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const"))
        print(f"eager, {ns} stream(s): {dt * 1e3:.3f} ms per batch of {N} -> {N / dt:.1f} views/s")
    for ns in (1, 2, 3):
        dt = run(ns, lambda k: graphs[k](ws, c, seed=k))
        print(f"graph, {ns} stream(s): {dt * 1e3:.3f} ms per batch of {N} -> {N / dt:.1f} views/s")
out = graphs[0](ws, c, seed=0)
torch.cuda.synchronize()
print("graph output finite:", bool(torch.isfinite(out["image"]).all()), "image_raw vs eager (different jitter):",
      float((out["image_raw"] - ref).abs().max()))
