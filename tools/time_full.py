#!/usr/bin/env python3
"""Stage timings of the full path (BASELINE config 3 shape): TriPlaneGenerator.synthesis at full width,
N views, R^2 neural render x (D + Di) samples, SR to 512^2.  Usage: tools/time_full.py [N] [R] [D] [Di] [math] [sr_math]
(sr_math defaults to math; "bf16x3 bf16" = fp32-grade backbone with a bf16 SR head, the split the reference makes with its fp16 SR)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerffaceediting_amd.training.triplane import TriPlaneGenerator

N, R, D, Di = (int(a) for a in (sys.argv[1:5] + ["4", "128", "48", "48"][len(sys.argv) - 1:])[:4])
math = sys.argv[5] if len(sys.argv) > 5 else "bf16x3"
sr_math = sys.argv[6] if len(sys.argv) > 6 else math
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
G.superresolution.conv_math = sr_math
from nerffaceediting_amd import apps

c = apps.orbit_cameras(max(N, 2), dev)[:N]
z = torch.randn(N, 512, device=dev)


def timed(fn, reps=10):
    for _ in range(3):          # back-to-back warm-up: lets torch's caching allocator reach the pool size of overlapping calls
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


t_map, ws = timed(lambda: G.mapping(z, c, truncation_psi=0.7, truncation_cutoff=14))
t_bb, planes = timed(lambda: G.backbone.synthesis.forward_nhwc(ws, out_planes=True, noise_mode="const"))
t_all, out = timed(lambda: G.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const"))
feat = torch.randn(N, 128, 128, 32, device=dev)
t_sr, _ = timed(lambda: G.superresolution.forward_nhwc(feat[..., :3].contiguous(), feat, ws, noise_mode="none"))
print(f"N={N} R={R} D={D}+{Di} math={math}: mapping {t_map:.2f} ms, backbone {t_bb:.2f} ms, SR {t_sr:.2f} ms, synthesis total {t_all:.2f} ms "
      f"-> {N / t_all * 1e3:.1f} views/s; backbone {46.55 * 2 * N / t_bb:.1f} TFLOP/s, SR {98.0 * 2 * N / t_sr:.1f} TFLOP/s (algorithmic)")
print("finite:", bool(torch.isfinite(out["image"]).all()), out["image"].shape)
if sr_math != math:          # what the cheaper SR arithmetic costs in the final image
    G.superresolution.conv_math = math
    G.renderer.inject_jitter(torch.rand(N, R * R, D, device=dev, generator=None), torch.rand(N * R * R, max(Di, 1), device=dev)[:, :Di] if Di else None)
    uj = G.renderer._jitter
    ref = G.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const")["image"]
    G.superresolution.conv_math = sr_math
    G.renderer.inject_jitter(*uj)
    alt = G.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const")["image"]
    print(f"  image with sr_math={sr_math} vs {math}: max-abs {float((alt - ref).abs().max()):.3e}, rms {float((alt - ref).square().mean().sqrt()):.3e} (image range {float(ref.min()):.2f}..{float(ref.max()):.2f})")
from nerffaceediting_amd.graphs import GraphedSynthesis
g = GraphedSynthesis(G, batch=N, neural_rendering_resolution=R, noise_mode="const")
t_graph, _ = timed(lambda: g(ws, c))
print(f"  hipGraph replay of synthesis: {t_graph:.2f} ms -> {N / t_graph * 1e3:.1f} views/s")
