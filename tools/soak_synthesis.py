"""Soak: the whole synthesis() repeated with fixed jitter must be bit-identical run to run (deterministic split-K, no racing
epilogues) — batch 1 and 4, both conv arithmetic modes.
    python tools/soak_synthesis.py [repeats]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerffaceediting_amd import apps  # noqa: E402
from nerffaceediting_amd.training.triplane import TriPlaneGenerator  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    R, D, Di = 64, 24, 24
    rk = dict(superresolution_module="training.superresolution.SuperresolutionHybrid8XDC", sr_antialias=True, c_gen_conditioning_zero=False,
              c_scale=1, superresolution_noise_mode="none", depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3,
              box_warp=1, disparity_space_sampling=False, clamp_mode="softplus", decoder_lr_mul=1)
    torch.manual_seed(0)
    G = TriPlaneGenerator(512, 25, 512, 512, 3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2), rendering_kwargs=rk,
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"), channel_base=32768,
                          channel_max=512, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None).to(dev).eval().requires_grad_(False)
    for N in (1, 4):
        for math in ("bf16x3", "bf16"):
            G.backbone.synthesis.conv_math = G.superresolution.conv_math = math
            c = apps.orbit_cameras(max(N, 2), dev)[:N]
            ws = G.mapping(torch.randn(N, 512, device=dev), c, truncation_psi=0.7, truncation_cutoff=14)
            uc, uf = torch.rand(N, R * R, D, device=dev), torch.rand(N * R * R, Di, device=dev)

            def run():
                G.renderer.inject_jitter(uc, uf)
                o = G.synthesis(ws, c, neural_rendering_resolution=R, noise_mode="const")
                return [o[k].clone() for k in ("image", "image_raw", "image_seg", "image_depth")]
            ref = run()
            bad = sum(int(not all(torch.equal(a, b) for a, b in zip(run(), ref))) for _ in range(reps))
            print(f"N={N} {math}: {reps} repeats, {bad} differing")
            assert bad == 0


if __name__ == "__main__":
    main()
