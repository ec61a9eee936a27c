"""Time nfe_render_backward (and the forward it differentiates) at plane-editing sizes.
    python tools/time_backward.py [N] [R] [D] [Di] [H]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerffaceediting_amd import ops  # noqa: E402


def main():
    N, R, D, Di, H = [int(a) for a in sys.argv[1:6]] + [1, 128, 48, 48, 256][len(sys.argv) - 1:]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    planes_n = torch.randn(N, 3, H, H, 32, generator=g).to(dev)
    planes_d = (torch.randn(N, 3, H, H, 32, generator=g) * 1.3 + 0.2).to(dev)
    shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
    heads = [torch.randn(*s, generator=g).to(dev) * (1.0 if len(s) == 2 else 0.2) for s in shapes]
    heads[3][0] += 2.0
    dec = ops.decoder_pack(*heads)
    th = torch.linspace(-0.4, 0.4, N)
    c2w = torch.eye(4).repeat(N, 1, 1)
    c2w[:, 0, 0], c2w[:, 0, 2], c2w[:, 2, 0], c2w[:, 2, 2] = torch.cos(th), torch.sin(th), -torch.sin(th), torch.cos(th)
    c2w[:, :3, 2] *= -1
    c2w[:, :3, 3] = -2.7 * c2w[:, :3, 2]
    K = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(N, 1, 1)
    c2w, K = c2w.to(dev), K.to(dev)
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1.0)
    M = R * R
    cots = (torch.randn(N, M, 32, device=dev), torch.randn(N, M, 15, device=dev), torch.randn(N, M, 1, device=dev), torch.randn(N, M, 1, device=dev))
    if os.environ.get("NO_RGB_COT"):          # geometry-only loss (seg / depth): the appearance head drops out of every kernel
        cots = (None,) + cots[1:]

    KEEP = not os.environ.get("NO_SAMPLE_COLORS")      # the forward keeps the decoders' per-sample outputs (ABI v11)

    def fwd():
        return ops.render(planes_n, planes_d, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, seed=1, taps=True, sample_colors=KEEP)

    out = fwd()
    depths = out[4]["depths_all"]
    colors = out[4].get("sample_colors")
    colors_res = out[4].get("sample_colors_resolution")

    def bwd(need=(True, True)):
        return ops.render_backward(planes_n, planes_d, heads, 1.0, opts, depths, cots, cam2world=c2w, intrinsics=K, resolution=R, need=need,
                                   sample_colors=colors, sample_colors_resolution=colors_res)

    def timeit(fn, it=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it

    if os.environ.get("BOTH_ONLY"):                    # per-kernel statistics of one mode (rocprofv3 averages over every launch)
        print(f"N={N} R={R} D={D}+{Di} planes {H}^2: backward (both sets) {timeit(bwd, 20):.3f} ms   [{N * M * (D + Di) / 1e6:.2f} M samples]")
        return
    print(f"N={N} R={R} D={D}+{Di} planes {H}^2: forward {timeit(fwd):.3f} ms, backward (both sets) {timeit(bwd):.3f} ms, "
          f"backward (geometry set only) {timeit(lambda: bwd((True, False))):.3f} ms   [{N * M * (D + Di) / 1e6:.2f} M samples]")


if __name__ == "__main__":
    main()
