"""Throughput map of the render core over ray counts, sample counts and plane-set modes (markdown table on stdout).
    python tools/perf_map.py > profiles/r01_perf_map.md
single = raw planes + affines (synthesis path), dual = separate norm / denorm plane sets (editing path)."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerffaceediting_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    H = 256
    shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
    heads = [torch.randn(*s, generator=g).to(dev) * (1.0 if len(s) == 2 else 0.2) for s in shapes]
    dec = ops.decoder_pack(*heads)
    K1 = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]])
    print("| views | rays/view | samples | planes | ms | M rays/s | G samples/s |")
    print("|---|---|---|---|---|---|---|")
    for N in (1, 4):
        raw = torch.randn(N, 96, H, H, generator=g).to(dev)
        mean, std = ops.plane_stats(raw)
        packed = ops.plane_pack(raw)
        aff = ops.make_affine(mean, std)
        packed2 = ops.plane_pack((torch.randn(N, 96, H, H, generator=g) * 1.2).to(dev))
        th = torch.linspace(-0.4, 0.4, N)
        c2w = torch.eye(4).repeat(N, 1, 1)
        c2w[:, 0, 0], c2w[:, 0, 2], c2w[:, 2, 0], c2w[:, 2, 2] = torch.cos(th), torch.sin(th), -torch.sin(th), torch.cos(th)
        c2w[:, :3, 2] *= -1
        c2w[:, :3, 3] = -2.7 * c2w[:, :3, 2]
        c2w, K = c2w.to(dev), K1.repeat(N, 1, 1).to(dev)
        for R in (64, 128, 256, 512):
            for D, Di in ((64, 0), (48, 48), (96, 96)):
                for mode in ("single", "dual"):
                    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1.0)

                    def run(i):
                        if mode == "single":
                            return ops.render(packed, packed, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, affines=aff, seed=i)
                        return ops.render(packed, packed2, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, seed=i)
                    run(0); torch.cuda.synchronize()
                    it = 20 if R <= 128 else 6
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(it):
                        run(i + 1)
                    e1.record(); torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / it
                    rays = N * R * R
                    samples = rays * (D + Di + (D if Di else 0))          # the coarse pass evaluates D samples once more
                    print(f"| {N} | {R}^2 | {D}+{Di} | {mode} | {ms:.3f} | {rays / ms / 1e3:.1f} | {samples / ms / 1e6:.2f} |")


if __name__ == "__main__":
    main()
