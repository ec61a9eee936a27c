"""Soak: the same launch repeated must give bit-identical outputs (no intra-kernel race).  Covers the headline configuration,
the two-pass dual-plane render, a depth-split launch and a mid-size two-pass render.
    python tools/soak_determinism.py [repeats]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerffaceediting_amd import ops  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    H = 256
    shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
    heads = [torch.randn(*s, generator=g).to(dev) * (1.0 if len(s) == 2 else 0.2) for s in shapes]
    dec = ops.decoder_pack(*heads)
    K1 = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]])
    for N, R, D, Di, dual in ((4, 512, 64, 0, False), (2, 512, 96, 96, True), (1, 128, 48, 48, True), (2, 256, 48, 48, False)):
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
        opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1.0)

        def run():
            if dual:
                return ops.render(packed, packed2, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, seed=11)
            return ops.render(packed, packed, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, affines=aff, seed=11)
        ref = [t.clone() for t in run()]
        bad = 0
        for i in range(reps):
            out = run()
            bad += int(not all(torch.equal(a, b) for a, b in zip(out, ref)))
        print(f"N={N} R={R} {D}+{Di} {'dual' if dual else 'single'}: {reps} repeats, {bad} differing")
        assert bad == 0


if __name__ == "__main__":
    main()
