"""Soak: a render launch must not depend on what earlier kernels left in LDS / registers.  Between repeats of the same render
launch, run convolutions and backward kernels on fresh random data (they fill the CUs' LDS with input-dependent bytes).
    python tools/soak_lds_garbage.py [repeats]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerffaceediting_amd import _lib, dense_ops as D, ops  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    H = 256
    shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
    heads = [torch.randn(*s, generator=g).to(dev) * (1.0 if len(s) == 2 else 0.2) for s in shapes]
    dec = ops.decoder_pack(*heads)
    from oracle import render_oracle as orc      # camera construction only
    import numpy as np
    N, R = 2, 256
    c2w = torch.from_numpy(np.concatenate([orc.lookat_pose(np.pi / 2 + y, np.pi / 2 + p, [0, 0, 0.2], 2.7).reshape(1, 4, 4) for y, p in ((0.3, -0.2), (-0.9, 0.4))])).to(dev)
    K = torch.from_numpy(np.repeat(orc.fov_to_intrinsics(18.837)[None], N, 0)).to(dev)
    raw = torch.randn(N, 96, H, H, generator=g).to(dev)
    mean, std = ops.plane_stats(raw)
    packed = ops.plane_pack(raw)
    packed2 = ops.plane_pack((torch.randn(N, 96, H, H, generator=g) * 1.2).to(dev))
    aff = ops.make_affine(mean, std)
    weight = torch.randn(128, 128, 3, 3, generator=g).to(dev)
    cpacked, wsq = D.conv_pack(weight)
    bias = torch.zeros(128, device=dev)
    configs = [dict(depth_resolution=64, depth_resolution_importance=0), dict(depth_resolution=32, depth_resolution_importance=32)]
    for dual in (False, True):
        for o in configs:
            opts = dict(o, ray_start=2.25, ray_end=3.3, box_warp=1.0)

            def run():
                if dual:
                    return ops.render(packed, packed2, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, seed=3, taps=True)
                return ops.render(packed, packed, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, affines=aff, seed=3, taps=True)
            ref = run()
            ref_t = [t.clone() for t in ref[:4]]
            bad = 0
            for i in range(reps):
                x = torch.randn(2, 128, 128, 128, device=dev) * (10.0 ** (i % 5 - 2))           # garbage makers
                st = torch.rand(2, 128, device=dev) + 0.5
                D.modulated_conv(x, st, cpacked, 128, _lib.NFE_CONV_3X3, bias=bias, dcoef=D.conv_demod(st, wsq), lrelu=True, act_gain=1.4,
                                 math="bf16x3" if i % 2 else "bf16")
                cots = (torch.randn(N, R * R, 32, device=dev), torch.randn(N, R * R, 15, device=dev), None, None)
                if i % 3 == 0:
                    ops.render_backward(packed, packed2, heads, 1.0, opts, ref[4]["depths_all"], cots, cam2world=c2w, intrinsics=K, resolution=R)
                out = run()
                bad += int(not all(torch.equal(a, b) for a, b in zip(out[:4], ref_t)))
            print(f"{'dual' if dual else 'single'} {o}: {reps} repeats after garbage kernels, {bad} differing")
            assert bad == 0


if __name__ == "__main__":
    main()
