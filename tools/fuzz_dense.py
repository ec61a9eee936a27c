"""One-off wide fuzz of nfe_modulated_conv against the torch-CPU oracle (oracle/dense_oracle.py: modulated_conv + bias_act +
upsample2d of the skip image) on random shapes: ragged sizes, every channel-count class, 3x3 / up-sampling / 1x1 modes, both
arithmetic modes, whatever kernel path the library picks.  Not part of the suite.
    python tools/fuzz_dense.py [seed] [cases]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerffaceediting_amd import _lib, dense_ops as D  # noqa: E402
from oracle import dense_oracle as O  # noqa: E402


def main():
    seed, n = (int(a) for a in (sys.argv[1:3] + ["0", "150"][len(sys.argv) - 1:]))
    rng = np.random.RandomState(seed)
    dev = torch.device("cuda:0")
    bad, paths = [], {}
    lib = _lib.load()
    for i in range(n):
        N = int(rng.choice([1, 2, 3, 5]))
        H, W = int(rng.choice([4, 8, 9, 16, 31, 32, 33, 40, 64, 72, 100])), int(rng.choice([4, 8, 16, 32, 33, 48, 64, 95, 128]))
        cin = int(rng.choice([16, 32, 48, 64, 128, 256]))
        cout = int(rng.choice([32, 64, 96, 128]))
        math = str(rng.choice(["bf16x3", "bf16"]))
        kind = int(rng.randint(4))                    # 0: 3x3, 1: up-sampling 3x3, 2: 1x1 ToRGB with skip, 3: 3x3 with the ToRGB fused in its epilogue
        g = torch.Generator(device="cpu").manual_seed(seed * 100003 + i)
        if kind == 1:
            H, W = min(H, 48), min(W, 64)
        if kind == 3:                                  # sizes the fused path accepts (LDS-DMA path, even sizes for the skip image)
            H, W, cout = int(rng.choice([32, 64, 96])), int(rng.choice([32, 64, 128])), int(rng.choice([64, 128, 256]))
            if not D.fuses_rgb(_lib.NFE_CONV_3X3, math, N, H, W, cin, cout, 3):
                kind = 0
        if kind == 2:
            H, W, cout = H - H % 2 or 2, W - W % 2 or 2, int(rng.choice([3, 96]))
        if kind == 3:
            x = torch.randn(N, H, W, cin, generator=g)
            styles = torch.randn(N, cin, generator=g) * 0.5 + 1.0
            weight = torch.randn(cout, cin, 3, 3, generator=g); bias = torch.randn(cout, generator=g)
            noise = torch.randn(H, W, generator=g)
            C = int(rng.choice([1, 3, 4]))
            rw = torch.randn(C, cout, 1, 1, generator=g); rs = torch.randn(N, cout, generator=g) * 0.05; rb = torch.randn(C, generator=g)
            skip = torch.randn(N, H // 2, W // 2, C, generator=g) if rng.rand() < 0.7 else None
            want_x = bool(rng.rand() < 0.5)
            xo = O.bias_act(O.modulated_conv(x.permute(0, 3, 1, 2), weight, styles, noise=noise[None, None] * 0.3), bias, act="lrelu", clamp=256.0)
            ref = O.bias_act(O.modulated_conv(xo, rw, rs, demodulate=False), rb, clamp=256.0)
            if skip is not None:
                ref = ref + O.upsample2d(skip.permute(0, 3, 1, 2))
            packed, wsq = D.conv_pack(weight.to(dev))
            out, rgb = D.modulated_conv(x.to(dev), styles.to(dev), packed, cout, _lib.NFE_CONV_3X3, bias=bias.to(dev), dcoef=D.conv_demod(styles.to(dev), wsq),
                                        noise=noise.to(dev), noise_strength=0.3, lrelu=True, act_gain=2 ** 0.5, clamp=256.0, math=math, want_out=want_x,
                                        rgb=(rw.to(dev), rs.to(dev), rb.to(dev), None if skip is None else skip.to(dev), 256.0))
            assert (out is not None) == want_x
            paths[(3, want_x)] = paths.get((3, want_x), 0) + 1
            err = float((rgb.cpu().permute(0, 3, 1, 2) - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
            if want_x:
                err = max(err, float((out.cpu().permute(0, 3, 1, 2) - xo).abs().max()) / float(xo.abs().max()))
            if not (err <= (3e-5 if math == "bf16x3" else 3e-2)):
                bad.append((kind, (N, H, W, cin, cout, C), math, want_x, err))
            continue
        k = 1 if kind == 2 else 3
        x = torch.randn(N, H, W, cin, generator=g)
        styles = torch.randn(N, cin, generator=g) * (0.05 if kind == 2 else 0.5) + (0.0 if kind == 2 else 1.0)
        weight = torch.randn(cout, cin, k, k, generator=g)
        bias = torch.randn(cout, generator=g)
        up = 2 if kind == 1 else 1
        noise = torch.randn(H * up, W * up, generator=g) if kind != 2 else None
        xn = x.permute(0, 3, 1, 2)
        if kind == 2:
            skip = torch.randn(N, H // 2, W // 2, cout, generator=g)
            ref = O.bias_act(O.modulated_conv(xn, weight, styles, demodulate=False), bias, clamp=256.0) + O.upsample2d(skip.permute(0, 3, 1, 2))
            packed, _ = D.conv_pack(weight.to(dev))
            got = D.modulated_conv(x.to(dev), styles.to(dev), packed, cout, _lib.NFE_CONV_1X1, bias=bias.to(dev), lrelu=False, act_gain=1.0,
                                   clamp=256.0, skip=skip.to(dev), math=math)
            mode = _lib.NFE_CONV_1X1
        else:
            ref = O.bias_act(O.modulated_conv(xn, weight, styles, noise=noise[None, None] * 0.3, up=up), bias, act="lrelu", clamp=256.0)
            packed, wsq = D.conv_pack(weight.to(dev))
            dcoef = D.conv_demod(styles.to(dev), wsq)
            mode = _lib.NFE_CONV_3X3_UP2 if up == 2 else _lib.NFE_CONV_3X3
            got = D.modulated_conv(x.to(dev), styles.to(dev), packed, cout, mode, bias=bias.to(dev), dcoef=dcoef, noise=noise.to(dev),
                                   noise_strength=0.3, lrelu=True, act_gain=2 ** 0.5, clamp=256.0, math=math)
        fastp = bool(lib.nfe_conv_accepts_split(mode, H, W, cin, cout)) if kind != 2 else None
        paths[(kind, fastp)] = paths.get((kind, fastp), 0) + 1
        err = float((got.cpu().permute(0, 3, 1, 2) - ref).abs().max()) / float(ref.abs().max())
        if not (err <= (3e-5 if math == "bf16x3" else 3e-2)):
            bad.append((kind, (N, H, W, cin, cout), math, fastp, err))
    print(f"{n} dense cases from seed {seed}: {len(bad)} failures; (kind, fast path) counts {paths}")
    for b in bad[:12]:
        print("  ", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
