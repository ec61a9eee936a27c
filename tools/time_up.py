#!/usr/bin/env python3
"""Time the up-sampling layers of config 3 alone (modsplit + conv3 UP2 [+ upfir]) through the C ABI:  python tools/time_up.py [math] [views]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerffaceediting_amd import _lib, dense_ops as D

dev = torch.device("cuda:0")
MATH = sys.argv[1] if len(sys.argv) > 1 else "bf16"
NV = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = torch.Generator(device="cpu").manual_seed(0)
out = []
LAYERS = (("SR1 up 256->512 256->128", 256, 256, 128), ("SR0 up 128->256 32->256", 128, 32, 256), ("b256 up 128->256 256->128", 128, 256, 128),
          ("b128 up 64->128 512->256", 64, 512, 256))
if os.environ.get("LAYER"):            # one layer only (counter passes average over a kernel name)
    LAYERS = (LAYERS[int(os.environ["LAYER"])],)
for name, H, cin, cout in LAYERS:
    x = torch.randn(NV, H, H, cin, generator=g).to(dev)
    st = (torch.randn(NV, cin, generator=g) * 0.5 + 1).to(dev)
    w = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
    packed, wsq = D.conv_pack(w)
    dc = D.conv_demod(st, wsq)
    bias = torch.zeros(cout, device=dev)
    # PIPE=1 (default): as conv0 of a SynthesisBlock runs it - no fp32 output, only the consumer's modulated bf16 image (next_styles);
    # PIPE=0: the fp32 activation is written (1 GB for the largest layer: a heavier epilogue than the pipeline's)
    kw = dict(dcoef=dc, math=MATH)
    if os.environ.get("PIPE", "1") != "0":
        kw.update(next_styles=(torch.randn(NV, cout, generator=g) * 0.5 + 1).to(dev), want_out=False)
    ITERS = int(os.environ.get("ITERS", "10"))
    for _ in range(max(3, ITERS // 2)):
        D.modulated_conv(x, st, packed, cout, _lib.NFE_CONV_3X3_UP2, bias, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        D.modulated_conv(x, st, packed, cout, _lib.NFE_CONV_3X3_UP2, bias, **kw)
    e1.record(); torch.cuda.synchronize()
    out.append(f"{name}: {e0.elapsed_time(e1) / ITERS * 1e3:7.1f} us")
print(f"[{MATH} x{NV}] " + " | ".join(out))
