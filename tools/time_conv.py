#!/usr/bin/env python3
"""Time single modulated 3x3 layers (modsplit + conv3 [+ upfir]) through the C ABI:  python tools/time_conv.py [math] [views]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerffaceediting_amd import _lib, dense_ops as D

dev = torch.device("cuda:0")
MATH = sys.argv[1] if len(sys.argv) > 1 else "bf16"
NV = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = torch.Generator(device="cpu").manual_seed(0)
for name, H, cin, cout, up in (("SR conv1 256^2 256->256", 256, 256, 256, 1), ("SR conv1 512^2 128->128", 512, 128, 128, 1),
                               ("b128 conv1 128^2 256->256", 128, 256, 256, 1), ("SR up 256->512 256->128", 256, 256, 128, 2)):
    x = torch.randn(NV, H, H, cin, generator=g).to(dev)
    st = (torch.randn(NV, cin, generator=g) * 0.5 + 1).to(dev)
    w = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
    packed, wsq = D.conv_pack(w)
    dc = D.conv_demod(st, wsq)
    bias = torch.zeros(cout, device=dev)
    mode = _lib.NFE_CONV_3X3_UP2 if up == 2 else _lib.NFE_CONV_3X3
    for _ in range(3):
        D.modulated_conv(x, st, packed, cout, mode, bias, dcoef=dc, math=MATH)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        D.modulated_conv(x, st, packed, cout, mode, bias, dcoef=dc, math=MATH)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2 * 9 * cin * cout * H * H * NV
    print(f"{name:28s} [{MATH}, {NV} views] {us:8.1f} us/launch (incl. modsplit / upfir)  {fl / us / 1e6:7.1f} TFLOP/s")
