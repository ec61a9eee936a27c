#!/usr/bin/env python3
"""Up-sampling layer (transposed conv -> fp32 scratch -> FIR epilogue) on 8 views at once vs view by view: does the scratch of
one view (135 MB at 512^2 x 128) survive in the 256 MB Infinity Cache between the two kernels?   python tools/upfir_batch_ab.py [math]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerffaceediting_amd import _lib, dense_ops as D

dev = torch.device("cuda:0")
MATH = sys.argv[1] if len(sys.argv) > 1 else "bf16"
NV = 8
g = torch.Generator(device="cpu").manual_seed(0)


def timeit(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for name, H, cin, cout in (("SR up 256->512 256->128", 256, 256, 128), ("SR up 128->256 32->256", 128, 32, 256), ("b256 up 128->256 256->128", 128, 256, 128),
                           ("b128 up 64->128 512->256", 64, 512, 256)):
    x = torch.randn(NV, H, H, cin, generator=g).to(dev)
    st = (torch.randn(NV, cin, generator=g) * 0.5 + 1).to(dev)
    w = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
    packed, wsq = D.conv_pack(w)
    dc = D.conv_demod(st, wsq)
    bias = torch.zeros(cout, device=dev)
    res = {}
    for per in (8, 4, 2, 1):
        def run():
            return [D.modulated_conv(x[i:i + per], st[i:i + per], packed, cout, _lib.NFE_CONV_3X3_UP2, bias, dcoef=dc[i:i + per], math=MATH) for i in range(0, NV, per)]
        res[per] = timeit(run)
    ref = D.modulated_conv(x, st, packed, cout, _lib.NFE_CONV_3X3_UP2, bias, dcoef=dc, math=MATH)
    one = torch.cat([D.modulated_conv(x[i:i + 1], st[i:i + 1], packed, cout, _lib.NFE_CONV_3X3_UP2, bias, dcoef=dc[i:i + 1], math=MATH) for i in range(NV)])
    print(f"{name:28s} [{MATH}] us per {NV} views, views per call 8/4/2/1: " + " / ".join(f"{res[p]:7.1f}" for p in (8, 4, 2, 1)) +
          f"   identical: {bool((ref == one).all())}")
