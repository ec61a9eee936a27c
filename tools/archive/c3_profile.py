#!/usr/bin/env python3
"""Where a conv3_kernel wave spends its cycles (diagnostic build -DC3_PROFILE, s_memtime stamps around the load / barrier phase,
the MFMA phase and the epilogue of every K-group, summed over all waves):
    bash tools/build_variant.sh c3prof -DC3_PROFILE && NFE_RENDER_LIB=.../c3prof.so python tools/c3_profile.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerffaceediting_amd import _lib, dense_ops as D

dev = torch.device("cuda:0")
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
MATH = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
NV = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = torch.Generator(device="cpu").manual_seed(0)
for name, N, H, cin, cout, up in (("SR conv1 256^2 256->256", NV, 256, 256, 256, 1), ("SR conv1 512^2 128->128", NV, 512, 128, 128, 1),
                                  ("b128 conv1 128^2 256->256", NV, 128, 256, 256, 1), ("b64 conv1 64^2 512->512", NV, 64, 512, 512, 1),
                                  ("SR up 256->512 256->128", NV, 256, 256, 128, 2), ("b64 up 32->64 512->512", NV, 32, 512, 512, 2)):
    x = torch.randn(N, H, H, cin, generator=g).to(dev)
    st = (torch.randn(N, cin, generator=g) * 0.5 + 1).to(dev)
    w = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
    packed, wsq = D.conv_pack(w)
    dc = D.conv_demod(st, wsq)
    bias = torch.zeros(cout, device=dev)
    mode = _lib.NFE_CONV_3X3_UP2 if up == 2 else _lib.NFE_CONV_3X3
    for _ in range(2):
        D.modulated_conv(x, st, packed, cout, mode, bias, dcoef=dc, math=MATH)
    torch.cuda.synchronize()
    raw.nfe_debug_c3_profile(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        D.modulated_conv(x, st, packed, cout, mode, bias, dcoef=dc, math=MATH)
    e1.record(); torch.cuda.synchronize()
    raw.nfe_debug_c3_profile(buf, 1)
    load, comp, epi, waves, vm, bar, life, _ = (int(v) for v in buf)
    tot = load + comp + epi
    print(f"{name:28s} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us/launch (incl. modsplit / upfir)  per wave: load+barrier {load / waves:9.0f} cyc ({100 * load / tot:4.1f} %), "
          f"MFMA phase {comp / waves:9.0f} ({100 * comp / tot:4.1f} %), epilogue {epi / waves:8.0f} ({100 * epi / tot:4.1f} %); "
          f"of load+barrier: vmcnt wait {vm / waves:8.0f}, barrier wait {bar / waves:8.0f}, issue {(load - vm - bar) / waves:8.0f}; wave life {life / waves:9.0f} (prologue {(life - tot) / waves:7.0f})  [{MATH}, {N} views, {D.describe(mode, MATH, N, H, H, cin, cout)}]")
