"""Which intermediate of nfe_render_backward differs between two builds of the library (or between two runs of one)?
    python tools/bwd_ws_diff.py <libA.so|base> <libB.so|base> [N] [R]
Runs the backward of one fixed problem with each library in ONE process (workspace zeroed first), copies the workspace, and
compares it region by region (the layout of nfe_render_bwd.hip: nfe_render_backward).  Ranks inside a bin (binrank.y) and the
sorted index list depend on the order in which waves reach their atomics and are expected to differ; everything else is a pure
function of the inputs.  Round 4, profiles/experiments/r04_bwd_restructure_race.md.
"""
import ctypes  # noqa: F401
import os
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerffaceediting_amd import _lib, ops  # noqa: E402


def align256(x):
    return (x + 255) & ~255


def use(path):
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.__file__), "libnfe_render.so") if path == "base" else os.path.abspath(path)
    return _lib.load()


def main():
    la, lb = sys.argv[1], sys.argv[2]
    N, R = [int(a) for a in sys.argv[3:5]] + [2, 128][len(sys.argv) - 3:]
    D = Di = 48
    H = 256
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    planes_n = torch.randn(N, 3, H, H, 32, generator=g).to(dev)
    planes_d = (torch.randn(N, 3, H, H, 32, generator=g) * 1.3 + 0.2).to(dev)
    shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
    heads = [torch.randn(*s, generator=g).to(dev) * (1.0 if len(s) == 2 else 0.2) for s in shapes]
    heads[3][0] += 2.0
    th = torch.linspace(-0.4, 0.4, N)
    c2w = torch.eye(4).repeat(N, 1, 1)
    c2w[:, 0, 0], c2w[:, 0, 2], c2w[:, 2, 0], c2w[:, 2, 2] = torch.cos(th), torch.sin(th), -torch.sin(th), torch.cos(th)
    c2w[:, :3, 2] *= -1
    c2w[:, :3, 3] = -2.7 * c2w[:, :3, 2]
    K = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(N, 1, 1)
    c2w, K = c2w.to(dev), K.to(dev)
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1.0)
    M, S = R * R, D + Di
    cots = (torch.randn(N, M, 32, device=dev), torch.randn(N, M, 15, device=dev), torch.randn(N, M, 1, device=dev), torch.randn(N, M, 1, device=dev))

    use(la)
    dec = ops.decoder_pack(*heads)
    out = ops.render(planes_n, planes_d, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, seed=1, taps=True, sample_colors=True)
    depths, colors, colors_res = out[4]["depths_all"], out[4].get("sample_colors"), out[4].get("sample_colors_resolution")

    def run(path):
        lib = use(path)
        need = lib.nfe_render_backward_workspace_bytes(N, M, S)
        ws = ops._workspace(dev, need)
        ws.zero_()
        grads = ops.render_backward(planes_n, planes_d, heads, 1.0, opts, depths, cots, cam2world=c2w, intrinsics=K, resolution=R,
                                    sample_colors=colors, sample_colors_resolution=colors_res)
        torch.cuda.synchronize()
        return ws[:need].clone(), [x.clone() for x in grads], need

    wa, ga, na = run(la)
    wb, gb, nb = run(lb)
    ns = N * M * S
    slots = N * ((M + 63) // 64) * 64 * S
    regions = [("dec", 32768)] + [(n, align256(slots * 4)) for n in ("rec_sig", "rec_a", "rec_T", "rec_t")] + \
              [("packed", align256(_lib.NFE_DECODER_PACKED_FLOATS * 4)), ("frags", align256(52 * 64 * 16)), ("df", align256(slots * 256)),
               ("rec_key", align256(slots * 3 * 8)), ("rec_w", align256(slots * 3 * 16)), ("binrank", align256(slots * 3 * 8)),
               ("perm", align256(slots * 3 * 4)), ("counts", align256((1 << 19) * 4)), ("offsets", align256((1 << 19) * 4))]
    off = 0
    print(f"{la} ({na} B) vs {lb} ({nb} B): N={N} R={R} S={S}, {slots} sample slots")
    for name, size in regions:
        a, b = wa[off:off + size], wb[off:off + size]
        if name == "binrank":       # (bin, rank) pairs: the bins must agree, the ranks need not
            a32, b32 = a.view(torch.int32).view(-1, 2), b.view(torch.int32).view(-1, 2)
            nbin = int((a32[:, 0] != b32[:, 0]).sum())
            nrank = int((a32[:, 1] != b32[:, 1]).sum())
            print(f"  {name:8s} @{off:>12d} +{size:>11d}: bins differ in {nbin} records, ranks in {nrank} (ranks may)")
            if nbin:
                idx = torch.nonzero(a32[:, 0] != b32[:, 0]).flatten()[:8].tolist()
                for i in idx:
                    print(f"      record {i}: plane {i // slots} slot {i % slots} = wave {(i % slots) // 64} lane {i % 64}: A {a32[i].tolist()} B {b32[i].tolist()}")
        else:
            ne = a != b
            nd = int(ne.sum())
            print(f"  {name:8s} @{off:>12d} +{size:>11d}: {nd} bytes differ")
            if nd and name not in ("perm",):
                words = torch.nonzero(ne.view(-1, 4).any(1)).flatten()
                w = words[:8].tolist()
                unit = {"df": 256, "rec_key": 8, "rec_w": 16, "rec_sig": 4, "rec_a": 4, "rec_T": 4}.get(name, 4)
                for i in w:
                    e = i * 4 // unit
                    av = a.view(torch.float32)[i].item() if name in ("df", "rec_w", "rec_sig", "rec_a", "rec_T") else a.view(torch.int32)[i].item()
                    bv = b.view(torch.float32)[i].item() if name in ("df", "rec_w", "rec_sig", "rec_a", "rec_T") else b.view(torch.int32)[i].item()
                    tag = f"wave {(e % slots) // 64} lane {e % 64} plane {e // slots}" if name in ("rec_key", "rec_w") else (f"row {e} = wave {e // 64} sample {e % 64} col {(i * 4 % 256) // 4}" if name == "df" else "")
                    print(f"      word {i} (element {e}; {tag}): A {av} B {bv}")
                if name == "rec_w" and os.environ.get("PKADD"):      # ISA patch: the packed multiply replaced by a packed add, v8 = wx0 + wy1
                    fa, fb = a.view(torch.float32).view(-1, 4)[:slots], b.view(torch.float32).view(-1, 4)[:slots]      # plane 0
                    wx0, wy1 = fa[:, 0] + fa[:, 2], fa[:, 2] + fa[:, 3]
                    inner = (fa > 0).all(1)                      # no folded / out-of-range taps
                    bad = torch.nonzero(inner & ((fb[:, 2] - (wx0 + wy1)).abs() > 1e-5)).flatten()
                    v = fb[bad, 2]
                    print(f"      packed add: {bad.numel()} interior records of plane 0 with w2 != wx0 + wy1; of them == wx0: {int(((v - wx0[bad]).abs() < 1e-5).sum())}, "
                          f"== wy1: {int(((v - wy1[bad]).abs() < 1e-5).sum())}, == 0: {int((v == 0).sum())}; lanes {sorted(set((bad % 64).tolist()))[:20]}")
                    for r in bad[:4].tolist():
                        print(f"      record {r} lane {r % 64}: wx0 {wx0[r].item():.6f} wy1 {wy1[r].item():.6f} B.w2 {fb[r, 2].item():.6f}")
                elif name == "rec_w" and os.environ.get("COPY_SLOT"):     # ISA-patch experiments: slot COPY_SLOT of B's records carries a probe
                    cs, ws_ = int(os.environ["COPY_SLOT"]), int(os.environ.get("WATCH_SLOT", "2"))
                    fa, fb = a.view(torch.float32).view(-1, 4), b.view(torch.float32).view(-1, 4)
                    bad = torch.nonzero(fa[:, ws_] != fb[:, ws_]).flatten()
                    print(f"      records whose word {ws_} differs: {bad.numel()}; probe (B word {cs}) in them: == A word {ws_}: "
                          f"{int((fb[bad, cs] == fa[bad, ws_]).sum())}, == 0: {int((fb[bad, cs] == 0).sum())}; probe != A word {ws_} in all records: "
                          f"{int((fb[:, cs] != fa[:, ws_]).sum())} of {fa.shape[0]}")
                    for r in bad[:4].tolist():
                        print(f"      record {r} (plane {r // slots} wave {(r % slots) // 64} lane {r % 64}): A {[round(x, 6) for x in fa[r].tolist()]}  B {[round(x, 6) for x in fb[r].tolist()]}")
                elif name == "rec_w":          # whole records of the first differing ones, and how many records differ per plane
                    recs = torch.unique(words // 4)
                    fa, fb = a.view(torch.float32).view(-1, 4), b.view(torch.float32).view(-1, 4)
                    for r in recs[:6].tolist():
                        print(f"      record {r}: A {[round(x, 6) for x in fa[r].tolist()]}  B {[round(x, 6) for x in fb[r].tolist()]}")
                    print(f"      differing records per plane: {torch.bincount(recs // slots, minlength=3).tolist()}; per word of the record: {torch.bincount(words % 4, minlength=4).tolist()}")
                lanes = (words * 4 // unit) % 64
                hist = torch.bincount(lanes, minlength=64).tolist()
                print(f"      differing words by lane (element % 64): {hist}")
        off += size
    for name, a, b in zip(("grad_geo", "grad_app"), ga, gb):
        if a is None:
            continue
        d = (a - b).abs().max().item()
        print(f"  {name}: max |A - B| = {d:.3e} of {a.abs().max().item():.3e}")


if __name__ == "__main__":
    main()
