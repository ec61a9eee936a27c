#!/usr/bin/env python3
"""A/B of render_kernel (fused) against render_ws_kernel (wave-specialised) on the headline shape: one child process per
NFE_RENDER_WS mode (the switch is read once per process), every child renders BASELINE config 2 (4 views x 512^2 x 64, Philox
jitter, fixed seed), prints the median / min launch time and a digest of the four outputs; the parent compares digests with mode 0
(the two kernels are meant to be bit-identical) and prints a table.
    python3 tools/r04_ws_ab.py [modes ...]      default: 0 42 63 33 84 44
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from nerffaceediting_amd import ops
    dev = torch.device("cuda:0")
    R, D = int(os.environ.get("AB_R", bench.R)), int(os.environ.get("AB_D", bench.D))
    planes, dec_t, _, c2w, K, _, _, _ = bench.synth_inputs(torch, dev, 1000)
    names = ["geo_net.0.weight", "geo_net.0.bias", "geo_net.2.weight", "geo_net.2.bias", "app_net.0.weight", "app_net.0.bias", "app_net.2.weight", "app_net.2.bias"]
    dec = ops.decoder_pack(*[dec_t[k] for k in names])
    opts = dict(depth_resolution=D, depth_resolution_importance=0, ray_start=2.25, ray_end=3.3, box_warp=1, disparity_space_sampling=False, clamp_mode="softplus")
    mean, std = ops.plane_stats(planes)
    aff, packed = ops.make_affine(mean, std), ops.plane_pack(planes)
    run = lambda seed: ops.render(packed, packed, dec, opts, cam2world=c2w, intrinsics=K, resolution=R, affines=aff, seed=seed, channels_first=True)
    out = run(7)
    torch.cuda.synchronize()
    dig = hashlib.sha256(b"".join(t.cpu().numpy().tobytes() for t in out)).hexdigest()[:16]
    reps = int(os.environ.get("AB_REPS", "30"))
    for i in range(5):
        run(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); run(100 + i); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    again = run(7)
    same = all(torch.equal(x, y) for x, y in zip(out, again))
    nan = any(bool(torch.isnan(t).any()) for t in out[:2])
    print("AB " + json.dumps({"mode": os.environ.get("NFE_RENDER_WS", "0"), "median_ms": ms[len(ms) // 2], "min_ms": ms[0], "digest": dig,
                              "repeatable": same, "nan": nan, "rgb_sum": float(out[0].double().sum())}))


def main():
    modes = sys.argv[1:] or ["0", "42", "63", "33", "84", "44"]
    rows = []
    for rep in range(int(os.environ.get("AB_ROUNDS", "2"))):
        for m in modes:
            env = dict(os.environ, NFE_RENDER_WS=m.split("@")[0], PYTHONPATH=ROOT)
            if "@" in m:            # mode@variant: a library built by tools/build_render_variant.sh
                env["NFE_RENDER_LIB"] = os.path.join(ROOT, "nerffaceediting_amd", "csrc", "build", "variants", m.split("@")[1] + ".so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
            if not line:
                print(f"mode {m}: FAILED rc={r.returncode} {r.stderr[-600:]}")
                continue
            rows.append(dict(json.loads(line[-1][3:]), mode=m))
            print(rows[-1], flush=True)
    ref = next((r["digest"] for r in rows if r["mode"] == "0"), None)
    for r in rows:
        print(f"mode {r['mode']:>12}: median {r['median_ms']:.3f} ms  min {r['min_ms']:.3f} ms  bit-identical to fused: {r['digest'] == ref}  repeatable: {r['repeatable']}")


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
