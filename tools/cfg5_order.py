"""Config 5, final pass: does the ORDER in which a tile's samples are evaluated matter?  (round 6, review item 6)

The final pass of the 96 + 96 two-pass render evaluates, per wave step, ONE index of the merged depth lists of an 8 x 4 ray tile; at one
index the tile's rays sit 8 - 12 strata apart (oracle/texel_window_census.py: 65 distinct texels under the 128 taps of a depth-dependent
plane instead of the coarse pass's 14 - 19).  A depth-bucket march would bring the taps of a step back into one stratum's footprint.  Its
BEST case is what this script launches as `ideal`: the same kernel family (`render_ws_kernel<4,2,DUAL>`, both plane sets, both heads)
over 192 STRATIFIED samples per ray - every step's 32 rays in one stratum, the footprint of the coarse pass - with exactly the final
pass's sample count, loads, MFMAs and march.  `merged` is the shipped two-pass step.  Same planes, decoder, cameras and seeds.

    python3 tools/cfg5_order.py merged|ideal [steps]        (under rocprofv3 --kernel-trace --stats, or PMC_PROG=... tools/pmc.sh)

Prints one JSON line with the event-timed step; the per-kernel figures come from the trace.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from nerffaceediting_amd import ops  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "merged"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    seed = 1000
    planes, dec_t, _, c2w_t, K_t, _, _, _ = bench.synth_inputs(torch, dev, seed)
    mean, std = ops.plane_stats(planes)
    gs, gb, as_, ab = ops.make_affine(mean, std, mean.roll(1, 0).contiguous(), std.roll(1, 0).contiguous())
    norm = ops.plane_pack(ops.plane_affine(planes, gs, gb))
    denorm = ops.plane_pack(ops.plane_affine(planes, as_, ab))
    names = ["geo_net.0.weight", "geo_net.0.bias", "geo_net.2.weight", "geo_net.2.bias",
             "app_net.0.weight", "app_net.0.bias", "app_net.2.weight", "app_net.2.bias"]
    dec_packed = ops.decoder_pack(*[dec_t[k] for k in names])
    D, Di = (96, 96) if mode == "merged" else (192, 0)
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1,
                disparity_space_sampling=False, clamp_mode="softplus")

    def step(i):
        return ops.render(norm, denorm, dec_packed, opts, cam2world=c2w_t, intrinsics=K_t, resolution=bench.R, seed=seed + i, channels_first=True)

    for i in range(2):
        step(i)
    kernels = ops.render_last_kernels()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(steps):
        out = step(2 + i)
    b.record()
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(t).all()) for t in out[:4])
    print(json.dumps({"mode": mode, "samples_per_ray": D + Di, "plane_sets": 2, "kernels": kernels, "ms_per_step": a.elapsed_time(b) / steps,
                      "steps": steps, "lost_handoffs": ops.render_handoff_aborts()}))


if __name__ == "__main__":
    main()
