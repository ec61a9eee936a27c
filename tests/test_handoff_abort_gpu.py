"""The error surface of a lost wave hand-off in render_ws_kernel (include/nfe_render.h, "lost hand-offs"; NFE_EHANDOFF).

The wave-specialised render kernel bounds its producer / consumer waits so that a lost partner can never hang the GPU.  A wait
that gives up must not become silent garbage: the reference's ops fail loudly (TORCH_CHECK, torch_utils/ops/bias_act.cpp:39-55).
Here a child interpreter shortens the bound to ONE poll (NFE_WS_SPIN_LIMIT=1, read once per process: a consumer's first wait
for a tile that needs a whole gather is then certain to be abandoned) and checks the three things the header promises:
  1. every output of the affected call is NaN - the four images AND the taps it was given (merged depths, coarse weights, fine depths);
  2. the failure belongs to THAT call (ABI v15): ops.render_call_status() - the per-call query on the call's own workspace, at the
     caller's synchronisation point - raises RuntimeError (NFE_EHANDOFF, -4) naming the count, and the process's sticky word
     (ops.render_status) counts (lost, 1 call) in one 64-bit word; ops.raise_if_handoff_lost() raises once and clears it;
  3. NO LATER CALL IS REFUSED: the next render launches and is judged on its own (v14 made it launch nothing and return the error
     of the earlier call - the wrong call, possibly another stream's or thread's).  With the one-poll bound still in force a
     wave-specialised launch would abort again, so the child checks this through the fused kernel (too few ray blocks).
Both passes of a two-pass render count (the coarse sigma-only pass has no depth min/max words but does have the abort counter).
The backward's decoder kernel has the same surface: whole gradient buffers NaN, ops.render_call_status(backward=True), sticky word."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

PROG = r"""
import numpy as np, torch
from nerffaceediting_amd import ops
from oracle import render_oracle as orc           # camera construction only
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(5)
N, R, H = 1, 512, 64
raw = torch.randn(N, 96, H, H, generator=g).to(dev)
mean, std = ops.plane_stats(raw)
packed, aff = ops.plane_pack(raw), ops.make_affine(mean, std)
shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
dec = ops.decoder_pack(*[(torch.randn(*s, generator=g) * (1.0 if len(s) == 2 else 0.2)).to(dev) for s in shapes])
c2w = torch.from_numpy(orc.lookat_pose(np.pi / 2 + 0.3, np.pi / 2 - 0.2, [0, 0, 0.2], 2.7).reshape(1, 4, 4).astype(np.float32)).to(dev)
K = torch.from_numpy(orc.fov_to_intrinsics(18.837)[None].astype(np.float32)).to(dev)
cam = dict(cam2world=c2w, intrinsics=K, affines=aff)
for D, Di, want in ((16, 0, ["render_ws_kernel<4,2>"]),
                    (12, 12, ["render_ws_kernel<4,2,SIGMA_ONLY>", "importance_kernel", "render_ws_kernel<4,2>"])):
    opts = dict(depth_resolution=D, depth_resolution_importance=Di, ray_start=2.25, ray_end=3.3, box_warp=1.0)
    assert ops.render_status(clear=True) is not None
    res = ops.render(packed, packed, dec, opts, resolution=R, taps=True, **cam)         # 8 192 ray blocks: wave-specialised launch
    out, tap = res[:4], res[4]
    assert ops.render_last_kernels() == want, ops.render_last_kernels()
    torch.cuda.synchronize()
    assert all(bool(torch.isnan(o).all()) for o in out), "outputs of a call that lost hand-offs must be NaN, all of them"
    assert all(bool(torch.isnan(v).all()) for k, v in tap.items() if isinstance(v, torch.Tensor)), "the taps of such a call must be NaN too"
    assert ("weights_coarse" in tap) == (Di > 0)
    lost, calls = ops.render_status()
    assert lost > 0 and calls == 1, (lost, calls)
    assert ops.render_handoff_aborts() == lost
    try:
        ops.render_call_status()
        raise SystemExit("the per-call status of a poisoned call must raise")
    except RuntimeError as e:
        assert "(-4)" in str(e) and "lost %d wave hand-offs" % lost in str(e), str(e)
    assert ops.render_status() == (lost, 1)                                         # sticky: the per-call query does not clear it
    small = ops.render(packed, packed, dec, opts, resolution=64, **cam)[:4]         # the NEXT call is not refused: 128 ray blocks, fused kernel, no hand-off
    assert not any(k.startswith("render_ws_kernel") for k in ops.render_last_kernels())
    ops.render_call_status()                                                        # ... and it is healthy: no exception, finite outputs
    assert all(bool(torch.isfinite(o).all()) for o in small) and ops.render_handoff_aborts() == 0
    assert ops.render_status() == (lost, 1)
    try:
        ops.raise_if_handoff_lost()
        raise SystemExit("raise_if_handoff_lost must raise while the sticky word is set")
    except RuntimeError as e:
        assert "lost %d wave hand-offs" % lost in str(e), str(e)
    assert ops.render_status() == (0, 0)                                            # reported once, then cleared (one atomic exchange)
    ops.raise_if_handoff_lost()
print("HANDOFF_ABORT_OK")
"""


def test_lost_handoff_poisons_outputs_and_fails_the_next_call():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NFE_WS_SPIN_LIMIT="1", PYTHONPATH=root)
    env.pop("NFE_RENDER_WS", None)
    r = subprocess.run([sys.executable, "-c", PROG], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "HANDOFF_ABORT_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


BWD_PROG = r"""
import numpy as np, torch
from nerffaceediting_amd import ops
from tests.test_render_backward_gpu import _editing_size_case
dev = torch.device("cuda:0")
args, kw = _editing_size_case(dev)                 # its forward: 2 x 128^2 rays = 1 024 ray blocks, the fused render kernel (no hand-off)
assert not any(k.startswith("render_ws_kernel") for k in ops.render_last_kernels())
ops.render_status(clear=True)
for k, need in enumerate(((True, True), (True, False), (False, True))):
    gg, ga = ops.render_backward(*args, need=need, **kw)
    torch.cuda.synchronize()
    for g, n in ((gg, need[0]), (ga, need[1])):
        if not n:
            continue
        assert bool(torch.isnan(g).all()), "EVERY entry of a gradient buffer of a backward that lost hand-offs must be NaN"
    lost = ops.render_handoff_aborts(backward=True)
    assert lost > 0
    try:
        ops.render_call_status(backward=True)
        raise SystemExit("the per-call status of a poisoned backward must raise")
    except RuntimeError as e:
        assert "(-4)" in str(e) and "lost %d wave hand-offs" % lost in str(e), str(e)
    assert ops.render_status()[1] == k + 1                 # poisoned calls so far, in the sticky word
assert ops.render_status(clear=True)[1] == 3 and ops.render_status() == (0, 0)
print("BWD_HANDOFF_ABORT_OK")
"""


def test_lost_handoff_in_the_backward_poisons_the_gradients():
    """bwd_decoder_kernel (producer / consumer wave pairs) under NFE_WS_SPIN_LIMIT=1: the consumer's first wait is abandoned, the
    launch ends, and the kernel that closes the call overwrites both gradient buffers with NaN, whichever chunk lost the hand-off;
    the count is in the call's workspace (nfe_render_backward_call_status -> NFE_EHANDOFF) and in the sticky word (ABI v15)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NFE_WS_SPIN_LIMIT="1", PYTHONPATH=root)
    env.pop("NFE_BWD_DECODER", None)
    r = subprocess.run([sys.executable, "-c", BWD_PROG], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BWD_HANDOFF_ABORT_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_default_bound_loses_nothing():
    """The shipped bound (2^18 polls, ~50 ms): a full-size launch reports zero lost hand-offs and a clean status word."""
    import torch
    from nerffaceediting_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(6)
    raw = torch.randn(2, 96, 128, 128, generator=g).to(dev)
    mean, std = ops.plane_stats(raw)
    shapes = [(64, 32), (64,), (16, 64), (16,), (64, 32), (64,), (32, 64), (32,)]
    dec = ops.decoder_pack(*[(torch.randn(*s, generator=g) * (1.0 if len(s) == 2 else 0.2)).to(dev) for s in shapes])
    import numpy as np
    from oracle import render_oracle as orc           # camera construction only
    c2w = torch.from_numpy(np.concatenate([orc.lookat_pose(np.pi / 2 + y, np.pi / 2 - 0.2, [0, 0, 0.2], 2.7).reshape(1, 4, 4)
                                           for y in (0.3, -0.3)]).astype(np.float32)).to(dev)
    K = torch.from_numpy(np.repeat(orc.fov_to_intrinsics(18.837)[None], 2, 0).astype(np.float32)).to(dev)
    ops.render_status(clear=True)
    opts = dict(depth_resolution=32, depth_resolution_importance=32, ray_start=2.25, ray_end=3.3, box_warp=1.0)
    out = ops.render(ops.plane_pack(raw), ops.plane_pack(raw), dec, opts, cam2world=c2w, intrinsics=K, resolution=512,
                     affines=ops.make_affine(mean, std))[:4]
    assert ops.render_last_kernels()[0] == "render_ws_kernel<4,2,SIGMA_ONLY>"
    assert ops.render_handoff_aborts() == 0 and ops.render_status() == (0, 0)
    assert all(bool(torch.isfinite(o).all()) for o in out)
