"""Row (e) on the one GPU a test box has: a ONE-RANK `nccl` (= RCCL) process group, with every frame exchange forced through the
collective instead of short-circuiting at world size 1.  This executes what the gloo tests cannot: RCCL communicator creation
with `device_id=`, `all_gather_into_tensor(async_op=True)` on uint8 / fp32 device tensors, the staging-buffer reuse of
`ChunkedFrameGather`, and the ordering of the collective's stream against the three render streams of `apps.StreamRing`.
It does NOT measure scaling (no second GPU): SCALE_rNN.json is the only place a curve can come from.

Reference precedent for the process-group setup: /root/reference/train.py:37-43 (init_process_group per rank, nccl backend).
The group lives in a child interpreter so that the pytest process itself never holds a process group.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import argparse, json, os, sys
import numpy as np, torch, torch.distributed as dist
root = sys.argv[1]; sys.path.insert(0, root)
import bench
from nerffaceediting_amd import apps, sharding, launch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
report = {"backend": dist.get_backend(), "world": dist.get_world_size()}

# 1. plain and async exchange of a frame block (fp32 and uint8), forced through RCCL
for dt in (torch.float32, torch.uint8):
    fr = (torch.arange(5 * 3 * 16 * 16, device=dev) % 251).to(dt).view(5, 3, 16, 16)
    got = sharding.all_gather_frames(fr, 5, force=True)
    assert got.data_ptr() != fr.data_ptr() and torch.equal(got, fr), dt          # a new tensor: the collective wrote it
    work, got2 = sharding.all_gather_frames_async(fr, 5, force=True)
    assert work is not None
    work.wait(); assert torch.equal(got2, fr), dt
# without force a one-rank group still short-circuits (the product default)
assert sharding.all_gather_frames(fr, 5).data_ptr() == fr.data_ptr()

# 2. the chunked schedule: ragged tail, more rounds than collectives in flight, staging buffers reused
for V, chunk in ((5, 2), (7, 3), (16, 8), (3, 4)):
    gat = sharding.ChunkedFrameGather(V, chunk, (8, 8, 3), torch.uint8, dev, force_collective=True, max_in_flight=2)
    assert gat.active and gat.world == 1
    for k in range(gat.rounds()):
        s, e = gat.local_slice(k)
        gat.submit(k, torch.stack([torch.full((8, 8, 3), f % 251, dtype=torch.uint8, device=dev) for f in range(s, e)]))
    out = gat.finish()
    want = (torch.arange(V) % 251).to(torch.uint8).to(dev).view(V, 1, 1, 1).expand(V, 8, 8, 3)
    assert torch.equal(out, want), (V, chunk)
    report.setdefault("staging_buffers", {})[f"{V}/{chunk}"] = len(gat.free)
    assert len(gat.free) <= 2, len(gat.free)          # reuse: never more staging buffers than collectives in flight

# 3. the config-4 job on 16 frames, three render streams, exchange over RCCL: bit-equal to the non-distributed run, 20 repeats
V, R, D = 16, bench.R, bench.D
G = bench.full_generator(torch, dev, D, 0, "bf16")
c_all = apps.orbit_cameras(V, dev)
u = torch.rand(V, R * R, D, generator=torch.Generator(device=dev).manual_seed(3), device=dev)
orig = G.synthesis
def synth(ws, c, **kw):
    f0 = int((c_all == c[0]).all(dim=1).nonzero()[0, 0])
    G.renderer.inject_jitter(u[f0:f0 + ws.shape[0]].contiguous())
    return orig(ws, c, **kw)
G.synthesis = synth
ref = bench.orbit_job(argparse.Namespace(streams=3, force_collective=False), torch, dist, dev, 0, 1, frames=V, G=G, return_frames=True).clone()
reps = int(sys.argv[2])
for i in range(reps):
    got = bench.orbit_job(argparse.Namespace(streams=3, force_collective=True), torch, dist, dev, 0, 1, frames=V, G=G, return_frames=True)
    assert torch.equal(got, ref), f"repeat {i}: frames gathered over RCCL differ from the non-distributed run"
report["orbit_repeats"] = reps
dist.barrier(); dist.destroy_process_group()
print("RCCL1 " + json.dumps(report))
"""


def test_one_rank_rccl_group_runs_the_frame_exchange():
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, "20"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL1 ")][-1]
    rep = json.loads(line[6:])
    print(rep)
    assert rep["backend"] == "nccl" and rep["world"] == 1 and rep["orbit_repeats"] == 20


def test_bench_force_collective_reports_the_nccl_backend():
    """`python bench.py --gpus 1 --force-collective`: the headline step with its frame exchange run through a one-rank RCCL group
    (two collectives in flight, drained inside the timed region) - the line must say which backend carried it."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "4", "--warmup", "1",
                        "--preroll-s", "0.2", "--no-cpu-baseline", "--orbit-frames", "16"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:6]          # ONE line on stdout: RCCL's version banner (NCCL_DEBUG=VERSION on the GPU boxes) goes to stderr
    out = json.loads(lines[0])
    assert out["distributed"]["backend"] == "nccl" and out["distributed"]["world_size"] == 1, out["distributed"]
    assert out["n_gpus"] == 1 and out["value"] > 2.0e6
    assert out["strong_scaling"]["frames"] == 16 and out["strong_scaling"]["rank_blocks"][0]["frames"] == [0, 16]
