"""CPU, world_size 2 and 8 over gloo: view sharding + frame all-gather (the N>1 path of bench.py / SURVEY §8e)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerffaceediting_amd import sharding


def test_shard_range_covers_all_views_in_order():
    for V in (1, 2, 7, 8, 512, 513):
        for G in (1, 2, 3, 8):
            spans = [sharding.shard_range(V, r, G) for r in range(G)]
            flat = [i for a, b in spans for i in range(a, b)]
            assert flat == list(range(V)), (V, G, spans)
            assert max(b - a for a, b in spans) == -(-V // G)


def _worker(rank, world, port, V, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = sharding.shard_range(V, rank, world)
        local = torch.stack([torch.full((3, 4, 4), float(i)) for i in range(a, b)]) if b > a else torch.zeros(0, 3, 4, 4)
        out = sharding.all_gather_frames(local, V)
        ok = out.shape == (V, 3, 4, 4) and all(float(out[i, 0, 0, 0]) == i for i in range(V))
        works = [sharding.all_gather_frames_async(local + k, V) for k in range(3)]          # bench.py keeps two in flight
        for k, (w, o) in enumerate(works):
            w.wait()
            ok = ok and o.shape == (V, 3, 4, 4) and all(float(o[i, 0, 0, 0]) == i + k for i in range(V))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("V", [4, 5])
def test_all_gather_frames_world2(V):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29511 + V
    procs = [ctx.Process(target=_worker, args=(r, 2, port, V, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _chunk_worker(rank, world, port, V, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sharding.ChunkedFrameGather(V, chunk, (2, 3), torch.int32, torch.device("cpu"), max_in_flight=2)
        order = []
        for k in range(g.rounds()):
            s, e = g.local_slice(k)
            order.append((s, e))
            frames = torch.stack([torch.full((2, 3), i, dtype=torch.int32) for i in range(s, e)]) if e > s else torch.zeros(0, 2, 3, dtype=torch.int32)
            g.submit(k, frames)                       # "render" of round k+1 proceeds while round k is exchanged
        out = g.finish()
        ok = out.shape == (V, 2, 3) and all(int(out[i, 1, 2]) == i for i in range(V))
        a, b = sharding.shard_range(V, rank, world)
        ok = ok and [i for s, e in order for i in range(s, e)] == list(range(a, b))      # every owned frame rendered once, in order
        q.put((rank, bool(ok), g.rounds()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("V,chunk,rounds", [(5, 2, 2), (7, 3, 2), (3, 4, 1), (9, 2, 3)])
def test_chunked_overlapped_gather_world2_uneven_shards(V, chunk, rounds):
    """The chunked schedule of the sharded orbit (bench.py --workload orbit, apps.render_views): uneven blocks (V odd),
    ragged last rounds, a rank that runs out of frames before the other, more rounds than collectives in flight."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611 + V * 7 + chunk
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, V, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, rounds), (1, True, rounds)]


@pytest.mark.parametrize("V,chunk", [(512, 8), (500, 8), (5, 8), (64, 3)])
def test_chunked_overlapped_gather_world8(V, chunk):
    """The TARGET's rank count (BASELINE config 4: one node, 8 GPUs), on gloo: the config-4 orbit itself (512 frames in chunks of 8:
    the staging tensor is [8 ranks, 8 frames, ...]), a frame count the ranks do not divide (500: blocks of 63 and a last block of 59,
    ragged last rounds on every rank), FEWER FRAMES THAN RANKS (5: ranks 5-7 own nothing and still have to join every collective),
    and a chunk that divides nothing.  Every rank must end with every frame, in view order, having rendered its own block once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from nerffaceediting_amd.launch import free_port
    port = free_port()
    world = 8
    rounds = -(-(-(-V // world)) // chunk)
    procs = [ctx.Process(target=_chunk_worker, args=(r, world, port, V, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True, rounds) for r in range(world)]


def test_all_gather_frames_world8_with_empty_ranks():
    """sharding.all_gather_frames / _async (bench.py's per-step frame exchange) at 8 ranks with 5 views: three ranks contribute nothing."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from nerffaceediting_amd.launch import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, 5, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(8)]


def test_chunked_gather_single_process():
    g = sharding.ChunkedFrameGather(5, 2, (1,), torch.float32, torch.device("cpu"))
    assert g.rounds() == 3 and not g.active
    for k in range(g.rounds()):
        s, e = g.local_slice(k)
        g.submit(k, torch.arange(s, e, dtype=torch.float32).reshape(-1, 1))
    assert g.finish().reshape(-1).tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]


def test_all_gather_single_process_is_identity():
    x = torch.arange(24.0).reshape(2, 3, 2, 2)
    assert sharding.all_gather_frames(x, 2) is x
    w, y = sharding.all_gather_frames_async(x, 2)
    assert w is None and y is x


# ---- bench.py's own launch path: `python bench.py --gpus N` with no launcher around it ----
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text
    return json.loads(lines[0])


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_bench_self_launches_its_ranks_on_gloo():
    """`python bench.py --gpus 2` starts two ranks itself (children, before anything touches a GPU), the ranks form a process
    group, the chunked frame exchange of the config-4 job runs (ragged blocks: 7 frames, chunk 3) and exactly one JSON line
    comes back, stating what the process group reported."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "exchange", "--backend", "gloo",
                        "--orbit-frames", "7", "--steps", "2"], capture_output=True, text=True, timeout=120, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["frames_ok"] is True
    assert line["distributed"] == {"backend": "gloo", "world_size": 2, "launcher": "self"}


@pytest.mark.parametrize("frames", [512, 5])
def test_bench_self_launches_eight_ranks_on_gloo(frames):
    """`python bench.py --gpus 8 --workload exchange --backend gloo`: the launch + frame-exchange plumbing of the config-4 job at the
    target's rank count - the 512-frame orbit, and 5 frames over 8 ranks (empty ranks).  No rendering, no scaling number: no 8-GPU
    node has been available in any round."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "exchange", "--backend", "gloo",
                        "--orbit-frames", str(frames), "--steps", "1"], capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 8 and line["frames_ok"] is True and line["config"]["frames"] == frames
    assert line["distributed"] == {"backend": "gloo", "world_size": 8, "launcher": "self"}


def test_bench_under_torch_distributed_run_on_gloo():
    """The driver's multi-GPU form keeps working: torch.distributed.run sets the rank environment, bench.py must not re-launch."""
    from nerffaceediting_amd.launch import free_port
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "exchange",
                        "--backend", "gloo", "--orbit-frames", "5", "--steps", "1"], capture_output=True, text=True, timeout=180, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["distributed"]["world_size"] == 2 and line["distributed"]["launcher"] == "torch.distributed.run"


def test_bench_gpus_mismatch_is_refused():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "exchange", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_spawn_ranks_reports_a_failed_rank(tmp_path):
    """A rank that dies makes the launch fail (non-zero) and the surviving ranks are stopped instead of waiting forever."""
    from nerffaceediting_amd import launch
    script = tmp_path / "ranks.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "if r == 1:\n    sys.exit(3)\n"
                      "if r == 0:\n    print('hello from rank 0', flush=True)\n"
                      "time.sleep(60)\n")
    import io
    import time as _t
    buf = io.StringIO()
    t0 = _t.time()
    rc, text = launch.spawn_ranks(str(script), [], 3, stdout=buf)
    assert rc == 3 and _t.time() - t0 < 30
    assert "hello from rank 0" in text and text == buf.getvalue()
    ok = tmp_path / "ok.py"
    ok.write_text("import os\nprint('rank', os.environ['RANK'], flush=True)\n")
    rc, text = launch.spawn_ranks(str(ok), [], 2, stdout=io.StringIO())
    assert rc == 0 and text == "rank 0\n"


def test_a_terminated_launcher_does_not_orphan_its_ranks(tmp_path):
    """SIGTERM (or Ctrl-C) in the parent of `spawn_ranks` must stop the rank processes - on a GPU box they hold the
    devices and may sit in a collective forever.  A launcher process starts two ranks that write their PIDs and sleep; the launcher
    is sent SIGTERM; both rank PIDs must be gone shortly after."""
    import signal
    import time as _t
    ranks = tmp_path / "sleepers.py"
    ranks.write_text("import os, time\n"
                     f"open(os.path.join({str(tmp_path)!r}, 'pid' + os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                     "time.sleep(120)\n")
    parent = tmp_path / "parent.py"
    parent.write_text(f"import sys\nsys.path.insert(0, {ROOT!r})\nfrom nerffaceediting_amd import launch\n"
                      f"rc, _ = launch.spawn_ranks({str(ranks)!r}, [], 2)\nsys.exit(rc)\n")
    p = subprocess.Popen([sys.executable, str(parent)], env=_clean_env(), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        t0 = _t.time()
        while not all((tmp_path / f"pid{r}").exists() and (tmp_path / f"pid{r}").read_text() for r in (0, 1)):
            assert _t.time() - t0 < 60 and p.poll() is None
            _t.sleep(0.1)
        pids = [int((tmp_path / f"pid{r}").read_text()) for r in (0, 1)]
        p.send_signal(signal.SIGTERM)
        p.wait(30)

        def alive(pid):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                return False
            try:                                     # a zombie still answers kill(0): look at its state
                return open(f"/proc/{pid}/stat").read().split(") ")[1][0] != "Z"
            except OSError:
                return False
        t0 = _t.time()
        while any(alive(pid) for pid in pids) and _t.time() - t0 < 20:
            _t.sleep(0.1)
        assert not any(alive(pid) for pid in pids), "rank processes survived their launcher"
    finally:
        if p.poll() is None:
            p.kill()


def test_self_launcher_retries_when_the_rendezvous_port_was_taken(tmp_path):
    """launch.free_port() releases its port before rank 0 binds it again; a rank that finds it taken exits with RENDEZVOUS_BUSY
    (launch.init_process_group) and spawn_ranks starts every rank again on a fresh port."""
    import io
    from nerffaceediting_amd import launch
    flag = tmp_path / "first_attempt_done"
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys\n"
        f"flag = {str(flag)!r}\n"
        "if not os.path.exists(flag):\n"
        "    if os.environ['RANK'] == '0':\n"
        "        open(flag, 'w').write(os.environ['MASTER_PORT'])\n"
        f"        os._exit({launch.RENDEZVOUS_BUSY})\n"
        "    import time; time.sleep(30)\n"                       # the other rank would wait in the rendezvous: the parent must stop it
        "print('port', os.environ['MASTER_PORT'], 'rank', os.environ['RANK'])\n")
    out = io.StringIO()
    rc, text = launch.spawn_ranks(str(script), [], 2, timeout=60, stdout=out)
    assert rc == 0, (rc, text)
    first = flag.read_text()
    assert "rank 0" in text and f"port {first} " not in text          # second attempt, different port
