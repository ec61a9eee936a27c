"""Self-launch of a one-process-per-GPU job, the counterpart of the reference's own multi-process entry
(`train.py:98-103`: `torch.multiprocessing.spawn(fn=subprocess_fn, nprocs=c.num_gpus)`; rendezvous `train.py:37-43`).

`spawn_ranks(script, argv, nprocs)` starts `nprocs` FRESH interpreter processes of `script`, one per rank, with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment (what `torch.distributed.run` would set),
so the same script works under either launcher.  The parent never touches the GPU and never replaces itself: the ranks
are children, the parent waits for them, forwards rank 0's standard output and returns non-zero if any rank failed
(the remaining ranks are then terminated by PID — a rank that died before a collective would otherwise leave the others
waiting in it).  No torch import here: this must stay safe to call before anything initialises HIP.
"""
import os
import re
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


RENDEZVOUS_BUSY = 98          # exit code of a rank whose rendezvous found the port taken (EADDRINUSE): spawn_ranks retries on another port


def init_process_group(dist, backend, **kw):
    """dist.init_process_group for a rank started by spawn_ranks: free_port() has released the port before rank 0 binds it again, so
    another process can take it in between.  That shows as 'address already in use' in rank 0; the rank then exits with
    RENDEZVOUS_BUSY and the parent starts all ranks again on a fresh port (three attempts)."""
    # RCCL prints a version banner ("RCCL version : ...", five lines) through C stdio when its first communicator is made (eagerly here:
    # device_id=).  On a pipe that text sits in libc's buffer and reaches stdout at exit - behind the one JSON line a bench rank prints.
    # File descriptor 1 points at stderr while the communicator is created and libc's buffers are flushed before it is restored.
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group(backend, **kw)
        if backend == "nccl" and "device_id" in kw:
            import torch
            dist.barrier(device_ids=[kw["device_id"].index])          # the communicator exists (and has said what it has to say) before fd 1 returns
            torch.cuda.synchronize(kw["device_id"])
    except Exception as e:          # torch raises DistNetworkError / RuntimeError depending on the store
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
        saved = None
        if os.environ.get("NFE_LAUNCHER") == "self" and _address_in_use(e):
            print(f"[launch] rank {os.environ.get('RANK')}: rendezvous port {os.environ.get('MASTER_PORT')} is taken", file=sys.stderr)
            sys.stderr.flush()
            os._exit(RENDEZVOUS_BUSY)
        raise
    finally:
        if saved is not None:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved, 1)
            os.close(saved)


def _address_in_use(exc):
    """EADDRINUSE from the rendezvous store, recognised by errno where the exception (or one it wraps) carries one, by type name
    (torch.distributed.DistNetworkError) together with the message otherwise - the message alone differs between torch versions."""
    import errno
    seen = set()
    e = exc
    while e is not None and id(e) not in seen:
        seen.add(id(e))
        if getattr(e, "errno", None) == errno.EADDRINUSE:
            return True
        text = str(e).lower()
        if "address already in use" in text or "eaddrinuse" in text or re.search(r"\b(errno|error code):\s*98\b", text):     # anchored: not "errno: 980"
            return True
        e = e.__cause__ or e.__context__
    return False


def launched_by_a_launcher(env=None):
    """True inside a rank process (torch.distributed.run or spawn_ranks has set WORLD_SIZE)."""
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def spawn_ranks(script, argv, nprocs, env=None, timeout=None, stdout=None):
    """Run `python script *argv` as `nprocs` ranks on this node.  Returns (exit_code, rank0_stdout_text).
    Rank 0's stdout is captured (and echoed to `stdout`, default sys.stdout, line by line); the other ranks' stdout goes to
    this process's stderr so that the caller still sees exactly one result line on stdout."""
    stdout = sys.stdout if stdout is None else stdout
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(nprocs), LOCAL_WORLD_SIZE=str(nprocs), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
                NFE_LAUNCHER="self")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    procs = []

    def stop_all():
        """terminate, then kill, every rank still alive - by the exact PIDs started here, never by a pattern"""
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        for p in alive:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()

    # SIGTERM / SIGINT in the parent must not orphan ranks that hold the GPUs (possibly blocked in a collective): turn the signal
    # into an exception here, and let the finally clause below stop the children.  (Handlers can only be set from the main thread.)
    restore = {}
    if threading.current_thread() is threading.main_thread():
        def on_signal(signum, frame):
            raise KeyboardInterrupt(f"signal {signum}")
        for sig in (signal.SIGTERM, signal.SIGINT):
            restore[sig] = signal.signal(sig, on_signal)
    deadline = None if timeout is None else time.time() + timeout          # ONE deadline for all attempts (it restarted per attempt)
    try:
        for attempt in range(3):
            rc, text = _run_ranks(script, argv, nprocs, base, procs, stop_all, deadline, timeout, stdout)
            if rc != RENDEZVOUS_BUSY or attempt == 2:
                return rc, text
            stop_all()
            del procs[:]
            base["MASTER_PORT"] = str(free_port())
            print(f"[launch] rendezvous port was taken; retrying on {base['MASTER_PORT']}", file=sys.stderr)
    finally:
        stop_all()
        for sig, old in restore.items():
            signal.signal(sig, old)


def _run_ranks(script, argv, nprocs, base, procs, stop_all, deadline, timeout, stdout):
    """One attempt, against the caller's single deadline.  Rank 0's stdout is buffered until the attempt has ended and is echoed only
    if it did not end in RENDEZVOUS_BUSY, so that nothing of an attempt that is going to be replayed reaches the caller's stdout (the
    contract is exactly one result line; a rank's own output is one line at its very end, so nothing is lost by not streaming it)."""
    for r in range(nprocs):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    captured = []

    def pump():
        for line in procs[0].stdout:
            captured.append(line)
            if len(captured) > 200000:          # a chatty rank must not grow the buffer without bound: keep the newest half
                del captured[:100000]
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    rc = 0
    live = set(range(nprocs))
    try:
        return _watch_ranks(procs, live, rc, stop_all, deadline, timeout, t, captured, stdout)
    except BaseException:
        # interrupted (SIGTERM / Ctrl-C become KeyboardInterrupt in spawn_ranks): what rank 0 has printed so far is still the caller's -
        # partial results must not vanish with the buffer
        t.join(1)
        for line in captured:
            stdout.write(line)
        stdout.flush()
        raise


def _watch_ranks(procs, live, rc, stop_all, deadline, timeout, t, captured, stdout):
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"[launch] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
        if rc != 0 or (deadline is not None and time.time() > deadline):
            if rc == 0:
                rc = 124
                print(f"[launch] timeout after {timeout} s; stopping all ranks", file=sys.stderr)
            stop_all()                                              # exact PIDs we started, never a pattern
            live.clear()
        else:
            time.sleep(0.05)
    t.join(5)
    if rc != RENDEZVOUS_BUSY:                  # a busy rendezvous is replayed on another port: its output is dropped, not echoed
        for line in captured:
            stdout.write(line)
        stdout.flush()
    return rc, "".join(captured)
