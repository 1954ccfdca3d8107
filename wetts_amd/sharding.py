"""Multi-GPU layout of the hot path: utterances are independent, so they shard across ranks with
NO collective in the decode loop.  The only communication is one broadcast of the
folded weight blob at start-up (RCCL over xGMI when the backend is "nccl"; gloo in CPU tests) and,
optionally, a gather of per-rank results on the host side.

The reference has no inference-side distributed code to mirror (inference.py:51-53 pins one
device; Triton uses replicas, runtime/gpu_triton/model_repo/generator/config.pbtxt:48-53).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process per GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


# process-group timeout (c10d's own default is 10-30 minutes: far too long for a bench).  It sits ABOVE the longest phase
# deadline bench.py arms (setup: 300 s) so that PhaseMonitor -- which can still print the partial line -- fires first;
# PhaseMonitor.enter clamps any longer request to this minus a margin
DEFAULT_TIMEOUT_S = 420.0


DIAG_PATH = os.path.join("gpurun_out", "scale_diag.json")  # (WETTS_SCALE_DIAG: another path -- the tests' tmp dir)


def diag(event, **fields):
    """Appends one JSON line to gpurun_out/scale_diag.json (relative to the working directory -- the repo root on the GPU
    box, the directory the driver pulls back): what every rank of a multi-GPU run says on stderr -- which device it bound,
    which phase failed -- plus rank 0's result or partial line, in ONE file that survives the run.  The first N > 1 run is
    the driver's and nobody can rehearse it; its post-mortem should not depend on captured stderr.  One small O_APPEND
    write per event (atomic between the ranks of a node); never raises."""
    import json
    import time
    try:
        if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
            return
        path = os.environ.get("WETTS_SCALE_DIAG", DIAG_PATH)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        line = json.dumps(dict(event=event, rank=int(os.environ.get("RANK", "0")),
                               world=int(os.environ.get("WORLD_SIZE", "1")), pid=os.getpid(),
                               t=round(time.time(), 3), **fields)) + "\n"
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
        try:
            os.write(fd, line.encode())
        finally:
            os.close(fd)
    except Exception:
        pass


def describe_device(local_rank=None):
    """One line naming the HIP device this rank is bound to (index, name, PCI address where torch exposes it): what a
    failed multi-GPU start-up needs on stderr to tell WHICH GPU did not come up."""
    if not torch.cuda.is_available():
        return "no HIP device (CPU)"
    i = torch.cuda.current_device() if local_rank is None else local_rank
    try:
        pr = torch.cuda.get_device_properties(i)
        pci = ":".join(f"{int(getattr(pr, k)):02x}" for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")
                       if hasattr(pr, k))
        return f"cuda:{i} {pr.name} pci {pci or 'n/a'} {pr.total_memory >> 30} GiB"
    except Exception as e:  # never let the diagnostics be the failure
        return f"cuda:{i} (properties unavailable: {e})"


def init_process_group(backend=None, timeout_s=None, force=False):
    """Initialises torch.distributed from MASTER_ADDR/MASTER_PORT/RANK/WORLD_SIZE if needed.

    `nccl` (= RCCL): the process group is BOUND to this rank's current device (`device_id`) -- c10d otherwise guesses
    the device from the global rank ("can cause a hang if rank to GPU mapping is heterogeneous") and creates the
    communicator lazily inside the first collective; with `device_id` it is created here, eagerly, under `timeout_s`
    (default 420 s -- above every phase deadline of PhaseMonitor; WETTS_DIST_TIMEOUT_S overrides).  NCCL_DEBUG defaults to WARN so that a failing rank explains
    itself on stderr; every rank names its device there first.  `force`: also at world size 1 (the RCCL start-up test
    of a 1-GPU box)."""
    import datetime
    import sys
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("WETTS_DIST_BACKEND") or \
                ("nccl" if torch.cuda.is_available() else "gloo")
        if timeout_s is None:
            timeout_s = float(os.environ.get("WETTS_DIST_TIMEOUT_S", DEFAULT_TIMEOUT_S))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # see launch_ranks
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        sys.stderr.write(f"[wetts rank {rank}/{world}] {backend} on {describe_device()} "
                         f"(timeout {timeout_s:.0f} s, NCCL_DEBUG={os.environ['NCCL_DEBUG']})\n")
        sys.stderr.flush()
        diag("init_process_group", backend=backend, device=describe_device(), timeout_s=timeout_s)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return rank, local_rank, world


class PhaseMonitor:
    """Deadline watchdog of a multi-rank job that nobody can rehearse (no multi-GPU node before the driver's run).

    The job moves through named phases (rendezvous, broadcast, warm-up, timed loop, reductions).  Entering a phase
    (a) counts this rank in on the rendezvous store (`wetts/at/<phase>`, a TCPStore counter: no collective, no device)
    and (b) arms a deadline.  A daemon thread watches the deadline: when a phase overruns -- a rank that never came up,
    a hung collective, a hung GPU -- it reads how many ranks reached the phase, calls `on_expire(phase, ranks_seen)`
    (rank 0 prints its partial result line there) and ends the process with exit code 4, instead of leaving the
    launcher to wait for c10d's watchdog or for ever.  The deadlines of the collective phases are kept BELOW the
    process-group timeout so that this monitor, which can still report, fires before NCCL's abort."""

    EXIT_CODE = 4

    def __init__(self, rank, world, on_expire=None, default_deadline_s=None):
        import threading
        self.rank, self.world = rank, world
        self.on_expire = on_expire
        env = os.environ.get("WETTS_BENCH_PHASE_DEADLINE_S")
        self.override = float(env) if env else default_deadline_s
        self.phase, self.deadline = None, None
        self.store = None
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self._thr = None
        self._sig_r = self._sig_w = None
        self._sig_prev, self._sig_prev_fd = None, -1
        if world > 1:
            self._watch_sigterm()
            self._thr = threading.Thread(target=self._watch, name="wetts-phase-monitor", daemon=True)
            self._thr.start()

    def _watch_sigterm(self):
        """torchrun answers one rank's crash by sending SIGTERM to the others.  A Python-level handler would only run
        once the main thread returns from the collective it is blocked in (never, then); the wake-up fd is written by
        the C-level handler at once, and the monitor thread reads it."""
        import signal
        import threading
        if threading.current_thread() is not threading.main_thread():
            return
        try:
            r, w = os.pipe()
            os.set_blocking(w, False)
            os.set_blocking(r, False)
            self._sig_prev = signal.signal(signal.SIGTERM, lambda *_: None)
            self._sig_prev_fd = signal.set_wakeup_fd(w, warn_on_full_buffer=False)
            self._sig_r, self._sig_w = r, w
        except (OSError, ValueError):
            self._sig_r = None

    def attach_store(self):
        """After init_process_group: the default rendezvous store carries the roll call."""
        if dist.is_initialized():
            try:
                self.store = dist.distributed_c10d._get_default_store()
            except Exception:
                self.store = None
        return self

    def enter(self, phase, deadline_s):
        import time
        # keep what the docstring promises: this monitor (which can still report) fires BEFORE c10d's own watchdog
        # aborts a hung collective -- a phase may ask for more than the process-group timeout allows (bench.py's timed
        # phase grows with --steps), so the deadline is clamped to that timeout minus a margin (non-zero ranks add 3 s)
        pg = float(os.environ.get("WETTS_DIST_TIMEOUT_S", DEFAULT_TIMEOUT_S))
        if self.world > 1 and not self.override:
            deadline_s = min(float(deadline_s), max(5.0, pg - 10.0))
        with self._lock:
            self.phase = phase
            # (the reporting rank fires first: a non-zero rank that left earlier would only make the launcher tear rank 0 down)
            self.deadline = time.monotonic() + (self.override if self.override else deadline_s) + (0.0 if self.rank == 0 else 3.0)
        if self.store is not None:
            try:
                self.store.add("wetts/at/" + phase, 1)
            except Exception:
                pass

    def ranks_at(self, phase):
        """How many ranks have entered `phase` (None when the store cannot be asked)."""
        if self.store is None:
            return None
        try:
            return int(self.store.add("wetts/at/" + phase, 0))
        except Exception:
            return None

    def done(self):
        """Stops the watcher and puts SIGTERM back the way it was found: from here on nobody reads the wake-up pipe, so
        a handler left in place would swallow the launcher's (or a harness `timeout`'s) SIGTERM for the rest of the
        process -- rank 0 still formats its line and may run the long CPU baseline after this."""
        import signal
        import threading
        self._stop.set()
        if self._thr is not None and self._thr is not threading.current_thread():
            self._thr.join(timeout=2.0)
        if self._sig_r is not None and threading.current_thread() is threading.main_thread():
            try:
                signal.set_wakeup_fd(self._sig_prev_fd if self._sig_prev_fd is not None else -1)
                signal.signal(signal.SIGTERM, self._sig_prev if self._sig_prev is not None else signal.SIG_DFL)
            except (OSError, ValueError):
                pass
            for fd in (self._sig_r, self._sig_w):
                try:
                    os.close(fd)
                except OSError:
                    pass
            self._sig_r = self._sig_w = None

    def expire_now(self, why):
        """The same exit path for a failure the main thread caught itself (an exception out of a collective)."""
        self._fire(self.phase or "start-up", why)

    def _fire(self, phase, why):
        import sys
        seen = self.ranks_at(phase)
        sys.stderr.write(f"[wetts rank {self.rank}/{self.world}] {why} in phase '{phase}' on {describe_device()}; "
                         f"ranks that reached it: {seen if seen is not None else 'unknown'} of {self.world}\n")
        sys.stderr.flush()
        diag("phase_failed", phase=phase, why=why, ranks_seen=seen, device=describe_device())
        try:
            if self.on_expire is not None:
                self.on_expire(phase, seen, why)
        finally:
            sys.stdout.flush()
            os._exit(self.EXIT_CODE)

    def _watch(self):
        import select
        import signal
        import time
        while not self._stop.is_set():
            if self._sig_r is not None:
                rd, _, _ = select.select([self._sig_r], [], [], 0.25)
                if rd:
                    try:
                        got = os.read(self._sig_r, 64)
                    except OSError:
                        got = b""
                    if bytes([signal.SIGTERM]) in got and not self._stop.is_set():
                        self._fire(self.phase or "start-up", "SIGTERM (the launcher is tearing the job down: another "
                                                             "rank failed)")
            else:
                self._stop.wait(0.25)
            with self._lock:
                phase, deadline = self.phase, self.deadline
            if deadline is not None and time.monotonic() > deadline and not self._stop.is_set():
                self._fire(phase, "deadline passed")


def launch_ranks(n, script, argv, require_gpus=True):
    """Starts `n` ranks of `script argv...` on THIS node, one process per GPU, by re-executing it
    under torch.distributed.run with a 127.0.0.1 rendezvous on a free port; returns the exit code.
    This is what `python bench.py --gpus N` does when it is not already running under torchrun.
    With `require_gpus` it refuses (code 3, message on stderr, nothing launched) when the node
    exposes fewer than `n` HIP devices -- a smaller job must never run under a larger label."""
    import socket
    import subprocess
    import sys
    if require_gpus:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < n:
            sys.stderr.write(f"{os.path.basename(script)}: {n} ranks requested but this node exposes "
                             f"{n_dev} HIP device(s); refusing to run a smaller job under a larger "
                             "label\n")
            return 3
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)]
    env = dict(os.environ)
    # the deployment image exports HSA_ENABLE_IPC_MODE_LEGACY=0 (its host driver only supports dmabuf IPC; without
    # it RCCL fails in hipIpcGetMemHandle); a child environment built here must carry the same setting, and an
    # operator's own value wins
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("NCCL_DEBUG", "WARN")  # a rank that fails explains itself on stderr
    return subprocess.call(cmd + list(argv), env=env)


def shard_utterances(lengths, world_size):
    """Longest-processing-time-first deal of utterances to ranks.

    lengths: per-utterance cost proxy (phoneme counts; frames are ~proportional).
    Returns a list (len world_size) of index lists; every index appears exactly once; ranks'
    total cost differs by at most one utterance's cost from the greedy optimum.  Within a rank
    the indices are sorted by length (descending) so consecutive batches pad little."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    counts = [0] * world_size
    shards = [[] for _ in range(world_size)]
    # keep batch counts equal (weak scaling: B/rank fixed): n = q * world + rem, so `rem` ranks take q + 1 utterances and
    # the others q -- a rank may only take its (q + 1)-th while fewer than `rem` ranks have one (a plain cap of
    # ceil(n / world) let two ranks fill up and left the third two short: 15 / 15 / 13 for 43 over 3)
    q, rem = divmod(len(lengths), world_size)
    for i in order:
        full = sum(1 for r in range(world_size) if counts[r] > q)
        cands = [r for r in range(world_size) if counts[r] < q or (counts[r] == q and full < rem)]
        r = min(cands, key=lambda c: (loads[c], c))
        shards[r].append(i)
        loads[r] += int(lengths[i])
        counts[r] += 1
    return shards


def broadcast_blob(blob, src=0):
    """One broadcast of the folded weight blob (float32, contiguous) from rank `src`.
    ~123 MB for v1; ring/tree over xGMI is per-link bound (~153 GB/s/link) => ~1-2 ms."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def gather_objects(obj, dst=0):
    """Host-side gather of small per-rank results (lengths, timings)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


def unshard(shards, per_rank_results):
    """Inverse of shard_utterances for host-side results: puts results back in input order."""
    n = sum(len(s) for s in shards)
    out = [None] * n
    for idxs, res in zip(shards, per_rank_results):
        for i, r in zip(idxs, res):
            out[i] = r
    return out
