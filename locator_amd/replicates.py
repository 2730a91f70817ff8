"""Sharding of independent replicate fits (--windows, --bootstrap) over the GPUs of one node.

Reference: the window loop (/root/reference/locator/locator.py:531-583) and the bootstrap loop
(:635-681) are strictly sequential `for` loops on one device.  The fits are independent once the
parts that advance the global NumPy stream (splits, reseeds, site orders) have been drawn, so here
the parent draws those sequentially in reference order and the fits run one model per GPU.

  * no collective on the data path: a unit's inputs go to exactly one worker, its outputs are the
    files it writes plus a small result record;
  * per-unit randomness (init, shuffles, dropout) is keyed by the unit's replicate index, never by
    the worker, so 1-GPU and 8-GPU runs write identical files;
  * dynamic queue (run_units) because early stopping makes unit durations vary several-fold; a
    static round-robin (shard_static / run_units_distributed) is provided for launches that are
    already one-process-per-GPU (torch.distributed.run), where ranks cannot share a queue;
  * a failed unit is reported and does not stop its siblings - including a unit whose WORKER PROCESS dies
    (GPU fault, OOM kill): the parent watches worker liveness, records the lost unit and starts a fresh process;
  * data common to all units (the bootstrap genotype matrix) is handed to the workers through shared memory.
"""
from __future__ import annotations

import traceback


def shard_static(n_units, rank, world):
    """Unit indices of `rank` under round-robin assignment."""
    return list(range(rank, n_units, world))


def _run_one(fit_fn, unit, shared, args, device):
    u = dict(shared or {})
    u.update(unit)
    u["args"] = args
    try:
        return fit_fn(u, device=device)
    except Exception as e:                                   # noqa: BLE001 — reported, siblings continue
        return {"name": unit.get("name", "?"), "error": f"{type(e).__name__}: {e}",
                "traceback": traceback.format_exc()}


# ---------------------------------------------------------------------------------------------------------------
# shared data: large NumPy arrays travel through POSIX shared memory (one copy for all workers, attached not
# pickled), everything else is pickled once per worker as before
# ---------------------------------------------------------------------------------------------------------------
_SHM_MIN_BYTES = 1 << 20


def _share(shared):
    """Split `shared` into (small picklable dict, descriptors of arrays placed in shared memory, handles to unlink)."""
    import numpy as np
    from multiprocessing import shared_memory
    small, descs, handles = {}, {}, []
    for k, v in (shared or {}).items():
        if isinstance(v, np.ndarray) and v.nbytes >= _SHM_MIN_BYTES and v.dtype != object:
            shm = shared_memory.SharedMemory(create=True, size=v.nbytes)
            np.ndarray(v.shape, v.dtype, buffer=shm.buf)[...] = v
            descs[k] = (shm.name, v.shape, v.dtype.str)
            handles.append(shm)
        else:
            small[k] = v
    return small, descs, handles


def _attach(small, descs):
    import numpy as np
    from multiprocessing import shared_memory
    out, keep = dict(small), []
    for k, (name, shape, dtype) in descs.items():
        shm = shared_memory.SharedMemory(name=name)
        keep.append(shm)
        a = np.ndarray(shape, np.dtype(dtype), buffer=shm.buf)
        a.flags.writeable = False
        out[k] = a
    return out, keep


# Fit threads of THIS process (set by the worker / the in-process path before the first fit starts).  Fits that share a
# process capture their HIP graphs under train.DEVICE_LOCK (round 5): capture is process-sensitive on this platform even in
# thread-local mode - another thread's device-wide wait, or a graph / event destroyed while a capture is open, aborts the
# process - so everything a fit does on the device outside its epoch loop holds that lock.  locator.train_network reads this
# count to decide whether capture pays at the fit's SNP count (GRAPH_MAX_SNPS_IN_FIT_THREADS).
_FIT_THREADS = 1


def fit_threads_in_process():
    return _FIT_THREADS


def _fit_stream(device):
    """A HIP stream of this thread's own (several fits share one process and GPU, each on its own thread and stream)."""
    import threading
    tls = _fit_stream.__dict__.setdefault("tls", threading.local())
    if getattr(tls, "stream", None) is None:
        import torch
        tls.stream = torch.cuda.Stream(device=device) if str(device).startswith("cuda") and torch.cuda.is_available() else False
    return tls.stream


def _run_on_own_stream(fit_fn, unit, shared, args, device, prepare):
    """prepare + fit of one unit on the calling thread's stream."""
    stream = _fit_stream(device)
    try:
        if stream:
            import torch
            with torch.cuda.stream(stream):
                if prepare is not None:
                    unit = prepare(unit)
                r = _run_one(fit_fn, unit, shared, args, device)
                stream.synchronize()
                return r
        if prepare is not None:
            unit = prepare(unit)
        return _run_one(fit_fn, unit, shared, args, device)
    except Exception as e:                                   # noqa: BLE001 - e.g. prepare() raising
        return {"name": unit.get("name", "?") if isinstance(unit, dict) else "?",
                "error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()}


def _worker(gpu, fit_fn, args, prepare, host_prepare, conn, t_parent, fit_threads=1, env=None):
    """Worker process on its own duplex pipe (no queue or lock is shared between workers, so one that is killed
    cannot wedge the others).  Start-up (import torch, device context, HIP library) happens right away - the parent
    spawns the pool BEFORE its own prologue so that the two overlap.  Messages in: ("shared", small, descs) once,
    ("unit", index, unit) per unit, None to exit.  A loader thread receives them and runs `host_prepare` (the
    zarr slice + filters of a window) for the NEXT unit while the main thread fits the current one; messages out:
    ("ready", info), ("start", index), ("done", record) - only the main thread sends, only the loader receives."""
    import queue
    import threading
    import time
    t0 = time.time()
    numa = None
    try:
        _bind_worker_to_gpu(gpu, env)                           # HIP_VISIBLE_DEVICES, before this process's first HIP call
        import torch
        if torch.cuda.is_available():
            device = "cuda:0"                                   # the only device this process sees
            torch.cuda.set_device(0)
            torch.zeros(1, device=device)                       # device context now, not inside the first fit
            from . import _lib
            _lib.load()
            numa = _bind_worker_to_numa_node(device)
        else:               # scheduler tests on CPU; a real fit_fn raises on this device (no CPU fallback)
            device = "cpu"
    except Exception as e:                                   # noqa: BLE001 - the parent turns it into error records
        conn.send(("dead", f"worker start-up failed: {type(e).__name__}: {e}"))
        return
    conn.send(("ready", {"startup_seconds": time.time() - t0, "spawn_seconds": t0 - t_parent, "ready_at": time.time(),
                         "cpu_affinity": numa}))
    todo = queue.Queue()
    box = {"shared": {}, "keep": []}

    def loader():
        # Anything that goes wrong OUTSIDE a unit's own host work (a shared-memory segment that cannot be attached, a
        # message that does not unpickle) must not leave the main thread blocked on `todo` in a live process: the parent
        # would see neither end-of-file nor a dead process.  It is handed over as a fatal item; the main thread reports
        # ("dead", why) - only the main thread sends - and the process exits, so the parent buries it like any other loss.
        try:
            while True:
                try:
                    item = conn.recv()
                except (EOFError, OSError):
                    item = None
                if item is None:
                    todo.put(None)
                    return
                if item[0] == "shared":
                    box["shared"], box["keep"] = _attach(item[1], item[2])
                    continue
                _, idx, unit = item
                t1, err = time.time(), None
                try:
                    if host_prepare is not None:
                        unit = host_prepare(unit, args)
                except Exception as e:                       # noqa: BLE001
                    err = {"name": unit.get("name", "?") if isinstance(unit, dict) else "?",
                           "error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()}
                todo.put((idx, unit, err, time.time() - t1))
        except BaseException as e:                           # noqa: BLE001
            todo.put(("fatal", f"worker loader thread failed: {type(e).__name__}: {e}"))

    threading.Thread(target=loader, daemon=True).start()
    send_lock = threading.Lock()

    def send(msg):
        with send_lock:
            conn.send(msg)

    def fit_loop():
        # `fit_threads` of these run side by side: one process, one device context, one start-up - each fit on its own
        # stream (a fit alternates between an HBM-bound phase and a latency-bound one; two of them interleave)
        while True:
            item = todo.get()
            if item is None or item[0] == "fatal":
                todo.put(item)                               # the siblings see it too
                return
            idx, unit, err, t_host = item
            # ("start", i) opens the unit's --unit_timeout clock in the parent.  A fit function that queues its units for
            # admission inside the process (locator._fit_unit: fits_per_gpu / SNP-count budget) sends it itself through
            # unit["on_admitted"] once the unit is admitted - the wait for a sibling's whole fit is not the unit's time;
            # any other fit function starts now.  Sent exactly once either way, before ("done", r).
            started = []

            def admitted(idx=idx, started=started):
                if not started:
                    started.append(1)
                    send(("start", idx))

            if getattr(fit_fn, "reports_admission", False) and err is None and isinstance(unit, dict):
                unit = dict(unit, on_admitted=admitted)
            else:
                admitted()
            t1 = time.time()
            r = err if err is not None else _run_on_own_stream(fit_fn, unit, box["shared"], args, device, prepare)
            admitted()
            r["unit_index"], r["gpu"] = idx, gpu
            r["host_prepare_seconds"], r["worker_seconds"] = t_host, time.time() - t1
            send(("done", r))

    global _FIT_THREADS
    _FIT_THREADS = max(1, int(fit_threads))
    fitters = [threading.Thread(target=fit_loop, daemon=True) for _ in range(max(1, int(fit_threads)) - 1)]
    for th in fitters:
        th.start()
    fit_loop()
    for th in fitters:
        th.join()
    last = todo.get()
    if last is not None and last[0] == "fatal":
        try:
            send(("dead", last[1]))
        except (OSError, ValueError):
            pass
    for shm in box["keep"]:
        shm.close()


def visible_gpus():
    """GPUs this process may use, counted WITHOUT importing torch or touching the GPU when the kernel driver's topology is
    readable (/sys/class/kfd: a node with simd_count > 0 whose render node this process may open is a GPU; HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES narrow it): the
    parent of a replicate run starts its workers first thing, and `import torch` alone is a second of that critical path."""
    import glob
    import os
    try:
        n = 0
        for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            with open(f) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue                                     # a CPU node
            # a container may see the whole host's topology but only some render nodes: a GPU counts when its device file does
            minor = int(props.get("drm_render_minor", "-1"))
            n += minor < 0 or os.access(f"/dev/dri/renderD{minor}", os.R_OK | os.W_OK)
        if n > 0:
            for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
                v = os.environ.get(var)
                if v is not None:
                    n = min(n, len([t for t in v.split(",") if t.strip() not in ("", "-1")]))
            return n
    except (OSError, ValueError):
        pass
    import torch
    return torch.cuda.device_count()


START_METHOD = "forkserver"       # how worker processes come to be: "forkserver" (default) or "spawn"
_PRELOAD = ["numpy", "torch", "locator_amd._lib", "locator_amd.net", "locator_amd.train", "locator_amd.replicates",
            "locator_amd.locator"]


def warm_start(method=None):
    """Start the fork server NOW (first thing in a replicate run, before the parent parses, imports or reads anything): a fresh
    interpreter that imports torch and this package ONCE - without any GPU call - and from then on forks every worker process on
    request.  A worker then costs a fork + one HIP context (about 0.3 s) instead of an interpreter start + `import torch`
    (1.0-1.1 s each, round 5), also when a lost worker is replaced in the middle of a run, and the server's imports overlap the
    parent's own start-up.  Never forks or execs a process that has touched the GPU: the server never does, and the parent
    starts it (an exec) before its own first GPU call.  No-op for "spawn"."""
    import multiprocessing as mp
    method = method or START_METHOD
    if method != "forkserver":
        return None
    try:
        ctx = mp.get_context("forkserver")
        ctx.set_forkserver_preload(list(_PRELOAD))
        from multiprocessing import forkserver
        forkserver.ensure_running()
        return ctx
    except (OSError, ValueError, ImportError) as e:          # no UNIX sockets / no fork server on this platform
        import sys
        print(f"replicate workers: fork server unavailable ({type(e).__name__}: {e}); spawning fresh interpreters instead",
              file=sys.stderr)
        return None


def _visible_env():
    """What the parent's environment says about visible devices, to be re-applied inside a worker: a forked worker inherits
    the FORK SERVER's environment (started before --gpu_number was parsed), not the parent's."""
    import os
    return {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")}


def _bind_worker_to_gpu(gpu, env=None):
    """Before the worker's first HIP call: make GPU `gpu` (an index into what the parent sees) the ONLY device this process
    sees - it is then cuda:0 here.  One process per GPU the ROCm way: no context is ever created on a
    sibling's device, and a stray `cuda:0` default cannot land on another worker's GPU."""
    import os
    for k, v in (env or {}).items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    base = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    ids = [t.strip() for t in base.split(",") if t.strip()] if base else None
    os.environ["HIP_VISIBLE_DEVICES"] = ids[gpu] if ids and gpu < len(ids) else str(gpu)
    os.environ.pop("CUDA_VISIBLE_DEVICES", None)            # (HIP honours both; one statement of the truth)


def _bind_worker_to_numa_node(device):
    """Best effort: keep the worker's threads on the CPUs local to its GPU (the PCI device's local_cpulist in sysfs).  Returns a
    short description for the pool's timeline, or None when the topology does not say."""
    import os
    try:
        import torch
        p = torch.cuda.get_device_properties(device)
        addr = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{addr}/local_cpulist") as fh:
            text = fh.read().strip()
        cpus = set()
        for part in text.split(","):
            if part:
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        have = os.sched_getaffinity(0)
        want = cpus & have
        if want and want != have:
            os.sched_setaffinity(0, want)
            return f"{addr}: {len(want)} of {len(have)} CPUs"
        return f"{addr}: all {len(have)} CPUs local"
    except Exception:                                        # noqa: BLE001 - affinity is a speed hint
        return None


class ReplicatePool:
    """The replicate scheduler: worker processes (one or more per GPU), dynamic dispatch, a timeline.

        pool = ReplicatePool(args, fit_fn, n_gpus=8, fits_per_gpu=2, host_prepare=load_window)
        pool.start()                     # BEFORE the parent's own prologue: spawn + import torch + device context overlap it
        ... parent prologue (draw splits, read sample table, ...) ...
        records = pool.run(units, shared)
        pool.close(); print(pool.summary())

    fits_per_gpu  worker processes per GPU.  A single fit alternates between an HBM-bound phase (layer 1) and a
                  latency-bound phase (hidden stack, 16 CUs); two fits on one GPU interleave them: measured 209k
                  vs 157k samples/s aggregate on the 1000 x 100k workload (1.33x), no further gain from a third.
    prepare       optional per-unit hook run in the worker's main thread right before fit_fn (device-side work)
    host_prepare  optional per-unit hook `f(unit, args) -> unit` run on a LOADER THREAD of the worker: the parent keeps
                  two units in flight per worker, so the host work of unit i + 1 (zarr slice + filters, 2 s per
                  150k-variant window) overlaps the fit of unit i instead of sitting on its critical path
    isolate       True: even a single worker is a separate process (crash isolation: a HIP abort in a fit costs that worker,
                  its units are retried on a fresh one); None = only when unit_timeout > 0 (a hung fit can be killed only
                  as a process); False = a single worker runs inside the calling process.
    unit_timeout  seconds a unit may spend in a worker after it reported ("start", i) - sent when the fit is ADMITTED, so time
                  queued behind a sibling fit of the same worker does not count -, or an otherwise idle worker may
                  spend on a unit's host work before that report; a worker that exceeds it is
                  killed (its exact process, never a pattern), the unit becomes an error record and a FRESH process
                  takes the slot - a hung (not dead) worker no longer blocks the run.  0 = no limit.

    Dynamic dispatch: the parent hands the next unit to whichever worker has room (early stopping makes unit durations
    vary several-fold).  A worker that dies (GPU fault, OOM kill, abort) does not hang the run: its pipe reports
    end-of-file, the parent records an error for the unit it was fitting, puts units it had only prefetched back in
    the queue, and starts a fresh process while units remain.  A send that fails because the worker vanished between
    two messages requeues the unit the same way."""

    def __init__(self, args, fit_fn, n_gpus=None, fits_per_gpu=1, prepare=None, host_prepare=None, log=print,
                 poll_s=1.0, unit_timeout=0.0, max_workers=None, procs_per_gpu=None, isolate=None, start_method=None):
        """fits_per_gpu concurrent fits per GPU, run by procs_per_gpu worker processes per GPU (None = fits_per_gpu: one fit
        per process, the rounds 1-3 layout) with ceil(fits_per_gpu / procs_per_gpu) fit threads each, every thread on its
        own stream.  max_workers caps the CONCURRENT FITS (= the number of units, usually)."""
        import time
        self.args, self.fit_fn, self.prepare, self.host_prepare = args, fit_fn, prepare, host_prepare
        self.log, self.poll_s, self.unit_timeout = log, poll_s, float(unit_timeout or 0.0)
        self.start_method = start_method or getattr(args, "worker_start", None) or START_METHOD
        n_vis = visible_gpus()
        self.n_g = max(1, min(n_gpus or n_vis, max(n_vis, 1)))
        fits = max(1, int(fits_per_gpu))
        procs = fits if procs_per_gpu is None else max(1, min(int(procs_per_gpu), fits))
        self.threads = -(-fits // procs)
        if max_workers is not None:                     # never more concurrent fits than units
            cap = max(1, int(max_workers))
            while self.threads > 1 and self.n_g * procs * (self.threads - 1) >= cap:
                self.threads -= 1
            self.n = max(1, min(self.n_g * procs, -(-cap // self.threads)))
        else:
            self.n = max(1, self.n_g * procs)
        self.fits = self.n * self.threads               # concurrent fits of the pool
        self.depth = self.threads + (1 if host_prepare is not None else 0)
        self.workers = {}                   # parent end of the pipe -> state dict
        self.ctx = None
        self.t0 = time.time()
        self.timeline = {"pool_created": self.t0, "workers": [], "units": {}, "n_workers": self.n, "n_gpus": self.n_g,
                         "fit_threads": self.threads}
        self.failed_starts = 0
        self._slot = 0
        # --unit_timeout promises that a hung fit is killed and replaced: only a worker PROCESS can be killed, so a timeout
        # keeps even a single worker out of this process (round 4's single-GPU default ran in-process and ignored the flag)
        # isolate=True asks for the same without a timeout: a HIP abort in one fit then costs a worker, not the whole run.
        self.isolate = (self.unit_timeout > 0) if isolate is None else bool(isolate)

    # ------------------------------------------------------------------ processes
    def start(self):
        """Spawn the workers now (no-op for a single in-process worker).  Returns self."""
        if self.workers:
            return self
        if self.n <= 1 and not self.isolate:
            return self                     # one process = this one: run() drives the fit threads itself
        # plain multiprocessing (only NumPy / pickled records cross the pipes): the parent does not import torch to start workers.
        # "forkserver" (default): the workers are forked from a server that has imported torch and this package once and has
        # never touched a GPU (warm_start - locator.main calls it first thing, so the imports overlap the parent's start-up);
        # "spawn": a fresh interpreter per worker (rounds 1-5).
        import multiprocessing as mp
        self.ctx = warm_start(self.start_method)
        if self.ctx is None:
            self.ctx, self.start_method = mp.get_context("spawn"), "spawn"
        for w in range(self.n):
            self._start_worker(w % self.n_g)
        return self

    def _start_worker(self, gpu):
        import time
        a, b = self.ctx.Pipe(duplex=True)
        t = time.time()
        p = self.ctx.Process(target=_worker, args=(gpu, self.fit_fn, self.args, self.prepare, self.host_prepare, b, t,
                                                   self.threads, _visible_env()), daemon=True)
        p.start()
        b.close()                           # the parent keeps only its own end: EOF then means "the worker is gone"
        self.workers[a] = {"p": p, "gpu": gpu, "inflight": [], "active": {}, "ready": False,
                           "shared_sent": False, "t_spawn": t, "slot": self._slot, "t_wait": t}
        self._slot += 1

    # ------------------------------------------------------------------ the run
    def run(self, units, shared=None):
        """Run every unit once; returns the result records in unit order."""
        import time
        tl = self.timeline
        tl["run_started"] = time.time()
        out = [None] * len(units)
        if not self.workers and not self.isolate and (self.n <= 1 or len(units) <= 1):
            # one worker process = this process: `threads` fit threads, each on its own stream, fed by ONE loader thread
            # that prepares the host side of the next units (at most threads + 1 prepared-or-fitting at a time)
            import threading
            from concurrent.futures import ThreadPoolExecutor
            n_fit = max(1, min(self.threads, len(units)))
            room = threading.Semaphore(n_fit + 1)
            global _FIT_THREADS
            _FIT_THREADS = n_fit

            def host(u):
                room.acquire()
                t1 = time.time()
                try:
                    return (self.host_prepare(u, self.args) if self.host_prepare is not None else u), None, time.time() - t1
                except Exception as e:                       # noqa: BLE001
                    return u, {"name": u.get("name", "?"), "error": f"{type(e).__name__}: {e}",
                               "traceback": traceback.format_exc()}, time.time() - t1

            def fit_one(i, fut):
                t1 = time.time()
                u, err, t_host = fut.result()
                t2 = time.time()
                try:
                    r = err if err is not None else _run_on_own_stream(self.fit_fn, u, shared, self.args, "cuda:0", self.prepare)
                finally:
                    room.release()
                r["unit_index"], r["gpu"] = i, 0
                r["host_prepare_seconds"], r["worker_seconds"] = t_host, time.time() - t2
                out[i] = r
                tl["units"][i] = {"gpu": 0, "dispatched": t1, "started": t2, "done": time.time()}
                if "error" in r:
                    self.log(f"replicate {r['name']} FAILED: {r['error']}")

            with ThreadPoolExecutor(1) as ex_host, ThreadPoolExecutor(n_fit) as ex_fit:
                hosts = [ex_host.submit(host, u) for u in units]
                fits = [ex_fit.submit(fit_one, i, f) for i, f in enumerate(hosts)]
                for f in fits:
                    f.result()
            _FIT_THREADS = 1
            tl["run_finished"] = time.time()
            return out
        from collections import deque
        from multiprocessing.connection import wait
        self.start()
        small, descs, handles = _share(shared)
        workers = self.workers
        state = {"got": 0, "next": 0}
        retry, crashed = deque(), set()          # crashed: units that were fitting (or first in line) when a worker was lost

        def record(r):
            i = r["unit_index"]
            if out[i] is None:
                out[i] = r
                state["got"] += 1
                tl["units"].setdefault(i, {})["done"] = time.time()
                if "error" in r:
                    self.log(f"replicate {r['name']} FAILED on GPU {r['gpu']}: {r['error']}")
                else:
                    self.log(f"replicate {r['name']} done on GPU {r['gpu']} in {r.get('seconds', 0):.1f} s")

        def next_index():
            while retry:
                i = retry.popleft()
                if out[i] is None:
                    return i
            if state["next"] < len(units):
                state["next"] += 1
                return state["next"] - 1
            return None

        def feed(conn):
            w = workers.get(conn)
            if w is None or not w["ready"]:
                return
            try:
                if not w["shared_sent"]:
                    conn.send(("shared", small, descs))
                    w["shared_sent"] = True
                while len(w["inflight"]) < self.depth:
                    if w.get("exclusive") is not None:
                        return                       # a crash suspect has this worker to itself
                    i = next_index()
                    if i is None:
                        return
                    if i in crashed:                 # lost a worker once: its second try runs ALONE, so that if it is the
                        if w["inflight"]:            # one that kills workers it takes nothing else with it
                            retry.appendleft(i)
                            return
                        w["exclusive"] = i
                    if not w["inflight"]:
                        w["t_wait"] = time.time()    # an idle worker starts waiting for this unit's host work now
                    w["inflight"].append(i)
                    tl["units"].setdefault(i, {}).update(gpu=w["gpu"], dispatched=time.time())
                    conn.send(("unit", i, units[i]))
            except (OSError, ValueError):        # the worker vanished between two messages: its units go back
                bury(conn, "pipe closed")

        def bury(conn, why, culprit=None):
            """The worker behind `conn` is gone.  culprit = the unit known to have caused it (the one that timed out): an error
            record at once, and everything else it held goes back to the queue as it is.  Without a culprit (a crash does not say
            which fit thread aborted) the units that were FITTING - or, if none was, the first in line, whose host work was
            running - become suspects: each gets one more try, alone on a worker, and fails when that worker is lost too; units
            that were only prefetched go back uncounted.  So a unit that kills workers costs two workers and nothing else, and
            an innocent sibling of it always completes."""
            w = workers.pop(conn, None)
            if w is None:
                return
            try:
                conn.close()
            except OSError:
                pass
            w["p"].join(5)
            suspects = set(w["active"]) if w["active"] else set(w["inflight"][:1])
            for i in w["inflight"]:
                if out[i] is not None:
                    continue
                if i == culprit or (culprit is None and i in suspects and i in crashed):
                    what = "while fitting this unit" + (" (second worker lost)" if i in crashed else "")
                    record({"name": units[i].get("name", "?"), "unit_index": i, "gpu": w["gpu"],
                            "error": f"worker process died ({why}, exit code {w['p'].exitcode}) {what}"})
                else:
                    if culprit is None and i in suspects:
                        crashed.add(i)
                    retry.append(i)                      # another worker (or the replacement) takes it
            if not w["inflight"] and why not in ("retired",):
                self.failed_starts += 1
            if state["got"] < len(units) and (retry or state["next"] < len(units)) and self.failed_starts < 4 * self.n:
                self._start_worker(w["gpu"])
            feed_idle()                                  # requeued units go to workers that are ready NOW, not only to
                                                         # whichever worker speaks next (or to a replacement's start-up)

        def feed_idle():
            for c in list(workers):
                w2 = workers.get(c)
                if w2 is not None and w2["ready"] and len(w2["inflight"]) < self.depth:
                    feed(c)

        try:
            while state["got"] < len(units):
                if not workers:                 # every start-up failed repeatedly: report what is left and stop
                    for i, r in enumerate(out):
                        if r is None:
                            record({"name": units[i].get("name", "?"), "unit_index": i, "gpu": -1,
                                    "error": "no worker process could be started"})
                    break
                for conn in wait(list(workers), timeout=self.poll_s):
                    if conn not in workers:
                        continue
                    try:
                        kind, payload = conn.recv()
                    except (EOFError, OSError):
                        bury(conn, "pipe closed")
                        continue
                    w = workers[conn]
                    if kind == "ready":
                        w["ready"] = True
                        info = dict(payload or {})
                        tl["workers"].append({"slot": w["slot"], "gpu": w["gpu"], "spawned": w["t_spawn"],
                                              "ready": info.pop("ready_at", time.time()), **info})
                    elif kind == "start":
                        w["active"][payload] = time.time()
                        tl["units"].setdefault(payload, {})["started"] = w["active"][payload]
                    elif kind == "done":
                        if w.get("exclusive") == payload["unit_index"]:
                            w["exclusive"] = None
                        if payload["unit_index"] in w["inflight"]:
                            w["inflight"].remove(payload["unit_index"])
                        w["active"].pop(payload["unit_index"], None)
                        w["t_wait"] = time.time()
                        record(payload)
                    elif kind == "dead":
                        self.log(f"replicate worker on GPU {w['gpu']}: {payload}")
                        bury(conn, "start-up failure" if not w["inflight"] else str(payload))
                        continue
                    feed(conn)
                now = time.time()
                if retry:
                    feed_idle()
                    if retry and not any(w2["ready"] or w2["p"].is_alive() for w2 in workers.values()):
                        while retry:            # nobody left who could ever take them: error records, not a spin
                            i = retry.popleft()
                            if out[i] is None:
                                record({"name": units[i].get("name", "?"), "unit_index": i, "gpu": -1,
                                        "error": "no worker process left to retry this unit"})
                for conn in list(workers):      # belt and braces: a process can be gone before its pipe says so
                    w = workers[conn]
                    if not w["p"].is_alive() and not conn.poll():
                        bury(conn, "not alive")
                    elif self.unit_timeout and w["active"] and now - min(w["active"].values()) > self.unit_timeout:
                        slow = min(w["active"], key=w["active"].get)
                        self.log(f"replicate {units[slow].get('name', '?')} on GPU {w['gpu']} exceeded "
                                 f"--unit_timeout {self.unit_timeout:g} s: killing its worker (pid {w['p'].pid})")
                        w["p"].kill()               # this exact process
                        bury(conn, f"timed out after {self.unit_timeout:g} s", culprit=slow)
                    elif (self.unit_timeout and not w["active"] and w["inflight"]
                          and now - w["t_wait"] > self.unit_timeout):
                        # nothing fitting, yet the next unit never reported "start": its host work (loader thread) hangs
                        self.log(f"replicate {units[w['inflight'][0]].get('name', '?')} on GPU {w['gpu']}: host work exceeded "
                                 f"--unit_timeout {self.unit_timeout:g} s: killing its worker (pid {w['p'].pid})")
                        w["p"].kill()
                        bury(conn, f"host work timed out after {self.unit_timeout:g} s", culprit=w["inflight"][0])
        finally:
            for shm in handles:
                shm.close()
                shm.unlink()
        tl["run_finished"] = time.time()
        return out

    def close(self):
        for conn, w in list(self.workers.items()):
            try:
                conn.send(None)
            except (OSError, ValueError):
                pass
        for conn, w in list(self.workers.items()):
            w["p"].join(30)
            if w["p"].is_alive():
                w["p"].terminate()
            try:
                conn.close()
            except OSError:
                pass
        self.workers = {}

    # ------------------------------------------------------------------ what happened, and what N GPUs would do
    def summary(self, records=None, program_started=None):
        """Phase timeline of the run as a dict (and `lines` for printing): parent time before the dispatch loop,
        worker start-up (and how much of it the parent's prologue hid), per-unit host / fit seconds, the part of the
        wall time that the parallel work does not explain (serial fraction), and the Amdahl projection for 1..8 GPUs at
        the same workers per GPU: T(g) = serial + unit_work / (g * workers_per_gpu), unit_work = sum of worker seconds."""
        tl = self.timeline
        t_prog = program_started if program_started is not None else tl["pool_created"]
        t_run0, t_run1 = tl.get("run_started", tl["pool_created"]), tl.get("run_finished", tl["pool_created"])
        wall = t_run1 - t_prog
        recs = [r for r in (records or []) if r is not None]
        work = sum(r.get("worker_seconds", 0.0) for r in recs)
        host = sum(r.get("host_prepare_seconds", 0.0) for r in recs)
        ready = [w["ready"] for w in tl["workers"]]
        startup = [w.get("startup_seconds", 0.0) + w.get("spawn_seconds", 0.0) for w in tl["workers"]]
        first_ready = (min(ready) - t_prog) if ready else 0.0
        n_w = max(1, min(self.fits, len(recs)) if recs else self.fits)
        per_gpu = max(1, self.fits // self.n_g)
        serial = max(0.0, wall - work / n_w)
        s = {"wall_seconds": wall, "parent_prologue_seconds": t_run0 - t_prog, "dispatch_loop_seconds": t_run1 - t_run0,
             "workers": self.n, "fit_threads": self.threads, "gpus": self.n_g, "worker_startup_seconds_mean": (sum(startup) / len(startup)) if startup else 0.0,
             "first_worker_ready_after_seconds": first_ready, "units": len(recs), "unit_work_seconds": work,
             "host_prepare_seconds_total": host, "serial_seconds": serial, "serial_fraction": serial / wall if wall > 0 else 0.0,
             "amdahl_projection_seconds": {g: serial + work / (g * per_gpu) for g in (1, 2, 4, 8)}}
        s["amdahl_speedup_vs_1gpu"] = {g: s["amdahl_projection_seconds"][1] / v for g, v in s["amdahl_projection_seconds"].items()}
        s["lines"] = [
            f"replicate timeline: wall {wall:.1f} s = parent prologue {s['parent_prologue_seconds']:.1f} s + dispatch loop "
            f"{s['dispatch_loop_seconds']:.1f} s; {self.n} worker process(es) x {self.threads} fit thread(s) on {self.n_g} GPU(s), start-up "
            f"{s['worker_startup_seconds_mean']:.1f} s each (first one ready {first_ready:.1f} s after program start)",
            f"  {len(recs)} units: {work:.1f} s of worker time (host slices + filters {host:.1f} s on loader threads), "
            f"serial part {serial:.1f} s = {100 * s['serial_fraction']:.0f} % of the wall",
            "  Amdahl projection at this serial part, same workers per GPU: " +
            ", ".join(f"{g} GPU {v:.1f} s ({s['amdahl_speedup_vs_1gpu'][g]:.2f}x)" for g, v in s["amdahl_projection_seconds"].items())]
        aff = sorted({f"GPU {w['gpu']} -> {w['cpu_affinity']}" for w in tl["workers"] if w.get("cpu_affinity")})
        if tl["workers"]:
            s["lines"].append(f"  workers: {self.start_method} start, one visible device each (HIP_VISIBLE_DEVICES)"
                              + (", CPU affinity from the device's PCI node: " + "; ".join(aff[:8]) if aff else ""))
        return s


def run_units(units, args, fit_fn, n_gpus=None, shared=None, prepare=None, log=print, fits_per_gpu=1, poll_s=1.0,
              host_prepare=None, unit_timeout=0.0, procs_per_gpu=None, isolate=None):
    """Run every unit once; returns the result records in unit order (ReplicatePool started, run and closed here).

    units         list of dicts (small per-unit data; window units carry their window, not their genotypes)
    shared        dict of data common to all units; NumPy arrays of >= 1 MB are placed in shared memory once and
                  attached by every worker (the 0.5 GB bootstrap matrix is not pickled 16 times)"""
    pool = ReplicatePool(args, fit_fn, n_gpus=n_gpus, fits_per_gpu=fits_per_gpu, prepare=prepare,
                         host_prepare=host_prepare, log=log, poll_s=poll_s, unit_timeout=unit_timeout,
                         max_workers=len(units), procs_per_gpu=procs_per_gpu, isolate=isolate)
    try:
        return pool.run(units, shared)
    finally:
        pool.close()


def run_units_distributed(units, args, fit_fn, shared=None, prepare=None, device=None):
    """One-process-per-GPU launches (torch.distributed.run): rank r fits units r, r+world, ...;
    the small result records are exchanged with all_gather_object (control plane only)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = {}
    for i in shard_static(len(units), rank, world):
        u = units[i]
        if prepare is not None:
            u = prepare(u)
        r = _run_one(fit_fn, u, shared, args, device or "cuda:0")
        r["unit_index"], r["gpu"] = i, rank
        mine[i] = r
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    out = [None] * len(units)
    for part in gathered:
        for i, r in part.items():
            out[i] = r
    return out
