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


def _worker(gpu, fit_fn, small, descs, args, prepare, conn):
    """Worker process on its own duplex pipe (no queue or lock is shared between workers, so one that is killed
    cannot wedge the others): receives (index, unit), sends back the result record, exits on None."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_device(gpu)
            device = f"cuda:{gpu}"
        else:               # scheduler tests on CPU; a real fit_fn raises on this device (no CPU fallback)
            device = "cpu"
        shared, keep = _attach(small, descs)
    except Exception as e:                                   # noqa: BLE001 - the parent turns it into error records
        conn.send(("dead", f"worker start-up failed: {type(e).__name__}: {e}"))
        return
    conn.send(("ready", None))
    while True:
        item = conn.recv()
        if item is None:
            break
        idx, unit = item
        try:
            if prepare is not None:
                unit = prepare(unit)
            r = _run_one(fit_fn, unit, shared, args, device)
        except Exception as e:                               # noqa: BLE001 - e.g. prepare() raising
            r = {"name": unit.get("name", "?") if isinstance(unit, dict) else "?",
                 "error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()}
        r["unit_index"], r["gpu"] = idx, gpu
        conn.send(("done", r))
    for shm in keep:
        shm.close()


def visible_gpus():
    import torch
    return torch.cuda.device_count()


def run_units(units, args, fit_fn, n_gpus=None, shared=None, prepare=None, log=print, fits_per_gpu=1, poll_s=1.0):
    """Run every unit once; returns the result records in unit order.

    units         list of dicts (small per-unit data; window units carry their own genotype slices)
    shared        dict of data common to all units; NumPy arrays of >= 1 MB are placed in shared memory once and
                  attached by every worker (the 0.5 GB bootstrap matrix is not pickled 16 times)
    prepare       optional per-unit hook run in the worker before fit_fn (e.g. column resampling)
    fits_per_gpu  worker processes per GPU.  A single fit alternates between an HBM-bound phase (layer 1) and a
                  latency-bound phase (hidden stack, 16 CUs); two fits on one GPU interleave them: measured 209k
                  vs 157k samples/s aggregate on the 1000 x 100k workload (1.33x), no further gain from a third.

    Dynamic dispatch: the parent hands the next unit to whichever worker is idle (early stopping makes unit durations
    vary several-fold).  A worker that dies (GPU fault, OOM kill, abort) does not hang the run: its pipe reports
    end-of-file, the parent records an error for the unit it was holding and starts a FRESH process in its place
    (never re-executes the dead one) while units remain."""
    n_vis = visible_gpus()
    n_g = max(1, min(n_gpus or n_vis, max(n_vis, 1)))
    n = max(1, min(n_g * max(1, int(fits_per_gpu)), len(units)))
    out = [None] * len(units)
    if n <= 1:
        for i, u in enumerate(units):
            try:
                if prepare is not None:
                    u = prepare(u)
                r = _run_one(fit_fn, u, shared, args, "cuda:0")
            except Exception as e:                           # noqa: BLE001
                r = {"name": u.get("name", "?"), "error": f"{type(e).__name__}: {e}",
                     "traceback": traceback.format_exc()}
            r["unit_index"], r["gpu"] = i, 0
            out[i] = r
            if "error" in r:
                log(f"replicate {r['name']} FAILED: {r['error']}")
        return out
    from multiprocessing.connection import wait

    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    small, descs, handles = _share(shared)
    workers = {}                            # parent end of the pipe -> [process, gpu, unit index in flight or None, ready]
    state = {"got": 0, "next": 0, "failed_starts": 0}

    def start_worker(gpu):
        a, b = ctx.Pipe(duplex=True)
        p = ctx.Process(target=_worker, args=(gpu, fit_fn, small, descs, args, prepare, b), daemon=True)
        p.start()
        b.close()                           # the parent keeps only its own end: EOF then means "the worker is gone"
        workers[a] = [p, gpu, None, False]

    def record(r):
        if out[r["unit_index"]] is None:
            out[r["unit_index"]] = r
            state["got"] += 1
            if "error" in r:
                log(f"replicate {r['name']} FAILED on GPU {r['gpu']}: {r['error']}")
            else:
                log(f"replicate {r['name']} done on GPU {r['gpu']} in {r.get('seconds', 0):.1f} s")

    def feed(conn):
        w = workers[conn]
        if w[3] and w[2] is None and state["next"] < len(units):
            i = state["next"]
            state["next"] += 1
            w[2] = i
            conn.send((i, units[i]))

    def bury(conn, why):
        p, gpu, idx, _ = workers.pop(conn)
        conn.close()
        p.join(5)
        if idx is not None and out[idx] is None:
            record({"name": units[idx].get("name", "?"), "unit_index": idx, "gpu": gpu,
                    "error": f"worker process died ({why}, exit code {p.exitcode}) while fitting this unit"})
        elif idx is None and why != "retired":
            state["failed_starts"] += 1
        if state["got"] < len(units) and state["next"] < len(units) and state["failed_starts"] < 4 * n:
            start_worker(gpu)

    try:
        for w in range(n):
            start_worker(w % n_g)
        while state["got"] < len(units):
            if not workers:                 # every start-up failed repeatedly: report what is left and stop
                for i, r in enumerate(out):
                    if r is None:
                        record({"name": units[i].get("name", "?"), "unit_index": i, "gpu": -1,
                                "error": "no worker process could be started"})
                break
            for conn in wait(list(workers), timeout=poll_s):
                try:
                    kind, payload = conn.recv()
                except (EOFError, OSError):
                    bury(conn, "pipe closed")
                    continue
                if kind == "ready":
                    workers[conn][3] = True
                elif kind == "done":
                    workers[conn][2] = None
                    record(payload)
                elif kind == "dead":
                    log(f"replicate worker on GPU {workers[conn][1]}: {payload}")
                    bury(conn, "start-up failure")
                    continue
                feed(conn)
            for conn in list(workers):      # belt and braces: a process can be gone before its pipe says so
                if not workers[conn][0].is_alive() and not conn.poll():
                    bury(conn, "not alive")
        for conn, w in list(workers.items()):
            try:
                conn.send(None)
            except (OSError, ValueError):
                pass
        for conn, w in list(workers.items()):
            w[0].join(30)
            if w[0].is_alive():
                w[0].terminate()
            conn.close()
    finally:
        for shm in handles:
            shm.close()
            shm.unlink()
    return out


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
