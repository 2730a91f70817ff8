"""Replicate workers on a real GPU: forked from a server that imported torch but never touched the GPU,
each bound to ONE device through HIP_VISIBLE_DEVICES before its first HIP call."""
import os

import numpy as np
import pytest

from locator_amd import replicates as R

pytestmark = pytest.mark.gpu


def _report_fit(unit, device="cpu"):
    """What a worker process sees: its device string, how many devices, the variable, whether torch was already imported
    when the process started working (fork server preload), and a kernel launch on the device it was given."""
    import sys
    import time
    import torch
    time.sleep(0.4)          # long enough for the second worker (a spawned one imports torch first) to take its share
    x = torch.arange(8, device=device, dtype=torch.float32)
    return {"name": unit["name"], "device": device, "count": torch.cuda.device_count(), "seconds": 0.0,
            "visible": os.environ.get("HIP_VISIBLE_DEVICES"), "cuda_visible": os.environ.get("CUDA_VISIBLE_DEVICES"),
            "sum": float((x * 2).sum().item()), "pid": os.getpid(), "lib_loaded": "locator_amd._lib" in sys.modules}


class _Args:
    out = "stem"


def _probe_visible(gpu, env, conn):
    R._bind_worker_to_gpu(gpu, env)
    import torch
    conn.send((os.environ.get("HIP_VISIBLE_DEVICES"), torch.cuda.device_count(), torch.cuda.is_available()))


@pytest.mark.parametrize("method", ["forkserver", "spawn"])
def test_workers_see_exactly_their_own_device_as_cuda0(method):
    units = [dict(name=f"u{i}", replicate=i) for i in range(10)]
    pool = R.ReplicatePool(_Args(), _report_fit, n_gpus=1, fits_per_gpu=2, procs_per_gpu=2, isolate=True, start_method=method,
                           log=lambda *a: None, poll_s=0.05)
    pool.start()
    res = pool.run(units)
    pool.close()
    assert all("error" not in r for r in res), res
    assert {r["device"] for r in res} == {"cuda:0"} and {r["count"] for r in res} == {1}
    assert {r["visible"] for r in res} == {"0"} and {r["cuda_visible"] for r in res} == {None}
    assert all(r["sum"] == 56.0 and r["lib_loaded"] for r in res)
    assert len({r["pid"] for r in res}) == 2 and os.getpid() not in {r["pid"] for r in res}
    s = pool.summary(res)
    assert f"{method} start" in s["lines"][-1]
    if method == "forkserver":          # a forked worker pays a device context, not an interpreter + `import torch`
        assert s["worker_startup_seconds_mean"] < 0.9, s["worker_startup_seconds_mean"]


def test_the_binding_is_applied_before_the_first_hip_call_of_a_forked_worker():
    """The fork server has imported torch long before a worker knows its GPU: HIP_VISIBLE_DEVICES set in the forked child must
    still decide what it sees.  Index 0 of the parent's list -> one device; an index past the list -> its own number, which
    does not exist on a one-GPU box -> no device at all (so the variable was read AFTER the fork); a parent restricted by
    --gpu_number (the value travels explicitly) maps its g-th entry."""
    ctx = R.warm_start("forkserver")
    for gpu, env, want in ((0, {}, ("0", 1, True)), (5, {}, ("5", 0, False)),
                           (1, {"HIP_VISIBLE_DEVICES": "3,0", "CUDA_VISIBLE_DEVICES": "3,0"}, ("0", 1, True))):
        a, b = ctx.Pipe()
        p = ctx.Process(target=_probe_visible, args=(gpu, env, b))
        p.start()
        got = a.recv()
        p.join(30)
        assert got == want, (gpu, env, got)
