#!/usr/bin/env python3
"""Headline benchmark: training samples/s of locator model fits on a synthetic 1,000-individual x 100,000-SNP
genotype matrix (BASELINE.json configs[2]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE EPOCH of model.fit on the 810-row training split: 26 minibatch steps of batch 32 (25 full + one of
10, kept as Keras keeps it), the validation sweep over the 90 held-out rows, and the host-side callbacks
(ModelCheckpoint snapshot when val_loss improves, LR plateau).  Inputs (genotypes, targets, weights) are resident
in HBM before the timed region starts.

N > 1: one process per GPU, each fitting its OWN bootstrap replicate of the same matrix (locator.py:635-681:
replicates are independent fits) - no data-path collective; torch.distributed is used only for the barrier and the
max-over-ranks of the elapsed time.  scaling = "weak".  Launched under torch.distributed.run the ranks come from
the environment; launched plainly with --gpus N > 1, this script starts the N rank processes itself (before it
touches any GPU) and relays rank 0's line.

--replicates-per-gpu R: R independent fits per process, each on its own HIP stream with its own captured epoch
graph, started together (the latency-bound hidden stack of one fit overlaps the HBM-bound layer-1 kernels of the
other); value counts the samples of all of them.  The headline configuration is R = 1.

The JSON line also carries
  roofline      the dominant kernel (l1_bwd_adam: fused layer-1 backward + Adam): algorithmic bytes per launch / its
                mean duration from HIP events recorded on the launch stream immediately before and after that
                kernel, vs 8 TB/s HBM.  `traffic` = PMC-measured bytes per launch from the committed rocprofv3
                passes (profiles/r06_pmc_traffic.json), reported only while the kernel sources still hash to what
                was profiled (`traffic_source` says which).
  cpu_baseline  the torch-CPU fp32 restatement of the same epochs (oracle/torch_cpu.py: "restated reference on CPU
                (torch), not TensorFlow") on this box's physical cores, validation sweep included,
                bounded to ~12 s; `numpy_port` = the single-threaded-Adam NumPy oracle step for comparison.
  l1_gemm       (N=1) a dozen-number summary of the large-M first-layer genotype GEMM (model.predict) in the default
                predict mode as a fraction of the dense bf16-MFMA peak (l1_gemm_summary).  The full mode x shape sweep
                lives in tools/l1_gemm_sweep.py / --l1-gemm-full FILE and never enters the line.

The line is the last thing printed, strict JSON, at most MAX_LINE_BYTES (format_line; tests/test_host.py).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BF16_PEAK_TFLOPS = 2500.0    # same guide: dense bf16 MFMA peak (no sparsity)
TRAFFIC_PROFILE = os.path.join("profiles", "r06_pmc_traffic.json")
TRAFFIC_SOURCES = ("locator_amd/csrc/l1_kernels.hip", "locator_amd/csrc/l1_chain.hip", "locator_amd/csrc/common.h")


def l1_bwd_bytes(K, H, n_b):
    """Algorithmic HBM bytes of one l1_bwd_adam launch: W1,m,v read + written once
    (24 B/weight), the batch's uint8 genotype rows once, BN scale/shift/mean/rstd read (16 B/SNP),
    gamma,beta and their Adam moments read + written (48 B/SNP), dZ1 and b1 state."""
    return 24 * K * H + n_b * K + 64 * K + 4 * 32 * H + 24 * H


def l1_chain_bytes(K, H, n_b, n_b_next, groups, tail_layers=0):
    """The same for one l1_bwd_adam_chain launch (locator_amd/csrc/l1_chain.hip): the layer-1 backward's bytes, plus --
    when the launch also computes the next minibatch's layer-1 forward -- that minibatch's genotype rows, its batch
    statistics (8 B/SNP), the next step's [scale|shift|mean|rstd] written (16 B/SNP) and the partial sums
    [groups][32][H] fp32.  W1, m, v still cross HBM exactly once each way."""
    b = l1_bwd_bytes(K, H, n_b)
    if n_b_next:
        b += n_b_next * K + 8 * K + 16 * K + groups * 32 * H * 4
    if tail_layers:
        # the step's hidden-layer / head Adam tail riding as trailing workgroups of the launch: W, m, v of the hidden
        # kernels read + written (24 B / weight) and the transposed copy written (4 B / weight)
        b += 28 * ((tail_layers - 1) * H * H + (tail_layers + 1) * H + 8)
    return b


def step_bytes(K, H, n_b, L=10):
    """SURVEY.md §8(d) bytes_step for a whole minibatch step."""
    return 28 * K * H + 2 * n_b * K + 64 * K + 28 * ((L - 1) * H * H + (L + 1) * H + 8)


def kernel_sources_sha():
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(K, H, n, kernel, batch=32):
    """PMC bytes per launch of `kernel` (name prefix) from the committed profile, or (None, why) when it does not describe the
    kernel that is running now."""
    path = os.path.join(ROOT, TRAFFIC_PROFILE)
    try:
        pm = json.load(open(path))
    except Exception:
        return None, f"{TRAFFIC_PROFILE} missing"
    if pm.get("workload") != {"K": K, "H": H, "n": n} or batch != 32:       # (the passes run the default --batch_size)
        return None, f"{TRAFFIC_PROFILE} is for another workload"
    if pm.get("kernel_sources_sha256_16") != kernel_sources_sha():
        return None, f"{TRAFFIC_PROFILE} was taken on other kernel sources (re-run tools/pmc_traffic.sh)"
    key = [k for k in pm["kernels"] if k.startswith(kernel)]
    if not key:
        return None, "kernel not in profile"
    return int(pm["kernels"][key[0]]["traffic_bytes"]), f"{TRAFFIC_PROFILE} (rocprofv3 --pmc passes, sources {pm['kernel_sources_sha256_16']})"


def l1_gemm_summary(net, n_matrix, iters=10):
    """The large-M first-layer genotype GEMM (model.predict / --jacknife, locator.py:414, :441, :713-741) in the DEFAULT
    predict mode (--predict_mode auto: int8 digit planes chosen by the dynamic-range guard) as a fraction of the dense
    bf16-MFMA peak: flops = 2*M*K*H counted once however many planes carry a weight.  Three shapes: every row of the
    matrix (M = n), 4096 and 16384 DISTINCT rows streaming from HBM, each from bytes and from the 2-bit packed copy the
    predict keeps for >= 3072 rows; us_prep = the once-per-predict weight image build, `incl_prep` has it in the
    denominator.  A dozen numbers; the full mode x shape sweep is tools/l1_gemm_sweep.py (--l1-gemm-full)."""
    import ctypes as C

    import torch
    from locator_amd import _lib
    from tools.l1_gemm_sweep import REPLAYS, _time_graphed, _time_graphed_ab, distinct_rows
    lib, d, lay = net.lib, net.d, net.lay
    P, dev = net.params.data_ptr(), net.params.device
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = net.quant_guard()
    digits = int(g[2])
    out = {"mode": "int8x%d" % digits if digits > 0 else "bf16x3", "peak_tflops": BF16_PEAK_TFLOPS, "timing": "median of %d graph replays" % REPLAYS,
           "guard_median": round(g[0], 1), "guard_max": round(g[1], 1)}
    if digits <= 0 or not lib.loc_l1_gemm_i8_supported(d.Hp, digits):
        return out
    bn4 = torch.zeros(4 * d.Kp, device=dev)
    _lib.check(lib.loc_bn_infer_scale_shift(d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta, P + 4 * lay.mov_mean,
                                            P + 4 * lay.mov_var, bn4.data_ptr(), st()))
    partial = torch.empty(256 * 128 * d.Hp, device=dev)
    image = torch.empty(lib.loc_l1_image_i8_bytes(C.byref(d), digits), dtype=torch.uint8, device=dev)
    prep = lambda: _lib.check(lib.loc_l1_image_i8_build(C.byref(d), bn4.data_ptr(), P + 4 * lay.w1, digits,
                                                        image.data_ptr(), st()))
    prep()
    us_prep = _time_graphed(prep, 10)
    out["us_prep"] = round(us_prep, 1)
    frac = lambda fl, us: round(fl / us * 1e-6 / BF16_PEAK_TFLOPS, 4)

    def shape(X, n_rows, n_src, packed):
        rows = (torch.arange(n_rows, dtype=torch.int32, device=dev) % n_src).contiguous()
        a1 = torch.empty(((n_rows + 127) // 128 * 128, d.Hp), device=dev)
        fl = 2.0 * n_rows * d.K * d.H
        run = lambda: _lib.check(lib.loc_l1_forward_gemm_i8(X.data_ptr(), X.stride(0), rows.data_ptr(), n_rows, C.byref(d),
                                                            image.data_ptr(), digits, 2, P + 4 * lay.b1, partial.data_ptr(),
                                                            partial.numel(), a1.data_ptr(), 0, None, st()))
        if not packed:
            us = _time_graphed(run, iters)
            sp = _time_graphed.last
            return {"us": round(us, 1), "min_max": [round(sp["min"], 1), round(sp["max"], 1)], "frac": frac(fl, us),
                    "incl_prep": frac(fl, us + us_prep)}
        else:
            X2 = torch.zeros((X.shape[0], d.Kp // 4), dtype=torch.uint8, device=dev)
            _lib.check(lib.loc_pack_genotypes_2bit(X.data_ptr(), X.stride(0), X.shape[0], d.Kp, X2.data_ptr(), X2.stride(0), st()))
            runp = lambda: _lib.check(lib.loc_l1_forward_gemm_i8_packed(X2.data_ptr(), X2.stride(0), rows.data_ptr(), n_rows,
                                                                        C.byref(d), image.data_ptr(), digits, P + 4 * lay.b1,
                                                                        partial.data_ptr(), partial.numel(), a1.data_ptr(), 0,
                                                                        None, st()))
            # bytes and packed interleaved replay by replay (same clock state): medians of REPLAYS each
            sb, sp = _time_graphed_ab([run, runp], iters)
            us, usp = sb["median"], sp["median"]
            r = {"us": round(us, 1), "min_max": [round(sb["min"], 1), round(sb["max"], 1)], "frac": frac(fl, us),
                 "incl_prep": frac(fl, us + us_prep)}
            r["packed"] = {"us": round(usp, 1), "min_max": [round(sp["min"], 1), round(sp["max"], 1)], "frac": frac(fl, usp),
                           "incl_prep": frac(fl, usp + us_prep)}
        return r

    out["rows_%d" % n_matrix] = shape(net.X, n_matrix, n_matrix, False)
    xd = distinct_rows(dev, d.Kp, 16384)
    out["distinct_4096"] = shape(xd[:4096], 4096, 4096, True)
    out["distinct_16384"] = shape(xd, 16384, 16384, True)
    return out


MAX_LINE_BYTES = 4096        # the driver keeps a bounded tail of stdout: the headline line must fit it whole


def format_line(out):
    """The ONE JSON line of the run: strict JSON (no NaN / Infinity), at most MAX_LINE_BYTES.  Optional detail is dropped -
    never the contract keys - should the line outgrow the bound; a line that still does not fit is an error, not output."""
    def dumps(o):
        return json.dumps(o, allow_nan=False, separators=(", ", ": "))

    def scrub(o):
        if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
            return None
        if isinstance(o, dict):
            return {k: scrub(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [scrub(v) for v in o]
        return o

    out = scrub(out)
    line = dumps(out)
    for path in (("cpu_baseline", "numpy_port"), ("roofline", "traffic_source"), ("config", "step"), ("l1_gemm",)):
        if len(line) <= MAX_LINE_BYTES:
            break
        o = out
        for k in path[:-1]:
            o = o.get(k) or {}
        o.pop(path[-1], None)
        line = dumps(out)
    if len(line) > MAX_LINE_BYTES:
        raise RuntimeError(f"bench line is {len(line)} bytes (> {MAX_LINE_BYTES})")
    missing = [k for k in REQUIRED_KEYS if k not in out]
    if missing:
        raise RuntimeError(f"bench line lacks {missing}")
    return line


REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")

def cpu_baseline(x, y_norm, train, test, K, H, seconds):
    """torch-CPU fp32 restatement of the same epochs on the host's physical cores (oracle/torch_cpu.py), plus the NumPy
    oracle's step rate as a second, smaller figure.  Bounded sample."""
    from oracle import locator_oracle as O
    from oracle import torch_cpu
    out = torch_cpu.time_epochs(x, y_norm, train, test, K, H, seconds=seconds)
    rng = np.random.default_rng(0)
    p = O.init_params(K, H, 10, rng, dtype=np.float32)
    m, v = O.zeros_like_trainable(p), O.zeros_like_trainable(p)
    xt, yt = x[train], y_norm[train].astype(np.float32)
    steps, t0 = 0, time.perf_counter()
    while steps < 3 or time.perf_counter() - t0 < 4.0:
        rows = rng.choice(len(train), 32, replace=False)
        O.train_step(p, m, v, steps + 1, np.float32(1e-3), xt[rows], yt[rows], rng.random((32, H)) >= 0.25, 0.25)
        steps += 1
    out["numpy_port"] = {"value": round(32 * steps / (time.perf_counter() - t0), 1), "unit": "samples/s",
                         "sample": f"{steps} minibatch steps, NumPy fp32 oracle (BLAS contractions, single-threaded "
                                   "elementwise Adam), no validation sweep"}
    out["value"] = round(out["value"], 1)
    return out


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start the N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment) and relay rank 0's output.  Runs before this process has touched a GPU and never touches one."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def selftest_launch(args):
    """The N-rank control plane without a GPU: rendezvous on 127.0.0.1, barrier, MAX all-reduce of a per-rank time,
    one JSON line from rank 0 - exactly what the timed path does around its epochs."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": world, "max_over_ranks": float(t.item())}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed epochs (5 ms each on one MI355X)")
    ap.add_argument("--warmup", type=int, default=5, help="untimed epochs")
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--snps", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32, help="--batch_size of the fit (headline: 32, the reference default)")
    ap.add_argument("--replicates-per-gpu", type=int, default=1, help="independent fits per process on separate streams")
    ap.add_argument("--nt-mask", type=int, default=0, help="loc_tuning.l1b_nt_mask (cache-policy measurement switch)")
    ap.add_argument("--stack-helpers", type=int, default=0, help="loc_tuning.stack_helpers (L2 warm-up workgroups of the hidden stack; measurement switch)")
    ap.add_argument("--stack-xcd-stride", type=int, default=0, help="loc_tuning.stack_xcd_stride (1, 2, 4, 8: the hidden stack's workers share 8 / n XCDs; measurement switch)")
    ap.add_argument("--stack-train-rows", type=int, default=0, help="loc_tuning.stack_train_rows (1, 2, 4 batch rows per workgroup of the training stack; 0 = default)")
    ap.add_argument("--epoch-times", action="store_true", help="measurement switch: host-observed completion interval of every timed epoch on stderr")
    ap.add_argument("--separate-tail", action="store_true",
                    help="measurement switch (loc_tuning.chain_tail = -1): the hidden-layer Adam tail of a chained step as "
                         "its own launch instead of trailing workgroups of the chained layer-1 launch")
    ap.add_argument("--l1-bwd-grid", type=int, default=0,
                    help="workgroups of the layer-1 backward (measurement switch; default 2 x CUs; e.g. 2 x CUs - 32 leaves "
                         "some CUs with one backward workgroup so a second fit's stack waves can co-reside)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-l1-gemm", action="store_true")
    ap.add_argument("--l1-gemm-full", default=None, metavar="FILE",
                    help="also run the full predict-mode x shape GEMM sweep (tools/l1_gemm_sweep.py) and write it to FILE")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-graph", action="store_true", help="enqueue every kernel from the host each epoch")
    ap.add_argument("--side-stats", action="store_true",
                    help="measurement switch: the next epoch's BN batch statistics on a side stream (FitLoop side_stats)")
    ap.add_argument("--no-xchain", action="store_true",
                    help="measurement switch: do not chain the last step of an epoch into the next epoch's first layer-1 forward")
    ap.add_argument("--sync-epochs", action="store_true",
                    help="measurement switch: wait for every epoch before enqueueing the next (FitLoop depth 0) instead of "
                         "running two epochs ahead of the device")
    ap.add_argument("--lib", default=None, help="measurement switch: another build of liblocator_hip.so (make ablate_chain ...)")
    ap.add_argument("--no-chain", action="store_true",
                    help="measurement switch: one layer-1 forward launch per step instead of chaining it into the "
                         "previous step's layer-1 backward (locator_amd/csrc/l1_chain.hip)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (testing)")
    ap.add_argument("--device-index", type=int, default=None,
                    help="testing: put every rank on this GPU instead of LOCAL_RANK")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="testing (no GPU needed): only the rank launch + barrier + max-over-ranks plumbing, over gloo")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.selftest_launch:
        sys.exit(selftest_launch(args))
    if args.lib:
        from locator_amd import _lib as _loc_lib
        _loc_lib.use_library(os.path.abspath(args.lib))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.device_index is not None:
        local = args.device_index
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(args.dist_backend)

    from locator_amd import _lib
    from locator_amd.net import LocatorNet, gather_columns, upload_genotypes
    from locator_amd.synth import normalize_locs, split_indices, synth_genotypes
    from locator_amd.train import FitLoop

    K, H, n, R = args.snps, args.width, args.n, max(1, args.replicates_per_gpu)
    x, locs = synth_genotypes(n, K, seed=20260101, n_na=n // 10)
    train, test, pred = split_indices(locs, 0.9, seed=12345)
    _, _, _, _, ynorm = normalize_locs(locs)
    X0 = upload_genotypes(x, dev)
    Y = torch.from_numpy(np.nan_to_num(ynorm).astype(np.float32)).to(dev)

    class Fit:
        """One model fit: its matrix (a bootstrap resample unless it is THE single fit), net, epoch runner, callbacks
        and stream."""

        def __init__(self, replicate):
            X = X0
            if world > 1 or R > 1 or replicate > 0:
                # every replicate resamples the SNP columns on device (locator.py:648-653)
                so = np.random.RandomState(1000 + replicate).choice(K, K, replace=True)
                X = gather_columns(X0, so, K)
            self.net = LocatorNet(X, Y, K, H, 10, 0.25, seed=12345, replicate=replicate, device=dev,
                                  tuning=({"l1b_nt_mask": args.nt_mask} if args.nt_mask else {}) |
                                         ({"chain_tail": -1} if args.separate_tail else {}) |
                                         ({"stack_helpers": args.stack_helpers} if args.stack_helpers else {}) |
                                         ({"stack_xcd_stride": args.stack_xcd_stride} if args.stack_xcd_stride else {}) |
                                         ({"stack_train_rows": args.stack_train_rows} if args.stack_train_rows else {}))
            if args.l1_bwd_grid:
                self.net.l1_bwd_grid = int(args.l1_bwd_grid)
            # the product's epoch loop (locator_amd/train.py FitLoop): callbacks on the device, epochs enqueued ahead of it.
            # Early stopping is parked (patience 10^6) so that exactly --steps epochs are timed; checkpoint copies and the
            # LR plateau (patience 16 = int(100 / 6), locator.py:354) are live.
            self.rng = np.random.default_rng(99 + replicate)
            self.loop = FitLoop(self.net, train, test, batch_size=args.batch, max_epochs=args.steps + max(args.warmup, 2) + 64, patience=10 ** 6,
                                lr_patience=16, use_graph=not args.no_graph, chain=False if args.no_chain else None,
                                perm_fn=lambda e: self.rng.permutation(len(train)), depth=0 if args.sync_epochs else 2, xchain=not args.no_xchain, side_stats=args.side_stats)
            self.runner = self.loop.runner
            self.stream = torch.cuda.Stream(device=dev) if R > 1 else torch.cuda.current_stream()

        @property
        def hist(self):
            h = self.loop.hist.history
            return list(zip(h["loss"], h["val_loss"]))

        def start(self, ev=None):
            with torch.cuda.stream(self.stream):
                self.loop.submit(ev)

        def finish(self, lag=None):
            with torch.cuda.stream(self.stream):
                self.loop.collect(lag)

    fits = [Fit(rank * R + r) for r in range(R)]
    torch.cuda.synchronize()
    n_train, steps_per_epoch = fits[0].runner.n_train, fits[0].runner.steps

    def epoch(ev=None):
        for f in fits:
            f.start(ev)
        for f in fits:
            f.finish(0 if ev is not None else None)        # instrumented epochs are read back at once

    def drain():
        for f in fits:
            f.finish(0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    warm = max(args.warmup, 2)                 # epoch 0 eager, epoch 1 captures the graph
    for _ in range(warm):
        epoch()
    drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        epoch()
    drain()                                    # every timed epoch's history row has been read back
    barrier()
    elapsed = time.perf_counter() - t0
    if args.epoch_times:
        es = fits[0].loop.hist.epoch_seconds[-args.steps:]
        print("epoch completion intervals of the timed epochs, ms:", " ".join(f"{1e3 * v:.2f}" for v in es), file=sys.stderr)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- dominant-kernel timing: HIP events around every l1_bwd_adam launch, recorded on the launch stream
    roof = None
    if rank == 0:
        lib = _lib.load()
        import ctypes as C
        f0 = fits[0]
        n_ep = 2
        evs = []
        for _ in range(n_ep * steps_per_epoch * 2):
            h = C.c_void_p()
            _lib.check(lib.loc_event_create(C.byref(h)))
            evs.append(h)
        ms, by = [], []
        chain_groups = max(1, min(f0.net.l1_bwd_grid // 2, f0.net.d.Kp // 32))     # api.hip: chain_grid_of
        for k in range(n_ep):
            pairs = [(evs[2 * (k * steps_per_epoch + j)], evs[2 * (k * steps_per_epoch + j) + 1])
                     for j in range(steps_per_epoch)]
            f0.start(pairs)
            f0.finish(0)
            for j, (a, b) in enumerate(pairs):
                o = C.c_float()
                _lib.check(lib.loc_event_elapsed_ms(a, b, C.byref(o)))
                ms.append(o.value)
                nb = int(f0.runner.step_sizes[j])
                if f0.runner.chain:
                    nb_next = int(f0.runner.step_sizes[j + 1]) if j + 1 < steps_per_epoch else 0
                    by.append(l1_chain_bytes(K, H, nb, nb_next, chain_groups, 0 if args.separate_tail else 10))
                else:
                    by.append(l1_bwd_bytes(K, H, nb))
        for h in evs:
            lib.loc_event_destroy(h)
        t_mean = float(np.mean(ms)) * 1e-3
        achieved = float(np.mean(by)) / t_mean / 1e9
        kname = "l1_bwd_adam_chain_kernel" if f0.runner.chain else "l1_bwd_adam_kernel<%d" % ((H + 31) // 32)
        traffic, traffic_source = measured_traffic(K, H, n, kname, args.batch)
        roof = {"bound": "hbm", "kernel": kname.split("<")[0], "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_source, "bytes_per_launch": int(np.mean(by)),
                "us_per_launch": round(t_mean * 1e6, 2), "launches_timed": len(ms)}

    if rank == 0:
        value = world * R * args.steps * n_train / elapsed
        ms_epoch = elapsed / args.steps * 1e3
        step_b = sum(step_bytes(K, H, int(s)) for s in fits[0].runner.step_sizes)
        out = {
            "metric": "training samples/sec on 1000x100k-SNP matrix",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": warm, "ms_per_step": round(ms_epoch, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {n} ind x {K} SNPs uint8 (BASELINE.json configs[2]), "
                                   f"{'single model fit' if R == 1 else str(R) + ' concurrent replicate fits'} per GPU, "
                                   f"batch {args.batch}, {n_train} train / {len(test)} validation",
                       "step": f"one epoch = {steps_per_epoch} minibatch steps + validation sweep + device-side callbacks + checkpoint copy"
                               + ("" if R == 1 else f", of each of the {R} fits"),
                       "width": H, "nlayers": 10, "graph": not args.no_graph,
                       "chained_steps": bool(fits[0].runner.chain), "epochs_in_flight": 0 if args.sync_epochs else 2,
                       "replicates_per_gpu": R,
                       "replicates": ("single model" if world == 1 and R == 1 else
                                      f"{R} model(s) per GPU, bootstrap resample per replicate")},
            "us_per_minibatch_step": round(ms_epoch * 1e3 / steps_per_epoch / R, 2),
            "whole_step_hbm_frac": round(R * step_b / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "final_loss": round(fits[0].hist[-1][0], 5), "final_val_loss": round(fits[0].hist[-1][1], 5),
            "roofline": roof,
        }
        if world == 1 and not args.no_l1_gemm:
            out["l1_gemm"] = l1_gemm_summary(fits[0].net, n)
            if args.l1_gemm_full:
                # the full predict-mode x shape sweep goes to a FILE, never into the line (round 4's 24 KB line was unparseable)
                from tools import l1_gemm_sweep
                full = l1_gemm_sweep.l1_gemm_roofline(fits[0].net, n, x_distinct=l1_gemm_sweep.distinct_rows(dev, fits[0].net.d.Kp))
                full["tolerances"] = l1_gemm_sweep.PREDICT_MODE_INFO
                with open(args.l1_gemm_full, "w") as f:
                    json.dump(full, f, indent=1)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(x, np.nan_to_num(ynorm), train, test, K, H, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        print(format_line(out), flush=True)          # the LAST thing this process prints
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
