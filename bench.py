#!/usr/bin/env python3
"""Headline benchmark: training samples/s of one locator model fit on a synthetic
1,000-individual x 100,000-SNP genotype matrix (BASELINE.json configs[2]), per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" here is ONE EPOCH of model.fit on the 810-row training split: 26 minibatch steps of
batch 32 (25 full + one of 10, kept as Keras keeps it), the validation sweep over the 90 held-out
rows, and the host-side callbacks (ModelCheckpoint snapshot when val_loss improves, LR plateau).
Inputs (genotypes, targets, weights) are resident in HBM before the timed region starts.

N > 1: one process per GPU, each fitting its OWN bootstrap replicate of the same matrix
(locator.py:635-681: replicates are independent fits) — no data-path collective; torch.distributed
is used only for the barrier and the max-over-ranks of the elapsed time.  scaling = "weak".

The JSON line also carries
  roofline      the dominant kernel (l1_bwd_adam: fused layer-1 backward + Adam), algorithmic bytes per
                launch / its mean duration from HIP events recorded on the launch stream immediately before and
                after that kernel, vs 8 TB/s HBM; `traffic` = PMC-measured bytes (profiles/r01_pmc_traffic.json)
  cpu_baseline  the NumPy fp32 port of the same step (oracle/) timed on this box's host cores on a
                bounded sample of the same workload.
  l1_gemm       (N=1) the large-M first-layer genotype GEMM of the predict / validation sweeps over all rows,
                as a fraction of the dense bf16-MFMA peak (outside the timed region; events on the stream).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BF16_PEAK_TFLOPS = 2500.0    # same guide: dense bf16 MFMA peak (no sparsity)


def l1_bwd_bytes(K, H, n_b):
    """Algorithmic HBM bytes of one l1_bwd_adam launch: W1,m,v read + written once
    (24 B/weight), the batch's uint8 genotype rows once, BN scale/shift/mean/rstd read (16 B/SNP),
    gamma,beta and their Adam moments read + written (48 B/SNP), dZ1 and b1 state."""
    return 24 * K * H + n_b * K + 64 * K + 4 * 32 * H + 24 * H


def step_bytes(K, H, n_b, L=10):
    """SURVEY.md §8(d) bytes_step for a whole minibatch step."""
    return 28 * K * H + 2 * n_b * K + 64 * K + 28 * ((L - 1) * H * H + (L + 1) * H + 8)


def l1_gemm_roofline(net, n_rows, iters=20):
    """The only large-M contraction on the path (model.predict / validation, locator.py:414, :441):
    a1 = ELU(BN(x) W1 + b1) for n_rows rows at once through loc_l1_forward_rows, timed with HIP events.
    flops = 2*M*K*H counted ONCE, however many bf16 pieces carry each fp32 weight (3 = exact products)."""
    import ctypes as C

    import torch
    from locator_amd import _lib
    lib, d, lay = net.lib, net.d, net.lay
    P = net.params.data_ptr()
    dev = net.params.device
    bn4 = torch.zeros(4 * d.Kp, device=dev)
    _lib.check(lib.loc_bn_infer_scale_shift(d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta, P + 4 * lay.mov_mean,
                                            P + 4 * lay.mov_var, bn4.data_ptr(), None))
    partial = torch.empty(256 * 128 * d.Hp, device=dev)
    rows = torch.arange(n_rows, dtype=torch.int32, device=dev)
    a1 = torch.empty(((n_rows + 127) // 128 * 128, d.Hp), device=dev)
    out = {"rows": n_rows, "flops": 2.0 * n_rows * d.K * d.H, "bytes": float(n_rows * d.K + 4 * d.K * d.H),
           "peak_tflops": BF16_PEAK_TFLOPS, "kernel": "l1_rows_partial_kernel + l1_reduce_kernel"}
    for pieces in (3, 1):
        if not lib.loc_l1_rows_supported(d.Hp, pieces):
            continue

        def run():
            _lib.check(lib.loc_l1_forward_rows(net.X.data_ptr(), net.X.stride(0), rows.data_ptr(), n_rows,
                                               C.byref(d), bn4.data_ptr(), P + 4 * lay.w1, P + 4 * lay.b1,
                                               partial.data_ptr(), partial.numel(), a1.data_ptr(), pieces, 0, None,
                                               None))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        tf = out["flops"] / us * 1e-6
        out["bf16x%d" % pieces] = {"us": round(us, 1), "tflops": round(tf, 1),
                                   "frac_bf16_peak": round(tf / BF16_PEAK_TFLOPS, 4),
                                   "mfma_issue_frac": round(pieces * tf / BF16_PEAK_TFLOPS, 4),
                                   "hbm_gbs": round(out["bytes"] / us * 1e-3, 1),
                                   "exact_fp32_products": pieces == 3}
    return out


def cpu_baseline(x, y_norm, train, K, H, seconds=20.0):
    """Oracle (NumPy fp32 port of the same training step) on the host cores; bounded sample."""
    from oracle import locator_oracle as O
    rng = np.random.default_rng(0)
    p = O.init_params(K, H, 10, rng, dtype=np.float32)
    m, v = O.zeros_like_trainable(p), O.zeros_like_trainable(p)
    xt, yt = x[train], y_norm[train].astype(np.float32)
    n_done, t_used, t = 0, 0.0, 0
    perm = rng.permutation(len(train))
    steps = 0
    t0 = time.perf_counter()
    while True:
        rows = perm[(steps * 32) % (len(train) - 32):][:32]
        mask = rng.random((32, H)) >= 0.25
        t += 1
        O.train_step(p, m, v, t, np.float32(1e-3), xt[rows], yt[rows], mask, 0.25)
        steps += 1
        n_done += 32
        t_used = time.perf_counter() - t0
        if t_used >= seconds and steps >= 3:
            break
    try:
        import threadpoolctl
        cores = max([i.get("num_threads", 1) for i in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count()
    return {"value": n_done / t_used, "unit": "samples/s", "cores": int(cores), "kind": "port",
            "sample": f"{steps} minibatch steps of 32 rows x {K} SNPs (NumPy fp32 oracle, BLAS threads = cores), "
                      f"{t_used:.1f} s; validation sweep not included"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed epochs")
    ap.add_argument("--warmup", type=int, default=3, help="untimed epochs")
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--snps", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32, help="--batch_size of the fit (headline: 32, the reference default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-graph", action="store_true", help="enqueue every kernel from the host each epoch")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (testing)")
    ap.add_argument("--device-index", type=int, default=None,
                    help="testing: put every rank on this GPU instead of LOCAL_RANK")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
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
    from locator_amd.train import Callbacks, EpochRunner

    K, H, n = args.snps, args.width, args.n
    x, locs = synth_genotypes(n, K, seed=20260101, n_na=n // 10)
    train, test, pred = split_indices(locs, 0.9, seed=12345)
    _, _, _, _, ynorm = normalize_locs(locs)
    X = upload_genotypes(x, dev)
    if world > 1 or rank > 0:
        # every rank fits its own bootstrap replicate: resample SNP columns on device (locator.py:648-653)
        so = np.random.RandomState(1000 + rank).choice(K, K, replace=True)
        X = gather_columns(X, so, K)
    Y = torch.from_numpy(np.nan_to_num(ynorm).astype(np.float32)).to(dev)
    net = LocatorNet(X, Y, K, H, 10, 0.25, seed=12345, replicate=rank, device=dev)
    runner = EpochRunner(net, train, test, args.batch, use_graph=not args.no_graph)
    cb = Callbacks(100, 1e-3)
    rng = np.random.default_rng(99 + rank)
    n_train, steps_per_epoch = runner.n_train, runner.steps
    hist = []

    def epoch(e, ev=None):
        loss, val = runner.run_epoch(rng.permutation(n_train), ev)
        save, stop, lr_logged = cb.on_epoch_end(e, val)
        if save:
            net.snapshot()
        if cb.lr != lr_logged:
            net.lr_t.fill_(cb.lr)
        hist.append((loss, val))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    e = 0
    for _ in range(max(args.warmup, 2)):       # epoch 0 eager, epoch 1 captures the graph
        epoch(e)
        e += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        epoch(e)
        e += 1
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- dominant-kernel timing: HIP events around every l1_bwd_adam launch, recorded on the launch stream
    roof = None
    if rank == 0:
        lib = _lib.load()
        import ctypes as C
        n_ep = 2
        evs = []
        for _ in range(n_ep * steps_per_epoch * 2):
            h = C.c_void_p()
            _lib.check(lib.loc_event_create(C.byref(h)))
            evs.append(h)
        ms, by = [], []
        for k in range(n_ep):
            pairs = [(evs[2 * (k * steps_per_epoch + j)], evs[2 * (k * steps_per_epoch + j) + 1])
                     for j in range(steps_per_epoch)]
            epoch(e, pairs)
            e += 1
            for j, (a, b) in enumerate(pairs):
                out = C.c_float()
                _lib.check(lib.loc_event_elapsed_ms(a, b, C.byref(out)))
                ms.append(out.value)
                by.append(l1_bwd_bytes(K, H, int(runner.step_sizes[j])))
        for h in evs:
            lib.loc_event_destroy(h)
        t_mean = float(np.mean(ms)) * 1e-3
        achieved = float(np.mean(by)) / t_mean / 1e9
        traffic = None
        try:    # PMC-measured HBM bytes per launch (separate rocprofv3 --pmc passes, profiles/r01_pmc_traffic.json)
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if pm["workload"] == {"K": K, "H": H, "n": n}:
                key = [k for k in pm["kernels"] if k.startswith("l1_bwd_adam_kernel<%d" % ((H + 31) // 32))][0]
                traffic = int(pm["kernels"][key]["traffic_bytes"])
        except Exception:
            traffic = None
        roof = {"bound": "hbm", "kernel": "l1_bwd_adam_kernel", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_per_launch": int(np.mean(by)), "us_per_launch": round(t_mean * 1e6, 2),
                "launches_timed": len(ms)}

    if rank == 0:
        value = world * args.steps * n_train / elapsed
        ms_epoch = elapsed / args.steps * 1e3
        step_b = sum(step_bytes(K, H, int(s)) for s in runner.step_sizes)
        out = {
            "metric": "training samples/sec on 1000x100k-SNP matrix",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 2), "ms_per_step": round(ms_epoch, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {n} ind x {K} SNPs uint8 (BASELINE.json configs[2]), single model "
                                   f"fit per GPU, batch {args.batch}, {n_train} train / {len(test)} validation",
                       "step": f"one epoch = {steps_per_epoch} minibatch steps + validation sweep + callbacks",
                       "width": H, "nlayers": 10, "graph": not args.no_graph,
                       "replicates": "1 model per GPU, bootstrap resample per rank" if world > 1 else "single model"},
            "us_per_minibatch_step": round(ms_epoch * 1e3 / steps_per_epoch, 2),
            "whole_step_hbm_frac": round(step_b / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "final_loss": round(hist[-1][0], 5), "final_val_loss": round(hist[-1][1], 5),
            "roofline": roof,
        }
        if world == 1:
            out["l1_gemm"] = l1_gemm_roofline(net, n)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(x, np.nan_to_num(ynorm), train, K, H, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
