"""Where does a many-row predict spend its time?  (loc_predict: weight image build, layer-1 GEMM, hidden stack + heads.)
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split; prints the end-to-end time per row count itself.
    python3 tools/predict_timeline.py [--rows 1000,4096,16384] [--mode auto|exact|fast]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="1000,4096,16384")
    ap.add_argument("--snps", type=int, default=100_000)
    ap.add_argument("--mode", default="auto")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--lib", default=None, help="measurement switch: another build of liblocator_hip.so")
    a = ap.parse_args()
    if a.lib:
        from locator_amd import _lib
        _lib.use_library(a.lib)
    import torch

    from locator_amd.net import LocatorNet

    K = a.snps
    g = torch.Generator(device="cuda").manual_seed(1)
    nmax = max(int(r) for r in a.rows.split(","))
    Kp = (K + 31) // 32 * 32
    X = torch.zeros((nmax, Kp), dtype=torch.uint8, device="cuda")
    for r0 in range(0, nmax, 4096):
        u = torch.rand((min(4096, nmax - r0), Kp), device="cuda", generator=g)
        X[r0:r0 + u.shape[0]] = (u < 0.25).to(torch.uint8) + (u < 0.08).to(torch.uint8)
    X[:, K:] = 0
    Y = torch.zeros((nmax, 2), device="cuda")
    net = LocatorNet(X, Y, K, 256, 10, 0.25, seed=3, predict_digits={"auto": 0, "exact": 3, "fast": 2}[a.mode])
    for n in (int(r) for r in a.rows.split(",")):
        rows = torch.arange(n, dtype=torch.int32, device="cuda")
        yhat = torch.zeros((n, 2), device="cuda")
        net.predict_rows(rows, n, yhat)          # builds the image, packs the matrix
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            net.predict_rows(rows, n, yhat)      # image kept: GEMM + hidden stack only
        torch.cuda.synchronize()
        warm = (time.perf_counter() - t0) / a.iters
        t0 = time.perf_counter()
        for _ in range(a.iters):
            net.params_changed()
            net.predict_rows(rows, n, yhat)      # as after a fit: scan + guard + image + GEMM + hidden stack
        torch.cuda.synchronize()
        cold = (time.perf_counter() - t0) / a.iters
        print(json.dumps({"rows": n, "snps": K, "mode": a.mode, "predict_us_image_kept": round(warm * 1e6, 1),
                          "predict_us_weights_changed": round(cold * 1e6, 1), "guard": net._guard}), flush=True)


if __name__ == "__main__":
    main()
