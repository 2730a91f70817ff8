#!/usr/bin/env python3
"""Summarise a set of replicate predictions (the step right after --windows / --bootstrap).

Reference: /root/reference/locator_py/plot_locator.py:26-44, :59-126 and scripts/plot_locator.R:57-113
(the reference computes these inside its plotting scripts).  For every sample that appears in the
`*predlocs*` files of a directory:

  gc_x, gc_y   geographic centroid = mean of the replicate predictions        (plot_locator.py:39-44)
  kd_x, kd_y   the replicate prediction with the highest Gaussian kernel density, bandwidth 0.2,
               density evaluated at the predictions themselves (sklearn KernelDensity.score_samples in
               the reference, plot_locator.py:26-37); falls back to the mean if the estimate fails

and, when the sample file has known coordinates, the error of both summaries.  Output
`{out}_centroids.txt` is tab-separated with the reference's columns sampleID, x, y, kd_x, kd_y, gc_x, gc_y
(plot_locator.py:117-119).  Plotting itself is out of scope.

Deviation: the reference's file loop reads `files[i]` for i in range(len(files[1:])) (plot_locator.py:67-70),
i.e. the first file twice and never the last; here every file is read once.

Every sample's kernel-density peak and centroid come from ONE launch of the library's `loc_kde_peak_batch` (one workgroup
per sample, float64, include/locator_hip.h; `device_summaries`).  There is no silent fallback: without a GPU (or without the
library) the run stops and says so.  `--host` asks for the NumPy form explicitly (`kde_peak` / `centroid` below) - a
post-processing convenience for a machine without a GPU, and what tests/test_gpu_summarize.py checks the kernel against.
"""
from __future__ import annotations

import argparse
import os

import numpy as np


def kde_peak(x, y, bandwidth=0.2):
    """Index-of-max Gaussian KDE over the points themselves; first maximum wins (np.argwhere order)."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if len(x) == 0 or not (np.isfinite(x).all() and np.isfinite(y).all()):
        return float(np.mean(x)) if len(x) else np.nan, float(np.mean(y)) if len(y) else np.nan
    d2 = (x[:, None] - x[None, :]) ** 2 + (y[:, None] - y[None, :]) ** 2
    a = -d2 / (2.0 * bandwidth * bandwidth)
    m = a.max(axis=1, keepdims=True)
    score = (m[:, 0] + np.log(np.exp(a - m).sum(axis=1)))      # log-density up to a constant
    i = int(np.argmax(score))
    return float(x[i]), float(y[i])


def centroid(x, y):
    return float(np.sum(x) / len(x)), float(np.sum(y) / len(y))


def device_summaries(groups, bandwidth=0.2, device="cuda:0"):
    """[(x array, y array), ...] -> (peak index, [kd_x, kd_y, gc_x, gc_y]) per sample from one loc_kde_peak_batch launch.
    peak index -1 = no density estimate (non-finite coordinate / no points): kd = centroid, as kde_peak reports it."""
    import torch
    from . import _lib
    lib = _lib.load()
    n = len(groups)
    counts = np.array([len(x) for x, _ in groups], dtype=np.int64)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    xy = np.empty((int(offsets[-1]), 2), dtype=np.float64)
    for (x, y), o0, o1 in zip(groups, offsets[:-1], offsets[1:]):
        xy[o0:o1, 0], xy[o0:o1, 1] = x, y
    with torch.cuda.device(device):
        d_xy = torch.from_numpy(xy).to(device)
        d_off = torch.from_numpy(offsets).to(device)
        d_idx = torch.empty(n, dtype=torch.int32, device=device)
        d_out = torch.empty((n, 4), dtype=torch.float64, device=device)
        _lib.check(lib.loc_kde_peak_batch(d_xy.data_ptr(), d_off.data_ptr(), n, float(bandwidth), d_idx.data_ptr(),
                                          d_out.data_ptr(), torch.cuda.current_stream().cuda_stream), "loc_kde_peak_batch")
        return d_idx.cpu().numpy(), d_out.cpu().numpy()


def summarize(indir, sample_data=None, out=None, bandwidth=0.2, silence=False, host=False):
    import pandas as pd
    files = sorted(f for f in os.listdir(indir) if "predlocs" in f)
    if not files:
        raise SystemExit(f"no *predlocs* files in {indir}")
    aeg = pd.concat([pd.read_csv(os.path.join(indir, f)) for f in files], ignore_index=True)
    aeg = aeg.rename(columns={"x": "xpred", "y": "ypred"})
    truth = None
    if sample_data is not None:
        truth = pd.read_csv(sample_data, sep="\t").set_index("sampleID")
    rows = []
    groups = [(sid, grp["xpred"].to_numpy(), grp["ypred"].to_numpy()) for sid, grp in aeg.groupby("sampleID", sort=False)]
    dev = None
    if not host:
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("locator_amd.summarize: no GPU visible (the summaries are one loc_kde_peak_batch launch); "
                             "pass --host for the NumPy form")
        dev = device_summaries([(x, y) for _, x, y in groups], bandwidth)[1]
    for i, (sid, xs, ys) in enumerate(groups):
        if dev is not None:
            kx, ky, gx, gy = (float(v) for v in dev[i])
        else:
            kx, ky = kde_peak(xs, ys, bandwidth)
            gx, gy = centroid(xs, ys)
        tx = ty = np.nan
        if truth is not None and sid in truth.index:
            tx, ty = float(truth.loc[sid, "x"]), float(truth.loc[sid, "y"])
        rows.append({"sampleID": sid, "x": tx, "y": ty, "kd_x": kx, "kd_y": ky, "gc_x": gx, "gc_y": gy})
    bp = pd.DataFrame(rows, columns=["sampleID", "x", "y", "kd_x", "kd_y", "gc_x", "gc_y"])
    if out is not None:
        bp.to_csv(out + "_centroids.txt", index=False, sep="\t")
    known = bp.dropna(subset=["x", "y"])
    if len(known) and not silence:
        kd = np.hypot(known.kd_x - known.x, known.kd_y - known.y)
        gc = np.hypot(known.gc_x - known.x, known.gc_y - known.y)
        print("mean kernel peak error = " + str(np.mean(kd)))
        print("median kernel peak error = " + str(np.median(kd)))
        print("mean centroid error = " + str(np.mean(gc)))
        print("median centroid error = " + str(np.median(gc)))
    return bp


def main(argv=None):
    ap = argparse.ArgumentParser(description="Per-sample centroid and kernel-density peak over replicate predictions.")
    ap.add_argument("--infile", required=True, help="directory holding the *predlocs* files")
    ap.add_argument("--sample_data", default=None, help="sample file with known x / y (optional)")
    ap.add_argument("--out", required=True, help="output stem ({out}_centroids.txt)")
    ap.add_argument("--bandwidth", type=float, default=0.2)
    ap.add_argument("--silence", action="store_true")
    ap.add_argument("--host", action="store_true", help="NumPy summaries even when a GPU is visible")
    a = ap.parse_args(argv)
    summarize(a.infile, a.sample_data, a.out, a.bandwidth, a.silence, host=a.host)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
