#!/usr/bin/env python3
"""Synthetic `allel.vcf_to_zarr`-shaped store for --windows runs (BASELINE.json configs[3], scaled by flags):
one chromosome, `--windows` windows of `--window_size` bp with `--per_window` variants each, 765 samples
(Ag1000G phase-1 size), 10 % of samples without coordinates.  Writes <out>.zarr and <out>_samples.txt."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locator_amd import genotypes as G  # noqa: E402


def make_store(out, n=765, windows=25, window_size=2_000_000, per_window=150_000, seed=20260104, compressor=None):
    """Writes <out>.zarr and <out>_samples.txt; returns (true coordinates [n, 2], indices of the samples written as NA)."""
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 50, (n, 2))
    gx = ((xy[:, 0] - 25) / 25).astype(np.float32)[None, :]
    gy = ((xy[:, 1] - 25) / 25).astype(np.float32)[None, :]
    V = windows * per_window
    gt = np.empty((V, n, 2), np.int8)
    pos = np.empty(V, np.int32)
    for w in range(windows):
        m = per_window
        pk = np.clip(rng.beta(0.3, 0.9, m), 0.002, 0.998).astype(np.float32)[:, None]
        p = np.clip(pk + rng.normal(0, 0.1, (m, 1)).astype(np.float32) * gx
                    + rng.normal(0, 0.1, (m, 1)).astype(np.float32) * gy, 0, 1)
        sl = slice(w * m, (w + 1) * m)
        gt[sl, :, 0] = rng.random((m, n), dtype=np.float32) < p
        gt[sl, :, 1] = rng.random((m, n), dtype=np.float32) < p
        pos[sl] = np.sort(rng.choice(window_size, m, replace=False)) + w * window_size + 1
    samples = np.array([f"AB{i:04d}" for i in range(n)])
    G.write_callset_zarr(out + ".zarr", gt, pos, samples, chunk_variants=65536, compressor=compressor)
    na = rng.choice(n, n // 10, replace=False)
    locs = xy.copy()
    locs[na] = np.nan
    with open(out + "_samples.txt", "w") as fh:
        fh.write("sampleID\tx\ty\n")
        for s, (x, y) in zip(samples, locs):
            fh.write(f"{s}\t{'NA' if np.isnan(x) else x}\t{'NA' if np.isnan(y) else y}\n")
    return xy, np.sort(na)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--n", type=int, default=765)
    ap.add_argument("--windows", type=int, default=25)
    ap.add_argument("--window_size", type=int, default=2_000_000)
    ap.add_argument("--per_window", type=int, default=150_000)
    ap.add_argument("--seed", type=int, default=20260104)
    ap.add_argument("--compressor", default=None, choices=[None, "zlib"])
    a = ap.parse_args()
    make_store(a.out, a.n, a.windows, a.window_size, a.per_window, a.seed, a.compressor)
    print(f"wrote {a.out}.zarr: {a.windows * a.per_window} variants x {a.n} samples, {a.windows} windows")


if __name__ == "__main__":
    main()
