"""Synthetic genotype matrices for the benchmark configurations.

Not part of the reference: the reference ships one small msprime fixture and no generator.  The
generator below gives a matrix with the fixture's allele-frequency skew and a spatial signal, and
applies the reference's SNP filters (biallelic + allele-1 count >= min_mac, locator.py:265-273)
by rejection so exactly K SNPs remain.
"""
from __future__ import annotations

import numpy as np


def synth_genotypes(n=1000, K=100_000, seed=20260101, n_na=100, min_mac=2, chunk=20_000):
    """Returns (x [n, K] uint8 allele counts, locs [n, 2] float64 with NaN for the first n_na samples)."""
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 50, (n, 2))
    gx = ((xy[:, 0] - 25) / 25).astype(np.float32)[:, None]
    gy = ((xy[:, 1] - 25) / 25).astype(np.float32)[:, None]
    out = np.empty((n, K), np.uint8)
    have = 0
    while have < K:
        pk = np.clip(rng.beta(0.3, 0.9, chunk), 0.002, 0.998).astype(np.float32)
        a = rng.normal(0, 0.1, chunk).astype(np.float32)
        b = rng.normal(0, 0.1, chunk).astype(np.float32)
        p = np.clip(pk[None, :] + a[None, :] * gx + b[None, :] * gy, 0, 1)
        g = (rng.random((n, chunk), dtype=np.float32) < p).astype(np.uint8)
        g += (rng.random((n, chunk), dtype=np.float32) < p).astype(np.uint8)
        alt = g.sum(0, dtype=np.int64)
        keep = (alt >= max(min_mac, 1)) & (alt < 2 * n)          # both alleles seen, allele-1 count >= min_mac
        g = g[:, keep]
        take = min(K - have, g.shape[1])
        out[:, have:have + take] = g[:, :take]
        have += take
    locs = xy.copy()
    locs[:n_na] = np.nan
    return out, locs


def split_indices(locs, train_split=0.9, seed=12345):
    """The reference's split (locator.py:295-308) on a private legacy RandomState: known rows ->
    validation = choice(known, round((1-train_split)*n_known), replace=False), train = the rest
    (ascending), pred = NaN rows."""
    rs = np.random.RandomState(seed)
    known = np.argwhere(~np.isnan(locs[:, 0]))[:, 0]
    pred = np.argwhere(np.isnan(locs[:, 0]))[:, 0]
    test = rs.choice(known, round((1 - train_split) * len(known)), replace=False)
    tset = set(test.tolist())
    train = np.array([i for i in known if i not in tset])
    return train, test, pred


def normalize_locs(locs):
    """locator.py:284-292."""
    meanlong, sdlong = np.nanmean(locs[:, 0]), np.nanstd(locs[:, 0])
    meanlat, sdlat = np.nanmean(locs[:, 1]), np.nanstd(locs[:, 1])
    out = np.stack([(locs[:, 0] - meanlong) / sdlong, (locs[:, 1] - meanlat) / sdlat], axis=1)
    return meanlong, sdlong, meanlat, sdlat, out
