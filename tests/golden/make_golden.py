#!/usr/bin/env python3
"""Generates tests/golden/oracle_*.npz — "restated-oracle" vectors (NOT outputs of the reference:
the reference cannot be run here, see oracle/locator_oracle.py's header).  Inputs and expected
outputs only: a toy problem and a slice of the reference's example VCF, fixed init, fixed
permutations, fixed dropout masks; expected loss, gradients, weights after 1 and 5 Adam steps,
BN moving statistics; callback traces for a scripted val_loss sequence.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import locator_oracle as O  # noqa: E402


def flat(p, prefix):
    d = {f"{prefix}gamma": p["gamma"], f"{prefix}beta": p["beta"]}
    if "mov_mean" in p:
        d[f"{prefix}mov_mean"], d[f"{prefix}mov_var"] = p["mov_mean"], p["mov_var"]
    for i, (w, b) in enumerate(zip(p["W"], p["b"])):
        d[f"{prefix}W{i}"], d[f"{prefix}b{i}"] = w, b
    return d


def case(name, x, y, width, nlayers, drop_p, seed):
    rng = np.random.default_rng(seed)
    n, K = x.shape
    p0 = O.init_params(K, width, nlayers, rng)
    p0["gamma"] = rng.uniform(0.8, 1.2, K)
    p0["beta"] = rng.normal(0, 0.05, K)
    out = {"x": x, "y": y, "width": width, "nlayers": nlayers, "drop_p": drop_p}
    out.update(flat(p0, "p0_"))
    batches = [rng.choice(n, min(32, n), replace=False) for _ in range(4)] + [rng.choice(n, 11, replace=False)]
    masks = [(rng.random((len(b), width)) >= drop_p).astype(np.uint8) for b in batches]
    out["batches"] = np.array([np.pad(b, (0, 32 - len(b)), constant_values=-1) for b in batches])
    out["masks"] = np.array([np.pad(m, ((0, 32 - len(m)), (0, 0))) for m in masks])
    p = O.copy_params(p0)
    loss, g, yhat = O.loss_and_grads(O.copy_params(p0), x[batches[0]], y[batches[0]], masks[0], drop_p)
    out["loss0"], out["yhat0"] = loss, yhat
    out.update(flat(g, "g0_"))
    m, v = O.zeros_like_trainable(p), O.zeros_like_trainable(p)
    losses = []
    for t, (b, mk) in enumerate(zip(batches, masks), start=1):
        losses.append(O.train_step(p, m, v, t, 1e-3, x[b], y[b], mk, drop_p))
        if t == 1:
            out.update(flat(O.copy_params(p), "p1_"))     # copy: Adam updates p in place
    out["losses"] = np.array(losses)
    out.update(flat(p, "p5_"))
    out["pred5"] = O.predict(p, x)
    np.savez_compressed(os.path.join(HERE, f"oracle_{name}.npz"), **out)


def main():
    rng = np.random.default_rng(20260101)
    x = rng.integers(0, 3, (40, 64)).astype(np.uint8)
    y = rng.normal(0, 1, (40, 2))
    case("toy_16x64", x, y, 32, 4, 0.25, 1)
    # slice of the reference's own example data: 64 samples x 512 filtered SNPs
    from locator_amd import genotypes as G
    v = G.read_vcf(os.path.join(HERE, "test_genotypes.vcf.gz"))
    ac = G.filter_snps(v["calldata/GT"], 2, verbose=False)
    xs = np.ascontiguousarray(ac[1000:1512, 50:114].T).astype(np.uint8)
    import pandas as pd
    sd = pd.read_csv(os.path.join(HERE, "test_sample_data.txt"), sep="\t")
    locs = np.array(sd[["x", "y"]])[50:114]
    _, _, _, _, ys = O.normalize_locs(locs)
    case("fixture_64x512", xs, ys, 64, 10, 0.25, 2)
    # callback traces for a scripted val_loss sequence
    cb = O.Callbacks(patience=12)
    vals = np.array([1.0, 0.9, 0.95, 0.96, 0.97, 0.9, 0.8] + [0.85] * 12 + [0.7])
    tr = [cb.on_epoch_end(e, float(vv)) for e, vv in enumerate(vals)]
    np.savez(os.path.join(HERE, "oracle_callbacks.npz"), val_loss=vals, save=np.array([t[0] for t in tr]),
             stop=np.array([t[1] for t in tr]), lr=np.array([t[2] for t in tr]))


if __name__ == "__main__":
    main()
