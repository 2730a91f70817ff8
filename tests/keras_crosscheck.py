#!/usr/bin/env python3
"""Pins oracle/locator_oracle.py against REAL Keras, wherever TensorFlow exists (it does not in the build container
or on the GPU box: `import tensorflow` fails, BASELINE.md §2 — this script then says so and exits 3).

It is a build-owned Keras spelling of the reference's model — same layers, loss and optimizer string as
/root/reference/locator/locator.py:311-327 (BatchNormalization(input) -> floor(L/2) x Dense(width, elu) -> Dropout
-> ceil(L/2) x Dense(width, elu) -> Dense(2) -> Dense(2), loss sqrt(sum((y_pred - y_true)^2, -1)), optimizer "Adam")
— with everything Keras draws at random taken out of the comparison: weights are INJECTED from the oracle's init,
batches are given explicitly (train_on_batch), and Dropout is 0 (its mask stream cannot be injected; the oracle's
dropout arithmetic is pinned separately by autograd, tests/test_oracle.py).

What is compared, step by step over `--steps` Adam steps with a partial batch in the middle:
  batch loss, every kernel / bias / gamma / beta, BatchNorm moving mean / variance, and model.predict afterwards,
against oracle.train_step / oracle.predict in float32 and float64.  A report (max abs deviation per tensor) goes to
stdout and, with --out, the Keras-side tensors go to an .npz that tests/test_golden.py-style fixtures can be built
from (tests/golden/keras_*.npz would then pin the oracle: DESIGN.md §2 "parity unpinned" could be lifted).

    python tests/keras_crosscheck.py [--snps 300] [--width 32] [--nlayers 4] [--steps 5] [--out keras_vectors.npz]
"""
import argparse
import importlib.util
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_keras_model(tf, n_snps, width, nlayers, dropout_prop):
    K = tf.keras.backend

    def euclidean_distance_loss(y_true, y_pred):
        return K.sqrt(K.sum(K.square(y_pred - y_true), axis=-1))

    model = tf.keras.Sequential()
    model.add(tf.keras.layers.BatchNormalization(input_shape=(n_snps,)))
    for _ in range(int(np.floor(nlayers / 2))):
        model.add(tf.keras.layers.Dense(width, activation="elu"))
    model.add(tf.keras.layers.Dropout(dropout_prop))
    for _ in range(int(np.ceil(nlayers / 2))):
        model.add(tf.keras.layers.Dense(width, activation="elu"))
    model.add(tf.keras.layers.Dense(2))
    model.add(tf.keras.layers.Dense(2))
    model.compile(optimizer="Adam", loss=euclidean_distance_loss)
    return model


def inject(model, p):
    """oracle params -> Keras layer weights (kernel orientation in x out is the same)."""
    bn = model.layers[0]
    bn.set_weights([p["gamma"], p["beta"], p["mov_mean"], p["mov_var"]])
    dense = [l for l in model.layers if l.__class__.__name__ == "Dense"]
    assert len(dense) == len(p["W"])
    for l, w, b in zip(dense, p["W"], p["b"]):
        l.set_weights([w, b])


def extract(model):
    g, be, mm, mv = model.layers[0].get_weights()
    dense = [l for l in model.layers if l.__class__.__name__ == "Dense"]
    return {"gamma": g, "beta": be, "mov_mean": mm, "mov_var": mv,
            "W": [l.get_weights()[0] for l in dense], "b": [l.get_weights()[1] for l in dense]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=300)
    ap.add_argument("--width", type=int, default=32)
    ap.add_argument("--nlayers", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if importlib.util.find_spec("tensorflow") is None:
        print("keras_crosscheck: TensorFlow is not installed here; nothing compared.  Run this script wherever the "
              "reference runs (pip install tensorflow numpy) and keep the printed report / --out file.")
        return 3
    import tensorflow as tf

    from oracle import locator_oracle as O
    rng = np.random.default_rng(7)
    n = 96
    x = rng.integers(0, 3, (n, a.snps)).astype(np.uint8)
    y = rng.normal(0, 1, (n, 2)).astype(np.float32)
    p0 = O.init_params(a.snps, a.width, a.nlayers, rng, dtype=np.float32)
    p0["gamma"] = rng.uniform(0.7, 1.3, a.snps).astype(np.float32)
    p0["beta"] = rng.normal(0, 0.05, a.snps).astype(np.float32)
    p0["mov_mean"] = rng.uniform(0, 1, a.snps).astype(np.float32)
    p0["mov_var"] = rng.uniform(0.2, 1.2, a.snps).astype(np.float32)
    model = build_keras_model(tf, a.snps, a.width, a.nlayers, 0.0)
    inject(model, p0)
    ref = {dt: (O.cast_params(p0, dt), None, None) for dt in (np.float32, np.float64)}
    ref = {dt: (p, O.zeros_like_trainable(p), O.zeros_like_trainable(p)) for dt, (p, _, _) in ref.items()}
    batches = [rng.choice(n, 32 if s != a.steps // 2 else 13, replace=False) for s in range(a.steps)]
    dump = {"x": x, "y": y, "batches": np.array([np.pad(b, (0, 32 - len(b)), constant_values=-1) for b in batches])}
    worst = 0.0
    for t, rows in enumerate(batches, start=1):
        k_loss = float(model.train_on_batch(x[rows].astype(np.float32), y[rows]))
        got = extract(model)
        for dt, (p, m, v) in ref.items():
            o_loss = float(O.train_step(p, m, v, t, 1e-3, x[rows], y[rows].astype(dt), None, 0.0))
            dev = {"loss": abs(k_loss - o_loss)}
            for k in ("gamma", "beta", "mov_mean", "mov_var"):
                dev[k] = float(np.abs(got[k].astype(np.float64) - p[k]).max())
            for i in range(len(p["W"])):
                dev[f"W{i}"] = float(np.abs(got["W"][i].astype(np.float64) - p["W"][i]).max())
                dev[f"b{i}"] = float(np.abs(got["b"][i].astype(np.float64) - p["b"][i]).max())
            worst = max(worst, max(dev.values())) if dt is np.float64 else worst
            print(f"step {t} n_b={len(rows):2d} vs oracle {np.dtype(dt).name}: loss dev {dev['loss']:.2e}, "
                  f"max tensor dev {max(v for k, v in dev.items() if k != 'loss'):.2e} "
                  f"({max((v, k) for k, v in dev.items() if k != 'loss')[1]})")
        dump[f"loss{t}"] = k_loss
        for k in ("gamma", "beta", "mov_mean", "mov_var"):
            dump[f"p{t}_{k}"] = got[k]
        for i, (w, b) in enumerate(zip(got["W"], got["b"])):
            dump[f"p{t}_W{i}"], dump[f"p{t}_b{i}"] = w, b
    pred_k = model.predict(x.astype(np.float32), verbose=0)
    pred_o = O.predict(ref[np.float64][0], x)
    print(f"model.predict vs oracle.predict (float64): {np.abs(pred_k - pred_o).max():.2e}")
    dump["pred"] = pred_k
    dump["versions"] = np.array([tf.__version__, getattr(tf.keras, "__version__", "?")])
    if a.out:
        np.savez_compressed(a.out, **dump)
    ok = worst < 1e-5 and np.abs(pred_k - pred_o).max() < 1e-4
    print("PINNED: the oracle restates this Keras" if ok else "DEVIATION: the oracle does NOT restate this Keras "
          f"within 1e-5 (worst {worst:.2e}) - report the TensorFlow / Keras versions printed above")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
