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
stdout and, with --out, the inputs, the injected starting weights and every Keras-side tensor go to an .npz.  Commit
that file as tests/golden/keras_<anything>.npz: tests/test_golden.py then replays it through the oracle (CPU suite) and
through the HIP path (GPU suite) at this script's bars, and DESIGN.md §2's "parity unpinned" is lifted by a file, not
by a code change.

    python tests/keras_crosscheck.py --out tests/golden/keras_vectors.npz  [--snps 300] [--width 32] [--nlayers 4] [--steps 5]
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


def make_problem(snps, width, nlayers, steps):
    """The seeded inputs of the cross-check: genotypes, targets, the injected starting weights, explicit batches with
    one partial batch in the middle."""
    from oracle import locator_oracle as O
    rng = np.random.default_rng(7)
    n = 96
    x = rng.integers(0, 3, (n, snps)).astype(np.uint8)
    y = rng.normal(0, 1, (n, 2)).astype(np.float32)
    p0 = O.init_params(snps, width, nlayers, rng, dtype=np.float32)
    p0["gamma"] = rng.uniform(0.7, 1.3, snps).astype(np.float32)
    p0["beta"] = rng.normal(0, 0.05, snps).astype(np.float32)
    p0["mov_mean"] = rng.uniform(0, 1, snps).astype(np.float32)
    p0["mov_var"] = rng.uniform(0.2, 1.2, snps).astype(np.float32)
    batches = [rng.choice(n, 32 if s != steps // 2 else 13, replace=False) for s in range(steps)]
    return x, y, p0, batches


def collect(x, y, p0, batches, train_on_batch, extract_fn, predict_fn, versions, nlayers):
    """Run `train_on_batch(rows) -> loss` over the batches and gather everything tests/test_golden.py replays:
    inputs, the starting weights, per-step losses and tensors, the final predictions, and who produced them."""
    dump = {"x": x, "y": y, "nlayers": np.int64(nlayers), "steps": np.int64(len(batches)),
            "batches": np.array([np.pad(b, (0, 32 - len(b)), constant_values=-1) for b in batches])}

    def put(prefix, p):
        for k in ("gamma", "beta", "mov_mean", "mov_var"):
            dump[f"{prefix}_{k}"] = np.asarray(p[k])
        for i, (w, b) in enumerate(zip(p["W"], p["b"])):
            dump[f"{prefix}_W{i}"], dump[f"{prefix}_b{i}"] = np.asarray(w), np.asarray(b)
    put("p0", p0)
    steps = []
    for t, rows in enumerate(batches, start=1):
        loss = float(train_on_batch(rows))
        got = extract_fn()
        dump[f"loss{t}"] = loss
        put(f"p{t}", got)
        steps.append((t, rows, loss, got))
    dump["pred"] = np.asarray(predict_fn())
    dump["versions"] = np.array(versions)
    return dump, steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=300)
    ap.add_argument("--width", type=int, default=32)
    ap.add_argument("--nlayers", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=None, help="e.g. tests/golden/keras_vectors.npz: tests/test_golden.py replays every "
                                                "tests/golden/keras_*.npz through the oracle and the HIP path")
    a = ap.parse_args()
    if importlib.util.find_spec("tensorflow") is None:
        print("keras_crosscheck: TensorFlow is not installed here; nothing compared.  Run this script wherever the "
              "reference runs (pip install tensorflow numpy) and keep the printed report / --out file.")
        return 3
    import tensorflow as tf

    from oracle import locator_oracle as O
    x, y, p0, batches = make_problem(a.snps, a.width, a.nlayers, a.steps)
    model = build_keras_model(tf, a.snps, a.width, a.nlayers, 0.0)
    inject(model, p0)
    dump, steps = collect(x, y, p0, batches,
                          lambda rows: model.train_on_batch(x[rows].astype(np.float32), y[rows]),
                          lambda: extract(model), lambda: model.predict(x.astype(np.float32), verbose=0),
                          [tf.__version__, getattr(tf.keras, "__version__", "?")], a.nlayers)
    ref = {dt: O.cast_params(p0, dt) for dt in (np.float32, np.float64)}
    ref = {dt: (p, O.zeros_like_trainable(p), O.zeros_like_trainable(p)) for dt, p in ref.items()}
    worst = 0.0
    for t, rows, k_loss, got in steps:
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
    pred_k = dump["pred"]
    pred_o = O.predict(ref[np.float64][0], x)
    print(f"model.predict vs oracle.predict (float64): {np.abs(pred_k - pred_o).max():.2e}")
    if a.out:
        np.savez_compressed(a.out, **dump)
    ok = worst < 1e-5 and np.abs(pred_k - pred_o).max() < 1e-4
    print("PINNED: the oracle restates this Keras" if ok else "DEVIATION: the oracle does NOT restate this Keras "
          f"within 1e-5 (worst {worst:.2e}) - report the TensorFlow / Keras versions printed above")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
