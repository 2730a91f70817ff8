"""Pins the CPU oracle (oracle/locator_oracle.py) with everything that can be
pinned without TensorFlow: NumPy known answers, an independent
torch-autograd float64 derivation of every gradient, hand-computed Adam / BN
values, and scripted callback traces."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import locator_oracle as O


# ---------------------------------------------------------------- NumPy KATs
@pytest.mark.parametrize("seed,first10,sha,reseed,so8", [
    (12345, [465, 459, 233, 149, 429, 423, 454, 140, 489, 165], "144d3567059195b1", 812135,
     [1092, 5266, 5679, 4778, 3551, 2418, 1321, 5729]),
    (54321, [334, 260, 165, 112, 292, 283, 364, 87, 65, 82], "94c04ad6e4a776db", 429793,
     [367, 2699, 4649, 5250, 3196, 5447, 143, 2576]),
])
def test_numpy_rng_chain_known_answers(seed, first10, sha, reseed, so8):
    # fixture shape: 500 samples, first 50 NA, K=5830 after filters
    locs = np.random.default_rng(1).uniform(0, 50, (500, 2))
    locs[:50] = np.nan
    ac = np.zeros((5830, 500), np.int8)
    np.random.seed(seed)
    train, test, tg, vg, tl, vl, pred, pg = O.split_train_test(ac, locs, 0.9)
    assert list(test[:10]) == first10
    assert hashlib.sha1(test.astype("int64").tobytes()).hexdigest()[:16] == sha
    assert len(train) == 405 and len(test) == 45 and len(pred) == 50
    assert tg.shape == (405, 5830) and vg.shape == (45, 5830) and pg.shape == (50, 5830)
    assert np.all(np.diff(train) > 0)
    chain = O.bootstrap_chain(2, 5830)
    assert chain[0][0] == reseed
    assert list(chain[0][1][:8]) == so8


def test_normalize_locs():
    locs = np.array([[np.nan, np.nan], [1.0, 10.0], [3.0, 30.0], [5.0, 20.0]])
    ml, sl, ma, sa, out = O.normalize_locs(locs)
    assert ml == 3.0 and ma == 20.0
    assert np.isclose(sl, np.std([1, 3, 5])) and np.isclose(sa, np.std([10, 30, 20]))
    assert np.isnan(out[0]).all()
    assert np.allclose(out[1:, 0], (np.array([1, 3, 5]) - 3) / sl)


# ---------------------------------------------------------------- torch cross-check
def _torch_loss(p, x, y, mask, drop_p):
    """Independent derivation: torch ops + autograd, float64, Keras semantics by hand
    (nn.BatchNorm1d / torch Adam are NOT equivalent, so not used)."""
    tp = {k: ([torch.tensor(a, requires_grad=True) for a in v] if isinstance(v, list)
              else torch.tensor(v, requires_grad=k in ("gamma", "beta"))) for k, v in p.items()}
    xt = torch.tensor(x.astype(np.float64))
    mu = xt.mean(0)
    var = ((xt - mu) ** 2).mean(0)
    a = (xt - mu) / torch.sqrt(var + O.BN_EPS) * tp["gamma"] + tp["beta"]
    nl = len(p["W"]) - 2
    if O.n_pre(nl) == 0 and drop_p > 0:           # --nlayers 1: Dropout directly after the BatchNormalization
        a = a * torch.tensor(mask.astype(np.float64)) / (1 - drop_p)
    for l in range(nl):
        a = torch.nn.functional.elu(a @ tp["W"][l] + tp["b"][l])
        if l == O.n_pre(nl) - 1 and drop_p > 0:
            a = a * torch.tensor(mask.astype(np.float64)) / (1 - drop_p)
    y1 = a @ tp["W"][nl] + tp["b"][nl]
    y2 = y1 @ tp["W"][nl + 1] + tp["b"][nl + 1]
    loss = torch.sqrt(((y2 - torch.tensor(y)) ** 2).sum(-1)).mean()
    loss.backward()
    return loss.item(), tp, y2.detach().numpy()


@pytest.mark.parametrize("n,K,width,nlayers,drop_p", [
    (16, 64, 32, 10, 0.25), (7, 40, 16, 4, 0.5), (32, 128, 64, 3, 0.0), (5, 33, 8, 2, 0.25), (9, 48, 16, 1, 0.25),
    (6, 20, 8, 1, 0.0)])
def test_gradients_match_torch_autograd(n, K, width, nlayers, drop_p):
    rng = np.random.default_rng(7)
    p = O.init_params(K, width, nlayers, rng)
    p["gamma"] = rng.uniform(0.5, 1.5, K)
    p["beta"] = rng.normal(0, 0.1, K)
    for l in range(len(p["b"])):
        p["b"][l] = rng.normal(0, 0.1, p["b"][l].shape)
    x = rng.integers(0, 3, (n, K)).astype(np.uint8)
    y = rng.normal(0, 1, (n, 2))
    mask = (rng.random((n, width if O.n_pre(nlayers) > 0 else K)) >= drop_p).astype(np.uint8)
    loss, g, yhat = O.loss_and_grads(O.copy_params(p), x, y, mask, drop_p)
    tl, tp, ty = _torch_loss(p, x, y, mask, drop_p)
    assert abs(loss - tl) < 1e-12
    assert np.allclose(yhat, ty, atol=1e-12)
    assert np.allclose(g["gamma"], tp["gamma"].grad.numpy(), atol=1e-12)
    assert np.allclose(g["beta"], tp["beta"].grad.numpy(), atol=1e-12)
    for l in range(len(p["W"])):
        assert np.allclose(g["W"][l], tp["W"][l].grad.numpy(), atol=1e-12), l
        assert np.allclose(g["b"][l], tp["b"][l].grad.numpy(), atol=1e-12), l


def test_bn_training_uses_biased_variance_and_updates_moving_stats():
    p = O.init_params(3, 4, 2, np.random.default_rng(0))
    x = np.array([[0, 1, 2], [2, 1, 0], [2, 2, 0], [0, 0, 0]], np.uint8)
    _, c = O.forward(p, x, True, np.ones((4, 4)), 0.25)
    mu = x.astype(float).mean(0)
    var = x.astype(float).var(0)      # ddof=0
    assert np.allclose(c["mu"], mu) and np.allclose(c["var"], var)
    assert np.allclose(p["mov_mean"], 0.01 * mu)
    assert np.allclose(p["mov_var"], 0.99 + 0.01 * var)
    # inference uses moving stats and ignores dropout
    y_inf, _ = O.forward(p, x, False)
    xh = (x - p["mov_mean"]) / np.sqrt(p["mov_var"] + 1e-3)
    a = xh
    for l in range(2):
        a = O.elu(a @ p["W"][l] + p["b"][l])
    ref = (a @ p["W"][2] + p["b"][2]) @ p["W"][3] + p["b"][3]
    assert np.allclose(y_inf, ref)


def test_adam_keras_form_known_value():
    # one step, g=3e-6, eps=1e-7 -> w = 0.9995132 (Keras form), not torch's 0.9990323
    p = {"gamma": np.array([1.0]), "beta": np.array([0.0]), "W": [], "b": []}
    g = {"gamma": np.array([3e-6]), "beta": np.array([0.0]), "W": [], "b": []}
    m = {"gamma": np.zeros(1), "beta": np.zeros(1), "W": [], "b": []}
    v = {"gamma": np.zeros(1), "beta": np.zeros(1), "W": [], "b": []}
    O.adam_apply(p, g, m, v, 1, 1e-3)
    assert abs(p["gamma"][0] - 0.9995132) < 1e-6
    assert np.isclose(m["gamma"][0], 3e-7) and np.isclose(v["gamma"][0], 9e-15)
    assert p["beta"][0] == 0.0


def test_loss_zero_distance_has_zero_gradient():
    rng = np.random.default_rng(3)
    p = O.init_params(8, 4, 2, rng)
    x = rng.integers(0, 3, (4, 8))
    yhat, _ = O.forward(O.copy_params(p), x, True, np.ones((4, 4)), 0.0)
    y = yhat.copy()
    y[1:] += 1.0                       # sample 0 sits exactly on its target
    loss, g, _ = O.loss_and_grads(p, x, y, np.ones((4, 4)), 0.0)
    assert np.isfinite(loss) and all(np.isfinite(a).all() for a in g["W"])


# ---------------------------------------------------------------- callbacks
def test_callbacks_state_machines():
    cb = O.Callbacks(patience=12)        # lr patience int(12/6)=2
    vals = [1.0, 0.9, 0.95, 0.96, 0.97, 0.9, 0.8] + [0.85] * 12
    saves, stops, lrs = [], [], []
    for e, v in enumerate(vals):
        s, st, lr = cb.on_epoch_end(e, v)
        saves.append(s); stops.append(st); lrs.append(lr)
    assert saves[:7] == [True, True, False, False, False, False, True]   # tie at 0.9 does not save
    assert not any(saves[7:])
    # LR: epochs 2,3 non-improving -> halve after epoch 3; epoch 4,5 (0.9 is a tie => not improving) -> halve after 5
    assert np.allclose(lrs[:7], [1e-3, 1e-3, 1e-3, 1e-3, 5e-4, 5e-4, 2.5e-4], rtol=1e-6)
    assert lrs[0] == float(np.float32(1e-3))
    # early stop: best at epoch 6, 12 non-improving epochs -> stop at epoch 18
    assert stops.index(True) == 18 and len(vals) == 19


def test_fit_history_and_best_weights():
    rng = np.random.default_rng(11)
    K, n = 48, 40
    x = rng.integers(0, 3, (n, K)).astype(np.uint8)
    w = rng.normal(0, 1, (K, 2))
    y = (x - x.mean(0)) @ w
    y = (y - y.mean(0)) / y.std(0)
    p = O.init_params(K, 16, 4, rng)
    hist, best = O.fit(p, x[:32], y[:32], x[32:], y[32:], batch_size=10, max_epochs=30, patience=6)
    ne = len(hist["loss"])
    assert ne == len(hist["val_loss"]) == len(hist["learning_rate"]) and 2 <= ne <= 30
    assert hist["loss"][-1] < hist["loss"][0]
    # best weights reproduce the minimum val_loss
    val = O.euclid(O.predict(best, x[32:]), y[32:]).mean()
    assert np.isclose(val, min(hist["val_loss"]))
    # partial last batch kept: 32 rows / 10 -> 4 steps/epoch (checked through Adam's t via determinism)
    p2 = O.init_params(K, 16, 4, np.random.default_rng(11))


def test_fp32_oracle_tracks_fp64():
    rng = np.random.default_rng(5)
    K, n, width = 256, 32, 32
    p = O.init_params(K, width, 4, rng)
    x = rng.integers(0, 3, (n, K)).astype(np.uint8)
    y = rng.normal(0, 1, (n, 2))
    mask = (rng.random((n, width)) >= 0.25)
    l64, g64, _ = O.loss_and_grads(O.copy_params(p), x, y, mask)
    l32, g32, _ = O.loss_and_grads(O.cast_params(p, np.float32), x, y, mask)
    assert abs(l64 - l32) < 1e-5
    assert np.allclose(g64["W"][0], g32["W"][0], atol=2e-6)


def test_torch_cpu_restatement_matches_the_numpy_oracle():
    """oracle/torch_cpu.py (the multi-threaded torch-CPU fp32 restatement timed as bench.py's cpu_baseline) against
    the fp64 NumPy oracle: 2 epochs with the same permutations and dropout masks, a partial last batch, and the
    validation sweep.  fp32 vs fp64: losses 1e-4, weights 5e-5."""
    import torch
    from oracle.torch_cpu import TorchCpuLocator
    rng = np.random.default_rng(5)
    n, K, H, L = 75, 300, 32, 4
    x = rng.integers(0, 3, (n, K)).astype(np.uint8)
    y = rng.normal(0, 1, (n, 2))
    p = O.init_params(K, H, L, rng)
    p["gamma"] = rng.uniform(0.7, 1.3, K)
    p["mov_mean"] = rng.uniform(0, 1, K)
    tr, va = np.arange(60), np.arange(60, 75)
    perms = [rng.permutation(60) for _ in range(2)]
    masks = [[(rng.random((32, H)) >= 0.25) for _ in range(2)] for _ in range(2)]
    pref = O.copy_params(p)
    href, _ = O.fit(pref, x[tr], y[tr], x[va], y[va], batch_size=32, max_epochs=2, patience=100, drop_p=0.25,
                    perm_fn=lambda e: perms[e], mask_fn=lambda e, s, nb: masks[e][s][:nb])
    torch.set_num_threads(2)
    xt, xv = torch.from_numpy(x[tr]), torch.from_numpy(x[va])
    yt, yv = torch.from_numpy(y[tr].astype(np.float32)), torch.from_numpy(y[va].astype(np.float32))
    for fused in (True, False):           # single-pass torch._fused_adam_ with the Keras-equivalent eps, and the foreach form
        net = TorchCpuLocator(O.cast_params(p, np.float32), 0.25, fused=fused)
        for e in range(2):
            loss, val = net.fit_epoch(xt, yt, xv, yv, perms[e],
                                      [torch.from_numpy(m.astype(np.float32)) for m in masks[e]], 1e-3)
            assert abs(loss - href["loss"][e]) < 1e-4 and abs(val - href["val_loss"][e]) < 1e-4
        got = net.export()
        for k in ("gamma", "beta", "mov_mean", "mov_var"):
            assert np.abs(got[k] - pref[k]).max() < 5e-5, (fused, k)
        for l in range(len(pref["W"])):
            assert np.abs(got["W"][l] - pref["W"][l]).max() < 5e-5, (fused, l)
            assert np.abs(got["b"][l] - pref["b"][l]).max() < 5e-5, (fused, l)
