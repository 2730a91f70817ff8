"""GPU parity tests: every HIP entry point, called through the C ABI, against the CPU oracle
(oracle/locator_oracle.py, float64) on the same seeded inputs.

Tolerances (all fp32 device arithmetic vs an fp64 oracle; stated per test):
  activations / predictions / losses : 2e-5 absolute on O(1) values
  one Adam step                      : 1e-5 absolute on weights (updates are <= lr = 1e-3), and at most
                                       0.1 % of a tensor's entries off by more than 2e-6: the entries
                                       whose gradient is within fp32 round-off of Adam's eps scale
                                       (|g| ~ 3e-6 at t = 1), where the update is steepest in g
  5 steps                            : 2e-5 absolute on weights, 1e-4 on losses
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import locator_oracle as O
from tests.gpu_util import build_net, make_problem, maxerr, params_err

pytestmark = pytest.mark.gpu


def _rows(idx):
    r = np.zeros(32, np.int32)
    r[:len(idx)] = idx
    return torch.from_numpy(r).cuda()


def _sync():
    torch.cuda.synchronize()


# ------------------------------------------------------------------ layout / utility kernels
def test_w1_swizzle_roundtrip_and_index():
    from locator_amd import _lib
    x, y, p, rng = make_problem(8, 70, 40, 2)
    net = build_net(x, y, p)
    back = net.export_params()
    assert np.array_equal(back["W"][0], p["W"][0].astype(np.float32))
    # raw buffer position of one element agrees with the documented index
    flat = net.params.cpu().numpy()
    for (h, k) in [(0, 0), (39, 69), (17, 33)]:
        assert flat[net.lay.w1 + net.lib.loc_w1s_index(h, k, net.d.Hp)] == np.float32(p["W"][0][k, h])
    # padding is zero
    assert np.count_nonzero(flat[net.lay.w1:net.lay.w1 + net.d.Hp * net.d.Kp]) <= 70 * 40
    for l in range(len(p["W"])):
        assert np.array_equal(back["W"][l], p["W"][l].astype(np.float32)), l
        assert np.array_equal(back["b"][l], p["b"][l].astype(np.float32)), l


def test_glorot_init_is_keyed_by_seed_and_replicate_and_bounded():
    from locator_amd.net import LocatorNet, upload_genotypes
    x, y, p, rng = make_problem(8, 300, 64, 4)
    X = upload_genotypes(x)
    Y = torch.from_numpy(y.astype(np.float32)).cuda()
    a = LocatorNet(X, Y, 300, 64, 4, seed=5, replicate=0).export_params()
    b = LocatorNet(X, Y, 300, 64, 4, seed=5, replicate=0).export_params()
    c = LocatorNet(X, Y, 300, 64, 4, seed=5, replicate=1).export_params()
    dd = LocatorNet(X, Y, 300, 64, 4, seed=6, replicate=0).export_params()
    for l, (fi, fo) in enumerate(O.layer_dims(300, 64, 4)):
        lim = np.sqrt(6.0 / (fi + fo))
        assert np.array_equal(a["W"][l], b["W"][l])
        assert not np.array_equal(a["W"][l], c["W"][l]) and not np.array_equal(a["W"][l], dd["W"][l])
        assert np.abs(a["W"][l]).max() <= lim * (1 + 1e-6)
        if a["W"][l].size > 1000:
            assert np.abs(a["W"][l]).max() > 0.98 * lim
            assert abs(a["W"][l].mean()) < 0.05 * lim
            assert abs(a["W"][l].std() - lim / np.sqrt(3)) < 0.03 * lim
        assert not a["b"][l].any()
    assert np.all(a["gamma"] == 1) and np.all(a["mov_var"] == 1) and not a["beta"].any() and not a["mov_mean"].any()


def test_dropout_mask_rate_and_determinism():
    x, y, p, rng = make_problem(8, 64, 256, 4)
    net = build_net(x, y, p, drop_p=0.25)
    m = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    net.fill_dropout_masks(m, m.numel(), 0)
    a = m.cpu().numpy().copy()
    assert set(np.unique(a)) == {0, 1}
    assert abs(a.mean() - 0.75) < 2e-3
    net.fill_dropout_masks(m, m.numel(), 0)
    assert np.array_equal(a, m.cpu().numpy())
    net.fill_dropout_masks(m, m.numel() // 2, m.numel() // 2)       # offsets address one global stream
    assert np.array_equal(a[m.numel() // 2:], m.cpu().numpy()[:m.numel() // 2])


def test_gather_columns_matches_numpy_fancy_index():
    from locator_amd.net import gather_columns, upload_genotypes
    rng = np.random.default_rng(3)
    x = rng.integers(0, 3, (37, 1000)).astype(np.uint8)
    so = rng.choice(1000, 1000, replace=True)
    X = upload_genotypes(x)
    out = gather_columns(X, so, 1000).cpu().numpy()
    assert np.array_equal(out[:, :1000], x[:, so])
    assert not out[:, 1000:].any()


# ------------------------------------------------------------------ BN + layer 1
@pytest.mark.parametrize("K,n_b", [(200, 32), (5830, 32), (97, 21), (64, 1)])
def test_bn_batch_stats(K, n_b):
    x, y, p, rng = make_problem(50, K, 32, 2, seed=K)
    net = build_net(x, y, p)
    idx = rng.choice(50, n_b, replace=False)
    d, lay = net.d, net.lay
    out4 = torch.zeros(4 * d.Kp, device="cuda")
    P = net.params.data_ptr()
    from locator_amd import _lib
    _lib.check(net.lib.loc_bn_batch_stats(net.X.data_ptr(), net.X.stride(0), _rows(idx).data_ptr(), n_b, d.K, d.Kp,
                                          P + 4 * lay.gamma, P + 4 * lay.beta, P + 4 * lay.mov_mean,
                                          P + 4 * lay.mov_var, out4.data_ptr(), None))
    _sync()
    o = out4.cpu().numpy().reshape(4, d.Kp)
    xb = x[idx].astype(np.float64)
    mu, var = xb.mean(0), xb.var(0)
    rstd = 1 / np.sqrt(var + 1e-3)
    assert maxerr(o[2, :K], mu) < 1e-6 and maxerr(o[3, :K], rstd) < 2e-5
    assert maxerr(o[0, :K], p["gamma"] * rstd) < 5e-5
    assert maxerr(o[1, :K], p["beta"] - mu * p["gamma"] * rstd) < 5e-5
    assert not o[:, K:].any()
    e = net.export_params()
    assert maxerr(e["mov_mean"], 0.99 * p["mov_mean"] + 0.01 * mu) < 1e-6
    assert maxerr(e["mov_var"], 0.99 * p["mov_var"] + 0.01 * var) < 1e-6


@pytest.mark.parametrize("K,width,nlayers,n", [(200, 64, 4, 32), (5830, 256, 10, 45), (97, 100, 3, 7),
                                               (1024, 512, 2, 33), (40, 32, 5, 64)])
def test_predict_matches_oracle_inference(K, width, nlayers, n):
    """loc_predict (BN moving stats -> layer 1 -> hidden stack -> heads) vs oracle.predict; also the
    per-sample distances used for val_loss.  Tolerance 2e-5 abs on O(1) outputs."""
    x, y, p, rng = make_problem(max(n, 8), K, width, nlayers, seed=K + n)
    net = build_net(x, y, p)
    rows = torch.from_numpy(rng.permutation(x.shape[0])[:n].astype(np.int32)).cuda()
    yhat = torch.zeros((n, 2), device="cuda")
    dist = torch.zeros(n, device="cuda")
    net.predict_rows(rows, n, yhat, dist)
    _sync()
    r = rows.cpu().numpy()
    ref = O.predict(p, x[r])
    assert maxerr(yhat.cpu().numpy(), ref) < 2e-5, maxerr(yhat.cpu().numpy(), ref)
    assert maxerr(dist.cpu().numpy(), O.euclid(ref, y[r])) < 2e-5


# ------------------------------------------------------------------ training step
def _one_step_case(K, width, nlayers, n_b, drop_p, seed):
    x, y, p, rng = make_problem(64, K, width, nlayers, seed=seed)
    net = build_net(x, y, p, drop_p=drop_p)
    idx = rng.choice(64, n_b, replace=False)
    Hp = net.d.Hp
    mask_np = (rng.random((32, Hp)) >= drop_p).astype(np.uint8)
    mask = torch.from_numpy(mask_np).cuda() if drop_p > 0 else None
    return x, y, p, net, idx, mask_np, mask


@pytest.mark.parametrize("K,width,nlayers,n_b,drop_p", [
    (200, 64, 4, 32, 0.25), (5830, 256, 10, 32, 0.25), (5830, 256, 10, 21, 0.25), (97, 100, 3, 10, 0.5),
    (333, 32, 2, 32, 0.25), (128, 96, 5, 1, 0.0)])
def test_one_training_step_matches_oracle(K, width, nlayers, n_b, drop_p):
    """BN stats -> forward -> loss -> backward -> Adam (loc_train_step) vs oracle.train_step with the
    same weights, batch rows and dropout mask.  Checks the loss, every updated tensor, both Adam
    moments and the BN moving statistics."""
    x, y, p, net, idx, mask_np, mask = _one_step_case(K, width, nlayers, n_b, drop_p, seed=K + n_b)
    loss = torch.zeros(1, device="cuda")
    net.train_step(_rows(idx), n_b, 1, mask, loss)
    _sync()
    pr = O.copy_params(p)
    m, v = O.zeros_like_trainable(pr), O.zeros_like_trainable(pr)
    ref_loss = O.train_step(pr, m, v, 1, 1e-3, x[idx], y[idx], mask_np[:n_b, :width], drop_p)
    assert abs(loss.item() - ref_loss) < 2e-5 * max(1, abs(ref_loss)), (loss.item(), ref_loss)
    got = net.export_params()
    errs = params_err(got, pr)
    assert max(errs.values()) < 1e-5, errs
    for l in range(len(p["W"])):
        frac = np.mean(np.abs(got["W"][l].astype(np.float64) - pr["W"][l]) > 2e-6)
        assert frac < 1e-3, (l, frac)
    # round-off floor: the same step in the fp32 NumPy oracle is not much closer to fp64 than the HIP path
    p32 = O.cast_params(p, np.float32)
    m32, v32 = O.zeros_like_trainable(p32), O.zeros_like_trainable(p32)
    O.train_step(p32, m32, v32, 1, 1e-3, x[idx], y[idx], mask_np[:n_b, :width], drop_p)
    floor = params_err(p32, pr)
    for k in errs:
        assert errs[k] <= 10 * floor[k] + 3e-6, (k, errs[k], floor[k])
    gm, gv = net.export_adam()
    assert max(params_err(gm, m).values()) < 1e-6
    for l in range(len(p["W"])):
        assert np.allclose(gv["W"][l], v["W"][l], rtol=2e-3, atol=1e-12), l
    # padding stays zero
    flat = net.params.cpu().numpy()
    w1 = flat[net.lay.w1:net.lay.w1 + net.d.Hp * net.d.Kp]
    assert np.count_nonzero(w1) <= K * width


def test_five_steps_with_partial_batch_and_lr_change():
    """Five consecutive steps (Adam t = 1..5, one short batch, LR halved before step 4)."""
    K, width, nlayers = 1000, 128, 6
    x, y, p, rng = make_problem(96, K, width, nlayers, seed=77)
    net = build_net(x, y, p, drop_p=0.25)
    pr = O.copy_params(p)
    m, v = O.zeros_like_trainable(pr), O.zeros_like_trainable(pr)
    losses, refs = [], []
    loss = torch.zeros(5, device="cuda")
    lr = 1e-3
    for s in range(5):
        n_b = 32 if s != 2 else 13
        idx = rng.choice(96, n_b, replace=False)
        mask_np = (rng.random((32, net.d.Hp)) >= 0.25).astype(np.uint8)
        if s == 3:
            lr = 5e-4
            net.lr_t.fill_(lr)
        net.train_step(_rows(idx), n_b, s + 1, torch.from_numpy(mask_np).cuda(), loss[s:])
        refs.append(O.train_step(pr, m, v, s + 1, lr, x[idx], y[idx], mask_np[:n_b, :width], 0.25))
    _sync()
    assert maxerr(loss.cpu().numpy(), refs) < 1e-4
    errs = params_err(net.export_params(), pr)
    assert max(errs.values()) < 2e-5, errs


def test_graph_replay_is_bit_identical_to_eager_and_deterministic():
    """Two fits from the same seed (one eager, one through the captured HIP graph) give bit-identical
    histories and weights: no atomics, fixed reduction orders."""
    from locator_amd.train import fit
    x, y, p, rng = make_problem(120, 700, 64, 4, seed=9)
    tr, va = np.arange(0, 100), np.arange(100, 120)
    outs = []
    for use_graph in (False, True, True):
        net = build_net(x, y, p, drop_p=0.25, seed=3)
        h = fit(net, tr, va, max_epochs=6, patience=100, use_graph=use_graph)
        outs.append((h.history, net.params.cpu().numpy().copy(), net.adam_v.cpu().numpy().copy()))
    for o in outs[1:]:
        assert o[0]["loss"] == outs[0][0]["loss"] and o[0]["val_loss"] == outs[0][0]["val_loss"]
        assert np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])


def test_short_fit_trajectory_matches_oracle_fit():
    """model.fit parity with injected randomness: same init, same per-epoch permutations, dropout masks
    read back from the device.  8 epochs x 4 steps on 100 training rows (last batch of 4 kept).
    Tolerance: 5e-4 absolute on per-epoch loss / val_loss (fp32 drift over 32 Adam steps), 1e-3
    relative on predictions (the north_star's bound)."""
    from locator_amd.train import Callbacks, EpochRunner
    K, width, nlayers = 600, 64, 4
    x, y, p, rng = make_problem(130, K, width, nlayers, seed=21)
    tr, va, pr_rows = np.arange(0, 100), np.arange(100, 120), np.arange(120, 130)
    net = build_net(x, y, p, drop_p=0.25, seed=11)
    runner = EpochRunner(net, tr, va, 32, use_graph=True)
    perms = [np.random.default_rng(100 + e).permutation(100) for e in range(8)]
    masks = []
    hist = {"loss": [], "val_loss": []}
    for e in range(8):
        l, vl = runner.run_epoch(perms[e])
        masks.append(runner.masks.cpu().numpy().reshape(runner.steps, 32, net.d.Hp).copy())
        hist["loss"].append(l)
        hist["val_loss"].append(vl)
    pref = O.copy_params(p)
    href, _ = O.fit(pref, x[tr], y[tr], x[va], y[va], batch_size=32, max_epochs=8, patience=100, drop_p=0.25,
                    perm_fn=lambda e: perms[e], mask_fn=lambda e, s, nb: masks[e][s, :nb, :width])
    assert maxerr(hist["loss"], href["loss"]) < 5e-4, (hist["loss"], href["loss"])
    assert maxerr(hist["val_loss"], href["val_loss"]) < 5e-4
    yhat = torch.zeros((10, 2), device="cuda")
    net.predict_rows(torch.from_numpy(pr_rows.astype(np.int32)).cuda(), 10, yhat)
    _sync()
    ref = O.predict(pref, x[pr_rows])
    rel = np.abs(yhat.cpu().numpy() - ref) / np.maximum(np.abs(ref), 1.0)
    assert rel.max() < 1e-3, rel.max()


def test_fit_callbacks_restore_best_weights_and_history_columns():
    from locator_amd.train import fit
    x, y, p, rng = make_problem(140, 400, 64, 4, seed=4)
    tr, va = np.arange(0, 110), np.arange(110, 140)
    net = build_net(x, y, p, seed=2)
    h = fit(net, tr, va, max_epochs=40, patience=6)
    hh = h.history
    assert list(hh) == ["loss", "val_loss", "learning_rate"]
    ne = len(hh["loss"])
    assert 2 <= ne <= 40 and len(hh["val_loss"]) == ne == len(hh["learning_rate"])
    assert hh["learning_rate"][0] == float(np.float32(1e-3))
    best = int(np.argmin(hh["val_loss"]))
    if ne < 40:
        assert ne - 1 - best == 6          # stopped after `patience` non-improving epochs
    # the reloaded weights reproduce the best epoch's val_loss exactly (same kernels, same weights)
    dist = torch.zeros(30, device="cuda")
    yhat = torch.zeros((30, 2), device="cuda")
    net.predict_rows(torch.from_numpy(va.astype(np.int32)).cuda(), 30, yhat, dist)
    _sync()
    assert abs(float(dist.cpu().numpy().astype(np.float64).mean()) - hh["val_loss"][best]) < 1e-7
    assert hh["loss"][-1] < hh["loss"][0]


def test_bad_arguments_fail_loudly():
    from locator_amd import _lib
    from locator_amd.train import EpochRunner
    x, y, p, rng = make_problem(40, 64, 32, 2)
    net = build_net(x, y, p)
    with pytest.raises(ValueError):
        EpochRunner(net, np.arange(30), np.arange(30, 40), batch_size=64)
    with pytest.raises(_lib.LocatorHipError):
        net.train_step(_rows(np.arange(4)), 0, 1, None, torch.zeros(1, device="cuda"))
    with pytest.raises(_lib.LocatorHipError):           # dropout > 0 without a mask
        net.train_step(_rows(np.arange(4)), 4, 1, None, torch.zeros(1, device="cuda"))


def _callback_run(pipelined, depth=2, use_graph=True, max_epochs=60, patience=5):
    from locator_amd.train import fit
    x, y, p, rng = make_problem(140, 400, 64, 4, seed=4)
    tr, va = np.arange(0, 110), np.arange(110, 140)
    net = build_net(x, y, p, seed=2)
    h = fit(net, tr, va, max_epochs=max_epochs, patience=patience, lr_patience=2, pipelined=pipelined, depth=depth,
            use_graph=use_graph)
    yhat = torch.zeros((140, 2), device="cuda")
    net.predict_rows(torch.arange(140, dtype=torch.int32, device="cuda"), 140, yhat)
    _sync()
    return h, net.params.cpu().numpy().copy(), net.lr_t.item(), yhat.cpu().numpy().copy()


def test_pipelined_fit_is_bit_identical_to_the_synchronous_loop():
    """the callbacks run on the device (loc_epoch_callbacks + loc_snapshot_if in the epoch's graph)
    and the host enqueues epochs ahead of it.  History, restored best weights, final learning rate and predictions of
    fit(pipelined=True) at two depths equal fit(pipelined=False) bit for bit, with and without graph capture; the epochs
    enqueued behind the stop epoch leave no trace (the device state is frozen once early stopping fires)."""
    ref = _callback_run(False)
    ne = len(ref[0].history["loss"])
    assert 7 <= ne < 60, ne                                   # early stopping did fire, and not at once
    assert len(set(ref[0].history["learning_rate"])) >= 2     # ... after at least one LR reduction
    for kw in ({"depth": 2}, {"depth": 4}, {"depth": 1, "use_graph": False}):
        got = _callback_run(True, **kw)
        assert got[0].history == ref[0].history, kw
        assert got[0].best_epoch == ref[0].best_epoch
        assert np.array_equal(got[1], ref[1]) and got[2] == ref[2] and np.array_equal(got[3], ref[3]), kw


def test_device_callbacks_follow_the_keras_state_machines():
    """The device's decisions (checkpoint / stop / LR) replayed through the host restatement of the three Keras callbacks
    (train.Callbacks = SURVEY.md A.5, itself checked against oracle.Callbacks in the CPU suite) on the device's own
    val_loss sequence: same save epochs, same stop epoch, same learning-rate column."""
    from locator_amd.train import Callbacks
    h, _, lr_end, _ = _callback_run(True, max_epochs=80, patience=6)
    hh = h.history
    cb = Callbacks(6, 1e-3, 2, 0.5)
    saves, stop_at, lrs = [], None, []
    for e, v in enumerate(hh["val_loss"]):
        save, stop, lr_logged = cb.on_epoch_end(e, v)
        lrs.append(lr_logged)
        if save:
            saves.append(e)
        if stop:
            stop_at = e
            break
    assert lrs == hh["learning_rate"]
    assert (stop_at is None and len(hh["loss"]) == 80) or stop_at == len(hh["loss"]) - 1
    assert h.best_epoch == saves[-1] == int(np.argmin(hh["val_loss"]))
    assert lr_end == float(np.float32(cb.lr))


def test_pipelined_fit_matches_oracle_fit_with_callbacks():
    """Callback-driven fit (early stopping + LR plateau live) against oracle.fit on the same permutations and the device's
    dropout masks: per-epoch loss / val_loss 5e-4, the same number of epochs and LR column, predictions 1e-3 relative."""
    from locator_amd.train import FitLoop
    K, width, nlayers = 600, 64, 4
    x, y, p, rng = make_problem(130, K, width, nlayers, seed=21)
    tr, va, pr_rows = np.arange(0, 100), np.arange(100, 120), np.arange(120, 130)
    net = build_net(x, y, p, drop_p=0.25, seed=11)
    perms = [np.random.default_rng(100 + e).permutation(100) for e in range(12)]
    loop = FitLoop(net, tr, va, batch_size=32, max_epochs=12, patience=12, perm_fn=lambda e: perms[e], depth=0)   # LR patience int(12 / 6) = 2
    masks = []
    while not loop.done:
        loop.submit()
        loop.collect(0)
        masks.append(loop.runner.masks.cpu().numpy().reshape(loop.runner.steps, 32, net.d.Hp).copy())
    hist = loop.finish().history
    pref = O.copy_params(p)
    href, best = O.fit(pref, x[tr], y[tr], x[va], y[va], batch_size=32, max_epochs=12, patience=12, drop_p=0.25,
                       perm_fn=lambda e: perms[e], mask_fn=lambda e, s, nb: masks[e][s, :nb, :width])
    assert len(hist["loss"]) == len(href["loss"])
    assert maxerr(hist["loss"], href["loss"]) < 5e-4 and maxerr(hist["val_loss"], href["val_loss"]) < 5e-4
    lr_key = "learning_rate" if "learning_rate" in href else "lr"
    assert [float(np.float32(v)) for v in href[lr_key]] == hist["learning_rate"]
    yhat = torch.zeros((10, 2), device="cuda")
    net.predict_rows(torch.from_numpy(pr_rows.astype(np.int32)).cuda(), 10, yhat)
    _sync()
    ref = O.predict(best, x[pr_rows])                 # fit reloads the best-val_loss weights (locator.py:379-388)
    rel = np.abs(yhat.cpu().numpy() - ref) / np.maximum(np.abs(ref), 1.0)
    assert rel.max() < 1e-3, rel.max()


def test_validation_sweep_arithmetic_does_not_depend_on_capture_with_many_validation_rows():
    """with >= 512 validation rows the sweep takes the int8 matrix pipe.  In round 4 an EAGER epoch asked
    the dynamic-range guard (host read-back, maybe two digit planes under --predict_mode auto) while a CAPTURED epoch ran
    three planes unguarded, so val_loss - and every strict-'<' callback decision - depended on --no_graph / the replicate
    layout.  predict_rows(in_fit=True) now pins the sweep: graph and eager fits agree bit for bit, and no guard is asked
    (the fit never synchronises for it)."""
    from locator_amd.train import fit
    x, y, p, rng = make_problem(900, 1500, 64, 4, seed=9)
    tr, va = np.arange(0, 300), np.arange(300, 900)                  # 600 validation rows
    out = []
    for use_graph in (True, False):
        net = build_net(x, y, p, seed=3, predict_digits=0)          # the command line's default: auto
        asked = []
        real = net.quant_guard
        net.quant_guard = lambda: asked.append(1) or real()
        h = fit(net, tr, va, max_epochs=6, patience=6, use_graph=use_graph)
        assert not asked, "the validation sweep consulted the guard"
        out.append((h.history, net.params.cpu().numpy().copy()))
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1])


def test_cross_epoch_chaining_refuses_a_permutation_it_was_not_promised():
    """with xchain an epoch trains on the permutation the PREVIOUS start_epoch uploaded as perm_next.
    A direct EpochRunner user who skips perm_next (run_epoch) used to train every later epoch on the previous permutation
    silently; now it is an error, and the FitLoop-style call sequence still works."""
    from locator_amd.train import EpochRunner
    x, y, p, rng = make_problem(100, 320, 64, 4, seed=5)
    net = build_net(x, y, p, seed=2)
    r = EpochRunner(net, np.arange(80), np.arange(80, 100), use_graph=False, xchain=True)
    if not r.xchain:
        pytest.skip("chained steps unsupported for this shape")
    p0, p1, p2 = (np.random.default_rng(i).permutation(80) for i in range(3))
    r.start_epoch(p0, perm_next=p1)
    r.finish_epoch()
    with pytest.raises(ValueError, match="perm_next"):
        r.start_epoch(p2)                       # p1 was promised
    r.start_epoch(p1, perm_next=p2)
    r.finish_epoch()


def test_two_fit_threads_capture_their_epoch_graphs_50_times_without_disturbing_each_other():
    """fits that share a process (replicates.py: one thread and stream each) capture their epoch
    graphs since round 5; train.DEVICE_LOCK keeps a capture apart from a sibling's set-up, read-back, predict and
    tear-down (round 4: a sibling's device-wide wait made captures fail, a sibling destroying a graph aborted the process
    now and then, so fit threads launched eagerly).  tests/thread_capture_stress.py: two threads x 25 small fits each = 100
    captures (two graph parities per fit) with the sibling in every phase, garbage collection forced between fits; every fit
    must equal the same fit run alone.  Run in a process of its own: the failure this guards against is an abort, which
    must fail this test and not take the session with it."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "thread_capture_stress.py")], capture_output=True, text=True,
                       timeout=600, cwd=os.path.dirname(here))
    assert r.returncode == 0 and "OK 50 fits" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
