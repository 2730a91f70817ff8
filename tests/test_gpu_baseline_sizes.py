"""GPU parity at the sizes BASELINE.json quotes (configs[2], [3], [4]) against the fp64 oracle.

Everything else under tests/ runs at K <= 5,830; the tile/workgroup index arithmetic of the layer-1
kernels (256-workgroup split, per-wave unit ranges, gamma/beta partial slots, the large-M SNP-group
split) only sees its real ranges here:

  configs[2]  synthetic 1000 x 100,000 (locator_amd.synth.synth_genotypes(1000, 100000, 20260101), the
              matrix bench.py times): one step B = 32, the epoch's real last batch B = 10, three consecutive
              steps, predict over all 1000 rows, a 2-epoch captured-graph fit vs oracle.fit;
  configs[3]  one window unit, 765 samples x 150,016 SNPs: one step + the validation sweep;
  configs[4]  K = 500,000: the device bootstrap column gather vs NumPy, then one step on the gathered matrix.

Tolerances are the ones of tests/test_gpu_parity.py (fp32 device arithmetic vs the fp64 oracle): loss 2e-5,
one Adam step 1e-5 absolute on every weight with < 0.1 % of a tensor's entries beyond 2e-6, predictions
2e-5, per-epoch losses of a short fit 5e-4.

Reference lines: model.fit / model.predict, /root/reference/locator/locator.py:367-376, :414, :441; the
window and bootstrap loop bodies :546-571, :648-676.
"""
import numpy as np
import pytest
import torch

from oracle import locator_oracle as O
from tests.gpu_util import build_net, maxerr, params_err, randomize_params

pytestmark = pytest.mark.gpu

WIDTH, NLAYERS, DROP = 256, 10, 0.25


def _rows(idx):
    r = np.zeros(32, np.int32)
    r[:len(idx)] = idx
    return torch.from_numpy(r).cuda()


def _targets(locs):
    from locator_amd.synth import normalize_locs
    return np.nan_to_num(normalize_locs(locs)[4])


def _step_check(net, x, y, pr, m, v, idx, t, mask_np, lr=1e-3, w_tol=1e-5):
    """One loc_train_step vs oracle.train_step on the same rows / mask; returns the two losses."""
    loss = torch.zeros(1, device="cuda")
    net.train_step(_rows(idx), len(idx), t, torch.from_numpy(mask_np).cuda(), loss)
    torch.cuda.synchronize()
    ref = O.train_step(pr, m, v, t, lr, x[idx], y[idx], mask_np[:len(idx), :WIDTH], DROP)
    assert abs(loss.item() - ref) < 2e-5 * max(1.0, abs(ref)), (loss.item(), ref)
    got = net.export_params()
    errs = params_err(got, pr)
    assert max(errs.values()) < w_tol, errs
    return got, errs


@pytest.fixture(scope="module")
def config2():
    from locator_amd.synth import split_indices, synth_genotypes
    x, locs = synth_genotypes(1000, 100_000, seed=20260101, n_na=100)
    train, test, pred = split_indices(locs, 0.9, seed=12345)
    rng = np.random.default_rng(2)
    p = randomize_params(O.init_params(100_000, WIDTH, NLAYERS, rng), rng)
    return x, _targets(locs), p, train, test, pred


@pytest.mark.parametrize("n_b", [32, 10])
def test_config2_one_step_matches_oracle(config2, n_b):
    """K = 100,000, H = 256, L = 10: the metric's minibatch step, full (32 rows) and the epoch's real last batch
    (810 = 25 x 32 + 10).  Every tensor, both Adam moments of W1, BN moving statistics."""
    x, y, p, train, test, pred = config2
    rng = np.random.default_rng(100 + n_b)
    net = build_net(x, y, p, drop_p=DROP)
    assert net.l1_bwd_grid == 512 and net.d.Kp == 100_000
    idx = rng.choice(train, n_b, replace=False)
    mask_np = (rng.random((32, WIDTH)) >= DROP).astype(np.uint8)
    pr = O.copy_params(p)
    m, v = O.zeros_like_trainable(pr), O.zeros_like_trainable(pr)
    got, errs = _step_check(net, x, y, pr, m, v, idx, 1, mask_np)
    frac = np.mean(np.abs(got["W"][0].astype(np.float64) - pr["W"][0]) > 2e-6)
    assert frac < 1e-3, frac
    gm, gv = net.export_adam()
    assert maxerr(gm["W"][0], m["W"][0]) < 1e-6 and maxerr(gm["gamma"], m["gamma"]) < 1e-6
    assert np.allclose(gv["W"][0], v["W"][0], rtol=2e-3, atol=1e-12)
    assert maxerr(got["mov_mean"], pr["mov_mean"]) < 1e-6 and maxerr(got["mov_var"], pr["mov_var"]) < 1e-6


def test_config2_three_consecutive_steps(config2):
    """Adam t = 1..3 with the moments carried over (32, 32 and 10 rows)."""
    x, y, p, train, test, pred = config2
    rng = np.random.default_rng(7)
    net = build_net(x, y, p, drop_p=DROP)
    pr = O.copy_params(p)
    m, v = O.zeros_like_trainable(pr), O.zeros_like_trainable(pr)
    for t, n_b in enumerate((32, 32, 10), start=1):
        idx = rng.choice(train, n_b, replace=False)
        mask_np = (rng.random((32, WIDTH)) >= DROP).astype(np.uint8)
        _step_check(net, x, y, pr, m, v, idx, t, mask_np, w_tol=2e-5)


PREDICT_MODES = {   # name: (LocatorNet settings, absolute bar on z-scored predictions, rows from which an image is built)
    "int8x3": ({"predict_digits": 3}, 2e-5, 512),
    "int8x2": ({"predict_digits": 2}, None, 512),
    "bf16x3": ({"predict_digits": -1, "predict_pieces": 3}, 2e-5, 1152),
    "bf16x2": ({"predict_digits": -1, "predict_pieces": 2}, None, 768),
    "bf16x1": ({"predict_digits": -1, "predict_pieces": 1}, None, 640),
}
# north_star: "predlocs within 1e-3 relative of the Keras reference".  Every many-row mode the bench quotes against the
# MFMA target must meet it here, at the metric's K; bf16 x 1 does NOT (it is held to 2e-2 and is not quoted).
NORTH_STAR_REL = 1e-3


@pytest.mark.parametrize("mode,reps", [("int8x3", 1), ("int8x3", 5), ("int8x2", 1), ("int8x2", 5), ("bf16x3", 1),
                                       ("bf16x3", 3), ("bf16x2", 1), ("bf16x1", 1)])
def test_config2_predict_all_rows(config2, mode, reps):
    """loc_predict over all 1000 rows, and over 3000 / 5000 (every row several times; more than one LOC_PREDICT_CHUNK is
    covered at small K in tests/test_gpu_gemm_i8.py), vs oracle.predict, plus the per-row validation
    distances.  The deviation of the predictions relative to the largest prediction is printed per mode and asserted
    against the north_star bound for every mode except plain bf16 weights."""
    x, y, p, train, test, pred = config2
    kw, tol_abs, min_rows = PREDICT_MODES[mode]
    net = build_net(x, y, p, drop_p=DROP, **kw)
    n = x.shape[0] * reps
    rows = torch.from_numpy((np.random.default_rng(3).permutation(n) % x.shape[0]).astype(np.int32)).cuda()
    yhat, dist = torch.zeros((n, 2), device="cuda"), torch.zeros(n, device="cuda")
    net.predict_rows(rows, n, yhat, dist)
    torch.cuda.synchronize()
    assert (net.l1_image is not None) == (n >= min_rows)
    r = rows.cpu().numpy()
    ref = O.predict(p, x[r], batch=250)
    err = maxerr(yhat.cpu().numpy(), ref)
    rel = err / float(np.abs(ref).max())
    print(f"predict mode {mode}, {n} rows: max |dev| {err:.3e}, relative to max |pred| {rel:.3e}")
    if mode == "bf16x1":
        assert err < 2e-2 and rel > 1e-4, (err, rel)            # outside the north_star tolerance: never quoted
    else:
        assert rel <= NORTH_STAR_REL, (mode, err, rel)
    if tol_abs is not None:
        assert err < tol_abs, err
        assert maxerr(dist.cpu().numpy(), O.euclid(ref, y[r])) < tol_abs


def test_config2_two_epoch_graph_fit_matches_oracle_fit(config2):
    """Two epochs of model.fit on the 810 / 90 split (26 steps each, last batch of 10, validation sweep),
    epoch 0 eager and epoch 1 from the captured HIP graph, vs oracle.fit with the same permutations and the
    device's dropout masks."""
    from locator_amd.train import EpochRunner
    x, y, p, train, test, pred = config2
    net = build_net(x, y, p, drop_p=DROP, seed=17)
    runner = EpochRunner(net, train, test, 32, use_graph=True)
    assert runner.steps == 26 and runner.step_sizes[-1] == 10
    perms = [np.random.default_rng(50 + e).permutation(len(train)) for e in range(2)]
    masks, hist = [], {"loss": [], "val_loss": []}
    for e in range(2):
        l, vl = runner.run_epoch(perms[e])
        masks.append(runner.masks.cpu().numpy().reshape(runner.steps, 32, net.d.Hp).copy())
        hist["loss"].append(l)
        hist["val_loss"].append(vl)
    assert runner.graph is not None
    pref = O.copy_params(p)
    href, _ = O.fit(pref, x[train], y[train], x[test], y[test], batch_size=32, max_epochs=2, patience=100,
                    drop_p=DROP, perm_fn=lambda e: perms[e], mask_fn=lambda e, s, nb: masks[e][s, :nb, :WIDTH])
    assert maxerr(hist["loss"], href["loss"]) < 5e-4, (hist["loss"], href["loss"])
    assert maxerr(hist["val_loss"], href["val_loss"]) < 5e-4, (hist["val_loss"], href["val_loss"])
    got = net.export_params()
    errs = params_err(got, pref)
    # 52 Adam steps at lr 1e-3: an entry whose gradient sits at Adam's eps scale moves by a fraction of lr per step in
    # a direction that fp32 round-off decides, so the bar on the worst of 25.6 M weights is set by the fp32 NumPy
    # oracle's own distance from the fp64 one (same permutations and masks), and the bulk must sit far below it
    p32 = O.cast_params(p, np.float32)
    O.fit(p32, x[train], y[train].astype(np.float32), x[test], y[test].astype(np.float32), batch_size=32,
          max_epochs=2, patience=100, drop_p=DROP, perm_fn=lambda e: perms[e],
          mask_fn=lambda e, s, nb: masks[e][s, :nb, :WIDTH])
    floor = params_err(p32, pref)
    for k in errs:
        assert errs[k] <= 4 * floor[k] + 1e-5, (k, errs[k], floor[k])
    assert max(errs.values()) < 5e-4, errs
    assert np.mean(np.abs(got["W"][0].astype(np.float64) - pref["W"][0]) > 2e-5) < 1e-3
    yhat = torch.zeros((len(pred), 2), device="cuda")
    net.predict_rows(torch.from_numpy(pred.astype(np.int32)).cuda(), len(pred), yhat)
    torch.cuda.synchronize()
    yhat = yhat.cpu().numpy()
    rel = lambda a, b: float((np.abs(a - b) / np.maximum(np.abs(b), 1.0)).max())
    # (a) the forward itself is exact: the oracle's predict on the weights the device ended with
    assert rel(yhat, O.predict(O.cast_params(got, np.float64), x[pred])) < 2e-5
    # (b) after 52 fp32 Adam steps in the steep part of the fit, round-off in the WEIGHTS moves the predictions of any
    # fp32 implementation by a few 1e-3 (measured: fp32 NumPy oracle vs fp64 oracle 6.0e-3, HIP vs fp64 3.3e-3), so
    # the north_star's 1e-3 bound on predlocs is only testable where fp32 drift is below it (the short fits of
    # tests/test_gpu_parity.py); here the bar is the fp32 oracle's own distance from the fp64 one
    ref = O.predict(pref, x[pred])
    floor_pred = rel(O.predict(p32, x[pred]), ref)
    assert rel(yhat, ref) < 1.5 * floor_pred + 1e-4, (rel(yhat, ref), floor_pred)
    assert rel(yhat, ref) < 1e-2


def test_config3_window_unit_step_and_validation():
    """One --windows unit at Ag1000G scale: 765 samples x 150,016 SNPs (a 2 Mb window), 688 / 77 split."""
    from locator_amd.synth import synth_genotypes
    n, K = 765, 150_016
    x, locs = synth_genotypes(n, K, seed=3, n_na=0)
    y = _targets(locs)
    rng = np.random.default_rng(33)
    p = randomize_params(O.init_params(K, WIDTH, NLAYERS, rng), rng)
    net = build_net(x, y, p, drop_p=DROP)
    val = rng.choice(n, 77, replace=False)
    train = np.setdiff1d(np.arange(n), val)
    idx = rng.choice(train, 32, replace=False)
    mask_np = (rng.random((32, WIDTH)) >= DROP).astype(np.uint8)
    pr = O.copy_params(p)
    m, v = O.zeros_like_trainable(pr), O.zeros_like_trainable(pr)
    _step_check(net, x, y, pr, m, v, idx, 1, mask_np)
    yhat, dist = torch.zeros((77, 2), device="cuda"), torch.zeros(77, device="cuda")
    net.predict_rows(torch.from_numpy(val.astype(np.int32)).cuda(), 77, yhat, dist)
    torch.cuda.synchronize()
    ref = O.predict(pr, x[val])
    assert maxerr(yhat.cpu().numpy(), ref) < 2e-5
    assert maxerr(dist.cpu().numpy(), O.euclid(ref, y[val])) < 2e-5


def test_config4_bootstrap_gather_and_step_at_500k_snps():
    """K = 500,000 (configs[4]'s matrix width; 160 rows are enough for a 32-row step): the bootstrap replicate's
    column resample on the device is bit-identical to NumPy's fancy index (locator.py:648-653), and one
    training step on the resampled matrix matches the oracle on the same matrix."""
    from locator_amd.net import gather_columns, upload_genotypes
    n, K = 160, 500_000
    rng = np.random.default_rng(44)
    af = rng.beta(0.4, 0.9, K).clip(0.02, 0.98)
    x0 = (rng.random((n, K), dtype=np.float32) < af).astype(np.uint8)
    x0 += (rng.random((n, K), dtype=np.float32) < af).astype(np.uint8)
    site_order = np.random.RandomState(4).choice(K, K, replace=True)
    xg = gather_columns(upload_genotypes(x0), site_order, K).cpu().numpy()
    x = x0[:, site_order]
    assert np.array_equal(xg[:, :K], x) and not xg[:, K:].any()
    del xg, x0
    y = rng.normal(0, 1, (n, 2))
    p = randomize_params(O.init_params(K, WIDTH, NLAYERS, rng), rng)
    net = build_net(x, y, p, drop_p=DROP)
    idx = rng.choice(n, 32, replace=False)
    mask_np = (rng.random((32, WIDTH)) >= DROP).astype(np.uint8)
    m, v = O.zeros_like_trainable(p), O.zeros_like_trainable(p)
    _step_check(net, x, y, p, m, v, idx, 1, mask_np)      # p is updated in place: it is not needed afterwards


def test_config4_many_row_gemm_at_500k_snps():
    """The image + GEMM path of a many-row predict at configs[4]'s width (K = 500,000: 7,813 SNP blocks padded to
    7,814, 3,907 pairs over 32 groups, 0.77 GB of three-piece weight image): 1,200 rows drawn with repeats from a
    160-row matrix, checked against the fp64 forward on 96 of them (first, last and random rows)."""
    from tests.test_gpu_gemm import _a1_reference, run_gemm
    n_mat, K, n = 160, 500_000, 1200
    rng = np.random.default_rng(45)
    af = rng.beta(0.4, 0.9, K).clip(0.02, 0.98)
    x = (rng.random((n_mat, K), dtype=np.float32) < af).astype(np.uint8)
    x += (rng.random((n_mat, K), dtype=np.float32) < af).astype(np.uint8)
    y = rng.normal(0, 1, (n_mat, 2))
    p = randomize_params(O.init_params(K, WIDTH, NLAYERS, rng), rng)
    net = build_net(x, y, p, drop_p=DROP)
    rows = rng.integers(0, n_mat, n).astype(np.int32)
    a1 = run_gemm(net, torch.from_numpy(rows).cuda(), n, 3)
    check = np.unique(np.concatenate([[0, 1, 127, 128, n - 1], rng.choice(n, 91, replace=False)]))
    ref, z = _a1_reference(p, x[rows[check]])
    err = np.abs(a1[check, :WIDTH] - ref).max()
    assert err < 5e-5, err                                   # exact products, fp32 accumulation over 500k terms
    # the int8 form of the same sweep (the default): 32 groups of 122-123 pairs, digit planes of 0.39 GB; the i32
    # accumulators hold at most 2 * 128 * 15,744 per group
    from tests.test_gpu_gemm_i8 import run_gemm_i8
    for digits, bar in ((3, 5e-5), (2, 5e-3)):
        a8 = run_gemm_i8(net, torch.from_numpy(rows).cuda(), n, digits)
        err8 = np.abs(a8[check, :WIDTH] - ref).max()
        assert err8 < bar, (digits, err8)
    # ... and through loc_predict (image kept for the second call), against oracle.predict
    r = torch.from_numpy(rows[:700].copy()).cuda()
    yhat = torch.zeros((700, 2), device="cuda")
    net.predict_rows(r, 700, yhat)
    net.predict_rows(r, 700, yhat)
    torch.cuda.synchronize()
    assert net._image_mode == 13 and net._net.l1_image_ready == 13
    sel = check[check < 700]
    assert maxerr(yhat.cpu().numpy()[sel], O.predict(p, x[rows[sel]])) < 5e-5

