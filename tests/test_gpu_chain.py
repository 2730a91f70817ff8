"""Chained training steps (locator_amd/csrc/l1_chain.hip): the layer-1 backward of minibatch t also produces the
layer-1 forward of minibatch t + 1 from the weights while they are in registers (consecutive steps of model.fit,
/root/reference/locator/locator.py:367-376).  The chained epoch must equal the unchained one (same kernels for
everything else) up to the summation order of two reductions, and the oracle like every other path."""
import numpy as np
import pytest
import torch

from oracle import locator_oracle as O
from tests.gpu_util import build_net, make_problem, maxerr, params_err

pytestmark = pytest.mark.gpu


def _sync():
    torch.cuda.synchronize()


def _assert_same_fit(p0, p1, m0, m1, v0, v1):
    """Two fp32 evaluations of the same steps (chained / unchained schedule).  They differ in summation order (the
    gamma / beta gradient: 8 per-wave partials against 2; the layer-1 partial sums) and in how dW1 and the gamma / beta
    gradients are associated (the chained kernel derives them from sum_b dZ xn, l1_chain.hip), i.e. by fp32 round-off in
    the gradients.  An entry whose gradient sits at Adam's eps scale moves by a fraction of lr = 1e-3 per step in a
    direction round-off decides, so: the worst entry within 5 % of one Adam step, all but 0.1 % of the entries of the big
    tensors within 2e-6, first moments 1e-5 (2e-6 for W1), second moments rtol 2e-3.  Measured against the fp64 oracle at
    100,000 SNPs after 10 steps (tests/chain_vs_oracle.py): gamma max error 2.4e-5 unchained, 2.2e-6 chained -- the
    chained kernel's association is the more accurate of the two."""
    err = params_err(p0, p1)
    assert max(err.values()) < 5e-5, err
    for a, b in ((p0["W"][0], p1["W"][0]), (p0["gamma"], p1["gamma"]), (p0["beta"], p1["beta"])):
        assert np.mean(np.abs(a.astype(np.float64) - b) > 2e-6) < 1e-3
    for l in range(1, len(p0["W"])):
        assert maxerr(p0["W"][l], p1["W"][l]) < 5e-6, (l, err)
    merr = params_err(m0, m1)
    assert max(merr.values()) < 1e-5 and merr["W0"] < 2e-6, merr
    for l in range(len(v0["W"])):
        np.testing.assert_allclose(v1["W"][l], v0["W"][l], rtol=2e-3, atol=1e-12)
    np.testing.assert_allclose(v1["gamma"], v0["gamma"], rtol=2e-3, atol=1e-12)
    np.testing.assert_allclose(v1["beta"], v0["beta"], rtol=2e-3, atol=1e-12)
    # the moving statistics do not depend on the schedule at all
    assert err["mov_mean"] == 0.0 and err["mov_var"] == 0.0


def _run_epochs(x, y, p, tr, va, perms, chain, drop_p, use_graph, seed=5):
    from locator_amd.train import EpochRunner
    net = build_net(x, y, p, drop_p=drop_p, seed=seed)
    runner = EpochRunner(net, tr, va, 32, use_graph=use_graph, chain=chain)
    assert runner.chain == bool(chain)
    hist, masks = [], []
    for perm in perms:
        hist.append(runner.run_epoch(perm))
        if runner.masks is not None:
            masks.append(runner.masks.cpu().numpy().reshape(runner.steps, 32, net.d.Hp).copy())
    _sync()
    m, v = net.export_adam()
    return net, hist, net.export_params(), m, v, masks


@pytest.mark.parametrize("K,nlayers,n_train,drop_p", [
    (5000, 4, 74, 0.25),        # fewer k-tiles than workgroups (one tile each), last minibatch of 10 rows
    (40010, 10, 96, 0.25),      # several k-tiles per workgroup, K not a multiple of 32 (padded SNPs stay zero)
    (3000, 2, 40, 0.0),         # no dropout, one full + one 8-row minibatch
    (2000, 3, 64, 0.25),        # nlayers 3: Dropout directly on layer 1's output (mask applied by the reduction)
])
def test_chained_epochs_equal_unchained_epochs(K, nlayers, n_train, drop_p):
    width = 256
    x, y, p, rng = make_problem(n_train + 20, K, width, nlayers, seed=K % 97)
    tr, va = np.arange(n_train), np.arange(n_train, n_train + 20)
    perms = [np.random.default_rng(7 + e).permutation(n_train) for e in range(3)]
    _, h0, p0, m0, v0, _ = _run_epochs(x, y, p, tr, va, perms, False, drop_p, True)
    _, h1, p1, m1, v1, _ = _run_epochs(x, y, p, tr, va, perms, True, drop_p, True)
    assert maxerr(h0, h1) < 2e-5, (h0, h1)          # 3 epochs x (train loss, validation loss)
    _assert_same_fit(p0, p1, m0, m1, v0, v1)


def test_chained_epochs_equal_unchained_epochs_at_the_baseline_width_of_snps():
    """BASELINE.json's SNP count (100,000: 3,125 k-tiles, 12 or 13 per workgroup, every iteration of the hand-counted
    pipeline in steady state under a saturated memory system), 160 training rows = 5 full minibatches, 2 epochs."""
    K, width, nlayers, n_train = 100000, 256, 10, 160
    x, y, p, rng = make_problem(n_train + 20, K, width, nlayers, seed=100)
    tr, va = np.arange(n_train), np.arange(n_train, n_train + 20)
    perms = [np.random.default_rng(70 + e).permutation(n_train) for e in range(2)]
    _, h0, p0, m0, v0, _ = _run_epochs(x, y, p, tr, va, perms, False, 0.25, True)
    _, h1, p1, m1, v1, _ = _run_epochs(x, y, p, tr, va, perms, True, 0.25, True)
    assert maxerr(h0, h1) < 2e-5, (h0, h1)
    _assert_same_fit(p0, p1, m0, m1, v0, v1)


def test_chained_epochs_match_the_oracle_fit():
    """Chained schedule against oracle.fit with the same permutations and the device's dropout masks: 4 epochs x 4
    steps (last minibatch of 4 rows) at width 256.  Tolerances of test_short_fit_trajectory_matches_oracle_fit."""
    K, width, nlayers = 2500, 256, 4
    x, y, p, rng = make_problem(130, K, width, nlayers, seed=33)
    tr, va, pr_rows = np.arange(0, 100), np.arange(100, 120), np.arange(120, 130)
    perms = [np.random.default_rng(100 + e).permutation(100) for e in range(4)]
    net, hist, pg, _, _, masks = _run_epochs(x, y, p, tr, va, perms, True, 0.25, True, seed=11)
    pref = O.copy_params(p)
    href, _ = O.fit(pref, x[tr], y[tr], x[va], y[va], batch_size=32, max_epochs=4, patience=100, drop_p=0.25,
                    perm_fn=lambda e: perms[e], mask_fn=lambda e, s, nb: masks[e][s, :nb, :width])
    assert maxerr([h[0] for h in hist], href["loss"]) < 5e-4
    assert maxerr([h[1] for h in hist], href["val_loss"]) < 5e-4
    err = params_err(pg, pref)
    assert max(err.values()) < 2e-4, err
    yhat = torch.zeros((10, 2), device="cuda")
    net.predict_rows(torch.from_numpy(pr_rows.astype(np.int32)).cuda(), 10, yhat)
    _sync()
    ref = O.predict(pref, x[pr_rows])
    rel = np.abs(yhat.cpu().numpy() - ref) / np.maximum(np.abs(ref), 1.0)
    assert rel.max() < 1e-3, rel.max()


def test_chained_graph_replay_equals_eager_enqueue():
    """The captured epoch (graph replay) and the eagerly enqueued one run the same chained launches: bit-identical."""
    K, width, nlayers = 3000, 256, 4
    x, y, p, rng = make_problem(90, K, width, nlayers, seed=3)
    tr, va = np.arange(70), np.arange(70, 90)
    perms = [np.random.default_rng(e).permutation(70) for e in range(3)]
    a = _run_epochs(x, y, p, tr, va, perms, True, 0.25, False)
    b = _run_epochs(x, y, p, tr, va, perms, True, 0.25, True)
    assert a[1] == b[1]
    assert max(params_err(a[2], b[2]).values()) == 0.0


def test_chain_is_refused_where_it_does_not_apply():
    from locator_amd.train import EpochRunner
    x, y, p, rng = make_problem(60, 500, 64, 4, seed=1)
    net = build_net(x, y, p)
    assert not net.chain_supported()                                    # width 64
    assert not EpochRunner(net, np.arange(40), np.arange(40, 60), 32, chain=True).chain
    x, y, p, rng = make_problem(60, 500, 256, 1, seed=1)
    assert not build_net(x, y, p).chain_supported()                      # no hidden stack, Dropout on the BatchNorm output
    x, y, p, rng = make_problem(80, 500, 256, 4, seed=1)
    net = build_net(x, y, p)
    assert net.chain_supported()
    assert not EpochRunner(net, np.arange(60), np.arange(60, 80), 64).chain      # --batch_size > 32
    loss = torch.zeros(1, device="cuda")
    rows = torch.arange(32, dtype=torch.int32, device="cuda")
    net.set_batch(32)
    with pytest.raises(Exception, match="batch statistics"):
        net.train_step_chain(rows, 32, 1, torch.ones(32 * 256, dtype=torch.uint8, device="cuda"), loss, None, rows, 32, False)
