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


def _run_epochs(x, y, p, tr, va, perms, chain, drop_p, use_graph, seed=5, batch=32):
    from locator_amd.train import EpochRunner
    net = build_net(x, y, p, drop_p=drop_p, seed=seed)
    runner = EpochRunner(net, tr, va, batch, use_graph=use_graph, chain=chain)
    assert runner.chain == bool(chain)
    hist, masks = [], []
    for perm in perms:
        hist.append(runner.run_epoch(perm))
        if runner.masks is not None:
            masks.append(runner.masks.cpu().numpy().reshape(runner.steps, runner.slot_rows, net.d.Hp).copy())
    _sync()
    m, v = net.export_adam()
    return net, hist, net.export_params(), m, v, masks


@pytest.mark.parametrize("K,nlayers,n_train,drop_p", [
    (5000, 4, 74, 0.25),        # fewer k-tiles than workgroups (one tile each), last minibatch of 10 rows
    (40010, 10, 96, 0.25),      # several k-tiles per workgroup, K not a multiple of 32 (padded SNPs stay zero)
    (3000, 2, 40, 0.0),         # no dropout, one full + one 8-row minibatch
    (2000, 3, 64, 0.25),        # nlayers 3: Dropout directly on layer 1's output (mask applied by the reduction)
])
@pytest.mark.parametrize("width", [512, 490, 256, 128, 100, 64, 40])
def test_chained_epochs_equal_unchained_epochs(K, nlayers, n_train, drop_p, width):
    """Widths padding to 256, 128 (100, 128) and 64 (40, 64): 8, 4, 2 unit tiles per k-tile; the narrower layers own 2 / 4
    k-tiles per workgroup at a time (round 4; K = 5000 and 40010 leave a short last super-tile for both)."""
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


def test_chained_epochs_equal_unchained_epochs_at_config4_width_of_snps():
    """BASELINE.json configs[4]'s SNP count (500,000 after filtering: 15,625 k-tiles, 61 or 62 per workgroup; byte offsets
    into W1 / m / v up to 2^29 in the kernel's 32-bit offset registers), 64 training rows = 2 minibatches, 1 epoch."""
    K, width, nlayers, n_train = 500000, 256, 10, 64
    rng = np.random.default_rng(4)
    af = rng.beta(0.4, 0.9, K).clip(0.02, 0.98).astype(np.float32)
    x = (rng.random((n_train + 10, K), dtype=np.float32) < af).astype(np.uint8) + \
        (rng.random((n_train + 10, K), dtype=np.float32) < af).astype(np.uint8)
    y = rng.normal(0, 1, (n_train + 10, 2))
    p = O.init_params(K, width, nlayers, rng)
    tr, va = np.arange(n_train), np.arange(n_train, n_train + 10)
    perms = [np.random.default_rng(9).permutation(n_train)]
    _, h0, p0, m0, v0, _ = _run_epochs(x, y, p, tr, va, perms, False, 0.25, False)
    _, h1, p1, m1, v1, _ = _run_epochs(x, y, p, tr, va, perms, True, 0.25, False)
    assert maxerr(h0, h1) < 2e-5, (h0, h1)
    _assert_same_fit(p0, p1, m0, m1, v0, v1)
    assert np.abs(p1["W"][0] - p["W"][0]).max() > 1e-4 and np.abs(p1["W"][0][-40:] - p["W"][0][-40:]).max() > 1e-4   # trained, to the last SNP


@pytest.mark.parametrize("width", [512, 256, 128, 64])
def test_chained_epochs_match_the_oracle_fit(width):
    """Chained schedule against oracle.fit with the same permutations and the device's dropout masks: 4 epochs x 4
    steps (last minibatch of 4 rows) at widths 256 / 128 / 64.  Tolerances of test_short_fit_trajectory_matches_oracle_fit."""
    K, nlayers = 2500, 4
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


@pytest.mark.parametrize("width", [512, 256, 128, 64])
def test_chained_graph_replay_equals_eager_enqueue(width):
    """The captured epoch (graph replay) and the eagerly enqueued one run the same chained launches: bit-identical."""
    K, nlayers = 3000, 4
    x, y, p, rng = make_problem(90, K, width, nlayers, seed=3)
    tr, va = np.arange(70), np.arange(70, 90)
    perms = [np.random.default_rng(e).permutation(70) for e in range(3)]
    a = _run_epochs(x, y, p, tr, va, perms, True, 0.25, False)
    b = _run_epochs(x, y, p, tr, va, perms, True, 0.25, True)
    assert a[1] == b[1]
    assert max(params_err(a[2], b[2]).values()) == 0.0


def test_chain_is_refused_where_it_does_not_apply():
    from locator_amd.train import EpochRunner
    x, y, p, rng = make_problem(60, 500, 300, 4, seed=1)
    net = build_net(x, y, p)
    assert not net.chain_supported()                                    # width 300 pads to 320: per-layer kernels
    assert not EpochRunner(net, np.arange(40), np.arange(40, 60), 32, chain=True).chain
    x, y, p, rng = make_problem(60, 500, 1024, 4, seed=1)
    assert not build_net(x, y, p).chain_supported()                      # width 1024: per-layer kernels
    for wdt in (64, 128, 512):
        x, y, p, rng = make_problem(60, 500, wdt, 4, seed=1)
        assert build_net(x, y, p).chain_supported()                      # since round 4
    x, y, p, rng = make_problem(60, 500, 256, 1, seed=1)
    assert not build_net(x, y, p).chain_supported()                      # no hidden stack, Dropout on the BatchNorm output
    x, y, p, rng = make_problem(80, 500, 256, 4, seed=1)
    net = build_net(x, y, p)
    assert net.chain_supported()
    assert EpochRunner(net, np.arange(60), np.arange(60, 80), 64).chain          # --batch_size 33..64 at width 256: two row blocks
    assert not EpochRunner(net, np.arange(60), np.arange(60, 80), 65).chain      # --batch_size > 64
    x5, y5, p5, _ = make_problem(80, 500, 128, 4, seed=1)
    assert not EpochRunner(build_net(x5, y5, p5), np.arange(60), np.arange(60, 80), 64).chain   # two row blocks: width 256 only
    loss = torch.zeros(1, device="cuda")
    rows = torch.arange(32, dtype=torch.int32, device="cuda")
    net.set_batch(32)
    with pytest.raises(Exception, match="batch statistics"):
        net.train_step_chain(rows, 32, 1, torch.ones(32 * 256, dtype=torch.uint8, device="cuda"), loss, None, rows, 32, False)


@pytest.mark.parametrize("width", [512, 128, 64])
def test_chained_epochs_equal_unchained_epochs_at_the_baseline_width_of_snps_narrow(width):
    """100,000 SNPs at widths 128 / 64: 1563 / 782 super-tiles over 256 workgroups, every iteration of the pipeline in
    steady state, the last super-tile short (3,125 k-tiles is odd); at width 512: 3,125 k-tiles of two sub-steps per wave."""
    K, nlayers, n_train = 100000, 10, 96
    x, y, p, rng = make_problem(n_train + 20, K, width, nlayers, seed=width)
    tr, va = np.arange(n_train), np.arange(n_train, n_train + 20)
    perms = [np.random.default_rng(70 + e).permutation(n_train) for e in range(2)]
    _, h0, p0, m0, v0, _ = _run_epochs(x, y, p, tr, va, perms, False, 0.25, True)
    _, h1, p1, m1, v1, _ = _run_epochs(x, y, p, tr, va, perms, True, 0.25, True)
    assert maxerr(h0, h1) < 2e-5, (h0, h1)
    _assert_same_fit(p0, p1, m0, m1, v0, v1)


@pytest.mark.parametrize("K,nlayers,n_train,batch,drop_p", [
    (5000, 4, 150, 64, 0.25),       # fewer k-tiles than workgroups; minibatches of 64, 64, 22 rows (the last uses ONE row block)
    (40010, 10, 170, 48, 0.25),     # several k-tiles per workgroup; 48, 48, 48, 26
    (3000, 4, 100, 33, 0.0),        # 33 rows: the second block carries one row; last minibatch of 1 row
    (100000, 10, 192, 64, 0.25),    # the baseline width of SNPs, three full minibatches
])
def test_chained_epochs_of_two_row_blocks_equal_unchained_epochs(K, nlayers, n_train, batch, drop_p):
    """--batch_size 33..64 at width 256 (round 4, late): l1_bwd_adam_chain_kernel<13, 8, 2> - dZ of 64 rows in LDS, two
    fp32 MFMA chains of 16 per unit each way, the hidden-layer tail with a run-time block count - against the unchained
    schedule of the same batch size (large-M bf16x3 forward + row-block backward + tail launch), graph replay and eager."""
    x, y, p, rng = make_problem(n_train + 20, K, 256, nlayers, seed=K % 97 + batch)
    tr, va = np.arange(n_train), np.arange(n_train, n_train + 20)
    perms = [np.random.default_rng(50 + e).permutation(n_train) for e in range(3)]
    _, h0, p0, m0, v0, _ = _run_epochs(x, y, p, tr, va, perms, False, drop_p, True, batch=batch)
    _, h1, p1, m1, v1, _ = _run_epochs(x, y, p, tr, va, perms, True, drop_p, True, batch=batch)
    _, h2, p2, m2, v2, _ = _run_epochs(x, y, p, tr, va, perms, True, drop_p, False, batch=batch)
    assert maxerr(h0, h1) < 2e-5, (h0, h1)
    _assert_same_fit(p0, p1, m0, m1, v0, v1)
    assert h1 == h2 and params_err(p1, p2) == {k: 0.0 for k in params_err(p1, p2)}      # graph replay = eager, bit for bit
    assert np.abs(p1["W"][0] - p["W"][0]).max() > 1e-4


def test_chained_epochs_of_two_row_blocks_match_the_oracle_fit():
    """Chained 48-row steps against oracle.fit with the same permutations and the device's dropout masks (3 epochs x 3
    steps, last minibatch of 4 rows)."""
    K, nlayers, width, batch = 2500, 4, 256, 48
    x, y, p, rng = make_problem(130, K, width, nlayers, seed=34)
    tr, va = np.arange(0, 100), np.arange(100, 120)
    perms = [np.random.default_rng(200 + e).permutation(100) for e in range(3)]
    net, hist, pg, _, _, masks = _run_epochs(x, y, p, tr, va, perms, True, 0.25, True, seed=11, batch=batch)
    pref = O.copy_params(p)
    href, _ = O.fit(pref, x[tr], y[tr], x[va], y[va], batch_size=batch, max_epochs=3, patience=100, drop_p=0.25,
                    perm_fn=lambda e: perms[e], mask_fn=lambda e, s, nb: masks[e][s, :nb, :width])
    assert maxerr([h[0] for h in hist], href["loss"]) < 5e-4
    assert maxerr([h[1] for h in hist], href["val_loss"]) < 5e-4
    err = params_err(pg, pref)
    assert max(err.values()) < 2e-4, err


@pytest.mark.parametrize("K,n_b,n_b_next,width", [(4000, 32, 32, 256), (4000, 17, 5, 256), (9990, 32, 32, 256),
                                                  (100000, 32, 10, 256), (4000, 32, 32, 512), (9990, 17, 5, 512),
                                                  (100000, 32, 10, 512)])
def test_chain_kernel_against_the_two_kernels_it_replaces(K, n_b, n_b_next, width):
    """loc_l1_backward_adam_chain through the C ABI against loc_l1_backward_adam followed by loc_l1_forward on the next
    minibatch, same inputs: W1 / m / v, b1, gamma / beta and their moments, the next step's [scale|shift|mean|rstd], and
    the next minibatch's layer-1 activations.  dZ1 and the batch statistics are synthetic (the kernels do not care)."""
    import ctypes as C
    from locator_amd import _lib
    x, y, p, rng = make_problem(80, K, width, 2, seed=K % 89 + n_b)
    net = build_net(x, y, p, drop_p=0.0)
    lib, d, lay = net.lib, net.d, net.lay
    Kp, Hp = d.Kp, d.Hp
    dev = "cuda"
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = torch.from_numpy(rng.choice(80, 32, replace=False).astype(np.int32)).to(dev)
    rows_next = torch.from_numpy(rng.choice(80, 32, replace=False).astype(np.int32)).to(dev)
    dz = torch.zeros((32, Hp), device=dev)
    dz[:n_b] = torch.from_numpy(rng.normal(0, 0.05, (n_b, Hp)).astype(np.float32)).to(dev)      # rows beyond n_b are zero
    # non-trivial Adam state and a third step, so that every term of the update is exercised
    net.adam_m.copy_(torch.from_numpy(rng.normal(0, 1e-3, net.adam_m.numel()).astype(np.float32)))
    # (second moments of the size of the squared first moments, as in a real fit: with v far below m^2 one Adam step
    # amplifies the round-off of the gradient a thousandfold and the comparison would measure that)
    net.adam_v.copy_(torch.from_numpy(((0.5 + rng.random(net.adam_v.numel())) * 1e-6).astype(np.float32)))
    pad = torch.arange(Kp, device=dev) >= K                       # padded SNPs carry zero state, as after init / import
    for buf in (net.adam_m, net.adam_v):
        buf[lay.gamma:lay.gamma + Kp][pad] = 0
        buf[lay.beta:lay.beta + Kp][pad] = 0
        buf[lay.w1:lay.w1 + Hp * Kp].view(-1)[:] = buf[lay.w1:lay.w1 + Hp * Kp]  # (W1S padding rows stay as drawn: zero gradient either way)
    net.t_base_t.fill_(2)

    def stats_of(r, n):       # [mean | biased var] of a minibatch, [2][Kp]
        xb = x[r.cpu().numpy()[:n]].astype(np.float64)
        s = np.zeros((2, Kp), np.float32)
        s[0, :K], s[1, :K] = xb.mean(0), xb.var(0)
        return torch.from_numpy(s).to(dev)

    next_stats = stats_of(rows_next, n_b_next).contiguous()
    cur = stats_of(rows, n_b)
    P0, M0, V0 = net.params.clone(), net.adam_m.clone(), net.adam_v.clone()

    def bn4_now(params):
        g, b = params[lay.gamma:lay.gamma + Kp], params[lay.beta:lay.beta + Kp]
        rstd = torch.where(pad, torch.zeros_like(cur[1]), 1.0 / torch.sqrt(cur[1] + 1e-3))
        sc = g * rstd
        return torch.cat([sc, b - cur[0] * sc, cur[0], rstd]).contiguous()

    grid = max(1, min(net.l1_bwd_grid // 2, Kp // 32))
    out = {}
    for which in ("pair", "chain"):
        net.params.copy_(P0); net.adam_m.copy_(M0); net.adam_v.copy_(V0)
        Pp, Mp, Vp = net.params.data_ptr(), net.adam_m.data_ptr(), net.adam_v.data_ptr()
        o = lambda base, off: C.c_void_p(base + 4 * off)
        bn4 = bn4_now(net.params)
        partial = torch.zeros(512 * 32 * Hp, device=dev)
        a1 = torch.zeros((32, Hp), device=dev)
        common = [o(Pp, lay.w1), o(Mp, lay.w1), o(Vp, lay.w1), o(Pp, lay.gamma), o(Pp, lay.beta), o(Mp, lay.gamma),
                  o(Vp, lay.gamma), o(Mp, lay.beta), o(Vp, lay.beta), o(Pp, lay.b1), o(Mp, lay.b1), o(Vp, lay.b1)]
        tail = [C.c_void_p(net.alpha_tab.data_ptr()), len(net.alpha_tab), C.c_void_p(net.lr_t.data_ptr()),
                C.c_void_p(net.t_base_t.data_ptr()), 1]
        if which == "pair":
            gbs = torch.zeros(4 * Kp + Hp, device=dev)
            _lib.check(lib.loc_l1_backward_adam(C.c_void_p(net.X.data_ptr()), net.X.stride(0), C.c_void_p(rows.data_ptr()),
                                                n_b, C.byref(d), C.c_void_p(bn4.data_ptr()), C.c_void_p(dz.data_ptr()),
                                                *common, C.c_void_p(gbs.data_ptr()), *tail, net.l1_bwd_grid,
                                                C.c_void_p(next_stats.data_ptr()), C.c_void_p(bn4.data_ptr()), None,
                                                C.byref(net.tuning), st), "backward")
            _lib.check(lib.loc_l1_forward(C.c_void_p(net.X.data_ptr()), net.X.stride(0), C.c_void_p(rows_next.data_ptr()),
                                          n_b_next, C.byref(d), C.c_void_p(bn4.data_ptr()), o(Pp, lay.w1), o(Pp, lay.b1),
                                          C.c_void_p(partial.data_ptr()), net.l1_fwd_grid, C.c_void_p(a1.data_ptr()),
                                          None, None, C.c_float(1.0), st), "forward")
        else:
            _lib.check(lib.loc_l1_backward_adam_chain(C.c_void_p(net.X.data_ptr()), net.X.stride(0),
                                                      C.c_void_p(rows.data_ptr()), n_b, C.c_void_p(rows_next.data_ptr()),
                                                      n_b_next, C.byref(d), C.c_void_p(bn4.data_ptr()),
                                                      C.c_void_p(next_stats.data_ptr()), C.c_void_p(dz.data_ptr()), *common,
                                                      *tail, grid, C.c_void_p(partial.data_ptr()), partial.numel(),
                                                      C.byref(net.tuning), st), "chain")
            z = partial[:grid * 32 * Hp].view(grid, 32, Hp).sum(0) + net.params[lay.b1:lay.b1 + Hp]
            a1 = torch.where(z > 0, z, torch.expm1(z))
        _sync()
        out[which] = dict(P=net.params.cpu().numpy().copy(), M=net.adam_m.cpu().numpy().copy(),
                          V=net.adam_v.cpu().numpy().copy(), bn4=bn4.cpu().numpy().copy(), a1=a1.cpu().numpy().copy())
    a, b = out["pair"], out["chain"]
    sl = lambda off, n: slice(off, off + n)
    for name, off, n in (("W1", lay.w1, Hp * Kp), ("gamma", lay.gamma, Kp), ("beta", lay.beta, Kp), ("b1", lay.b1, Hp)):
        assert maxerr(a["P"][sl(off, n)], b["P"][sl(off, n)]) < 2e-6, name           # one Adam step: round-off of the gradient
        assert maxerr(a["M"][sl(off, n)], b["M"][sl(off, n)]) < 1e-6, name           # = 1e-5 on a gradient of size ~0.3: the two
        #   kernels associate xhat differently (x * scale + shift, whose two terms cancel, against gamma * xn + beta)
        np.testing.assert_allclose(b["V"][sl(off, n)], a["V"][sl(off, n)], rtol=2e-3, atol=1e-13, err_msg=name)
    assert np.abs(a["P"][sl(lay.w1, Hp * Kp)] - P0.cpu().numpy()[sl(lay.w1, Hp * Kp)]).max() > 1e-4     # it did train
    np.testing.assert_allclose(b["bn4"], a["bn4"], rtol=2e-6, atol=2e-6)      # rstd reaches 1 / sqrt(BN eps) = 31.6
    assert maxerr(a["a1"][:n_b_next], b["a1"][:n_b_next]) < 2e-5 * max(1.0, float(np.abs(a["a1"]).max()))


_CHAIN_BITCMP = """
import hashlib, sys
import numpy as np
sys.path.insert(0, sys.argv[2])
if sys.argv[1] != "-":
    from locator_amd import _lib
    _lib.use_library(sys.argv[1])
import torch
from tests.gpu_util import build_net, make_problem
from locator_amd.train import EpochRunner
h = hashlib.sha256()
for width, K in ((512, 30010), (256, 40010), (128, 20000), (64, 9000)):
    x, y, p, rng = make_problem(116, K, width, 4, seed=width)
    net = build_net(x, y, p, drop_p=0.25, seed=5)
    r = EpochRunner(net, np.arange(96), np.arange(96, 116), 32, use_graph=False, chain=True)
    for e in range(2):
        l, v = r.run_epoch(np.random.default_rng(7 + e).permutation(96))
        h.update(np.float64([l, v]).tobytes())
    torch.cuda.synchronize()
    h.update(net.params.cpu().numpy().tobytes()); h.update(net.adam_m.cpu().numpy().tobytes()); h.update(net.adam_v.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
"""


def test_hand_counted_wait_of_the_chained_kernel_equals_its_drained_build_bit_for_bit(tmp_path):
    """the chained kernel's streaming loads are untracked asm with ONE hand-counted wait per iteration.  The
    parity-debug twin library (`make debug_drain`, part of build()) compiles it with every count replaced by vmcnt(0); two
    chained epochs at each width (several k-tiles per workgroup, a short last super-tile) must leave identical losses,
    weights and Adam moments.  Separate processes: one library per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drain = os.path.join(root, "locator_amd", "liblocator_hip_drain.so")
    if not os.path.exists(drain):
        pytest.skip("locator_amd/liblocator_hip_drain.so not built (make -C locator_amd/csrc debug_drain)")
    script = tmp_path / "chain_bitcmp.py"
    script.write_text(_CHAIN_BITCMP)
    out = []
    for lib in ("-", drain):
        r = subprocess.run([sys.executable, str(script), lib, root], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert out[0] == out[1], out


def _run_xchain(x, y, p, tr, va, perms, xchain, use_graph, width_seed=5, batch=32):
    """Epochs through EpochRunner with (or without) the cross-epoch hand-over: epoch e is given epoch e + 1's permutation."""
    from locator_amd.train import EpochRunner
    net = build_net(x, y, p, drop_p=0.25, seed=width_seed)
    runner = EpochRunner(net, tr, va, batch, use_graph=use_graph, chain=True, xchain=xchain)
    assert runner.xchain == xchain
    hist = []
    for e, perm in enumerate(perms):
        runner.start_epoch(perm, None, perms[e + 1] if xchain and e + 1 < len(perms) else None)
        hist.append(runner.finish_epoch())
    _sync()
    m, v = net.export_adam()
    return net, hist, net.export_params(), m, v


@pytest.mark.parametrize("width,K,n_train,batch", [(512, 30010, 96, 32), (256, 40010, 96, 32), (256, 5000, 74, 32), (128, 20000, 96, 32),
                                                   (64, 9000, 70, 32), (256, 20000, 150, 64)])
def test_cross_epoch_chaining_equals_the_per_epoch_schedule(width, K, n_train, batch):
    """Round 4: the last step of an epoch also computes the first layer-1 forward of the next epoch (its rows and batch
    statistics are known an epoch early), so every epoch but the first starts from a hand-over instead of an unchained
    forward; the validation sweep in between works in a second workspace.  Five epochs (eager, eager, then one captured
    graph per epoch parity and a replay of each) against the per-epoch chained schedule: same bars as chained against
    unchained (two fp32 evaluations of the same steps); graph replay against eager enqueue: bit-identical."""
    x, y, p, rng = make_problem(n_train + 20, K, width, 4, seed=K % 89)
    tr, va = np.arange(n_train), np.arange(n_train, n_train + 20)
    perms = [np.random.default_rng(17 + e).permutation(n_train) for e in range(5)]
    _, h0, p0, m0, v0 = _run_xchain(x, y, p, tr, va, perms, False, True, batch=batch)
    _, h1, p1, m1, v1 = _run_xchain(x, y, p, tr, va, perms, True, True, batch=batch)
    assert maxerr(h0, h1) < 5e-5, (h0, h1)
    _assert_same_fit(p0, p1, m0, m1, v0, v1)
    _, h2, p2, m2, v2 = _run_xchain(x, y, p, tr, va, perms, True, False, batch=batch)
    assert h1 == h2 and max(params_err(p1, p2).values()) == 0.0
