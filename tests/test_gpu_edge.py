"""Edge cases on the GPU path: shapes the reference tolerates (or crashes on) that the HIP path must
handle or reject loudly — odd layer counts, widths that are not multiples of 32, fewer SNPs than a
tile, training sets smaller than a batch, batch sizes below 32 and between 33 and 128, no samples to
predict."""
import numpy as np
import pytest
import torch

from oracle import locator_oracle as O
from tests.gpu_util import build_net, make_problem, maxerr, params_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,width,nlayers,batch,n_train", [
    (20, 8, 2, 32, 50),        # K < one 32-SNP tile, tiny width (padded to 32), minimum depth
    (100, 100, 7, 32, 40),     # odd depth: 3 layers before dropout, 4 after; width padded 100 -> 128
    (300, 64, 4, 7, 30),       # --batch_size 7: five steps per epoch, last batch of 2
    (64, 256, 3, 32, 12),      # training set smaller than one batch
    (257, 33, 5, 16, 33),      # width 33 -> 64, K = 8 tiles + 1 SNP, last batch of 1
    (300, 64, 4, 64, 150),     # --batch_size 64: two row blocks per step, last batch of 22 (one block)
    (5830, 256, 10, 48, 200),  # --batch_size 48 on the fixture's shape: second row block half full, last batch of 8
    (1000, 128, 6, 33, 100),   # --batch_size 33: one row spills into the second block; last batch of 1
    (700, 100, 5, 64, 64),     # exactly one full 64-row step per epoch; width 100 -> 128
    (900, 256, 4, 128, 300),   # --batch_size 128: four row blocks, last batch of 44 (two blocks)
    (400, 64, 6, 96, 200),     # --batch_size 96: three row blocks, last batch of 8
    (2048, 128, 4, 100, 230),  # --batch_size 100: fourth block holds 4 rows; last batch of 30
    (300, 64, 4, 200, 450),    # --batch_size 200 (> 128: row blocks streamed from L2, 256-row scratch); last batch of 50
    (1000, 256, 6, 256, 600),  # --batch_size 256 at the default width: eight row blocks; last batch of 88
    (500, 128, 4, 300, 700),   # --batch_size 300: ten row blocks (run-time block count past the eight waves); last 100
    (700, 256, 10, 129, 400),  # --batch_size 129: one row past the row-block kernels; last batch of 13
    (300, 600, 3, 32, 70),     # --width 600 (> 512: per-layer kernels, 19 unit tiles, single-buffered layer-1 forward)
    (97, 1024, 2, 16, 40),     # --width 1024, the limit: 32 unit tiles; K = 3 tiles + 1 SNP
    (500, 520, 4, 32, 64),     # --width 520 pads to 544 (17 unit tiles)
])
def test_fit_matches_oracle_on_odd_shapes(K, width, nlayers, batch, n_train):
    """3 epochs of fit (eager epoch 0, captured graph afterwards) vs oracle.fit with the same init,
    permutations and the device's dropout masks.  Tolerance 5e-4 on per-epoch losses."""
    from locator_amd.train import EpochRunner
    n_val = 9
    x, y, p, rng = make_problem(n_train + n_val, K, width, nlayers, seed=K + width)
    tr, va = np.arange(n_train), np.arange(n_train, n_train + n_val)
    net = build_net(x, y, p, drop_p=0.25, seed=5)
    runner = EpochRunner(net, tr, va, batch, use_graph=True)
    perms = [np.random.default_rng(e).permutation(n_train) for e in range(3)]
    masks, hist = [], {"loss": [], "val_loss": []}
    for e in range(3):
        l, vl = runner.run_epoch(perms[e])
        masks.append(runner.masks.cpu().numpy().reshape(runner.steps, runner.slot_rows, net.d.Hp).copy())
        hist["loss"].append(l)
        hist["val_loss"].append(vl)
    pref = O.copy_params(p)
    href, _ = O.fit(pref, x[tr], y[tr], x[va], y[va], batch_size=batch, max_epochs=3, patience=100, drop_p=0.25,
                    perm_fn=lambda e: perms[e], mask_fn=lambda e, s, nb: masks[e][s, :nb, :width])
    # 5e-4 on the epoch losses, 1e-4 on every weight.  Where a wide, freshly initialised network overshoots (width 600:
    # the epoch loss goes 1.57 -> 2.45 -> 1.68) three epochs amplify fp32 round-off beyond that - one step of the same
    # net matches the oracle to 7e-8 / 1e-6 - so the bar there is the fp32 NumPy oracle's own distance from the fp64 one
    dl, dv = maxerr(hist["loss"], href["loss"]), maxerr(hist["val_loss"], href["val_loss"])
    errs = params_err(net.export_params(), pref)
    if max(dl, dv) >= 5e-4 or max(errs.values()) >= 1e-4:
        p32 = O.cast_params(p, np.float32)
        h32, _ = O.fit(p32, x[tr], y[tr].astype(np.float32), x[va], y[va].astype(np.float32), batch_size=batch,
                       max_epochs=3, patience=100, drop_p=0.25, perm_fn=lambda e: perms[e],
                       mask_fn=lambda e, s, nb: masks[e][s, :nb, :width])
        floor = max(maxerr(h32["loss"], href["loss"]), maxerr(h32["val_loss"], href["val_loss"]))
        wfloor = max(params_err(p32, pref).values())
        assert width > 512 and floor > 1e-4, (dl, dv, floor)       # only the wide, overshooting cases may need this
        assert max(dl, dv) < 3 * floor + 5e-4, (dl, dv, floor)
        assert max(errs.values()) < 3 * wfloor + 1e-4, (errs, wfloor)
    # padding of the width / SNP axes stayed exactly zero through training
    d, lay = net.d, net.lay
    flat = net.params.cpu().numpy()
    wh = flat[lay.wh:lay.wh + (d.L - 1) * d.Hp * d.Hp].reshape(d.L - 1, d.Hp, d.Hp)
    assert not wh[:, d.H:, :].any() and not wh[:, :, d.H:].any()
    assert not flat[lay.b1 + d.H:lay.b1 + d.Hp].any() and not flat[lay.gamma + d.K:lay.gamma + d.Kp].any()


@pytest.mark.parametrize("K,width,batch,n_train,drop_p", [
    (300, 64, 32, 100, 0.25),      # last batch of 4
    (5830, 256, 32, 90, 0.25),     # the fixture's SNP count at the default width; last batch of 26
    (97, 33, 7, 30, 0.5),          # K = 3 tiles + 1 SNP, width padded 33 -> 64, batch 7
    (200, 256, 32, 64, 0.0),       # no dropout at all
])
def test_single_hidden_layer_puts_dropout_on_the_batchnorm_output(K, width, batch, n_train, drop_p):
    """--nlayers 1 (accepted by the reference, locator.py:77, :319-323): floor(1/2) = 0 Dense layers come before the
    Dropout layer, so it masks the BatchNorm output (a K-wide mask per row), then Dense(width, elu), Dense(2), Dense(2).
    3 epochs of fit vs oracle.fit with the same init, permutations and the DEVICE's masks; then predict."""
    from locator_amd.train import EpochRunner
    n_val = 9
    x, y, p, rng = make_problem(n_train + n_val, K, width, 1, seed=K + width)
    assert len(p["W"]) == 3
    tr, va = np.arange(n_train), np.arange(n_train, n_train + n_val)
    net = build_net(x, y, p, drop_p=drop_p, seed=5)
    assert net.d.L == 1 and net.d.n_pre == 0 and net.mask_width == net.d.Kp and not net.use_fused
    runner = EpochRunner(net, tr, va, batch, use_graph=True)
    perms = [np.random.default_rng(e).permutation(n_train) for e in range(3)]
    masks, hist = [], {"loss": [], "val_loss": []}
    for e in range(3):
        l, vl = runner.run_epoch(perms[e])
        if drop_p > 0:
            masks.append(runner.masks.cpu().numpy().reshape(runner.steps, 32, net.d.Kp).copy())
            assert 0.8 * (1 - drop_p) < masks[-1][:, :, :K].mean() < 1.2 * (1 - drop_p)
        hist["loss"].append(l)
        hist["val_loss"].append(vl)
    pref = O.copy_params(p)
    href, _ = O.fit(pref, x[tr], y[tr], x[va], y[va], batch_size=batch, max_epochs=3, patience=100, drop_p=drop_p,
                    perm_fn=lambda e: perms[e], mask_fn=lambda e, s, nb: masks[e][s, :nb, :K])
    assert maxerr(hist["loss"], href["loss"]) < 5e-4, (hist["loss"], href["loss"])
    assert maxerr(hist["val_loss"], href["val_loss"]) < 5e-4
    errs = params_err(net.export_params(), pref)
    assert max(errs.values()) < 1e-4, errs
    n = x.shape[0]
    for rows_n in (n, min(n, 35)):                           # more than 32 rows (large-M layer 1 + heads per 32) and a remainder
        r = torch.arange(rows_n, dtype=torch.int32, device="cuda")
        yhat = torch.zeros((rows_n, 2), device="cuda")
        net.predict_rows(r, rows_n, yhat)
        torch.cuda.synchronize()
        assert maxerr(yhat.cpu().numpy(), O.predict(pref, x[:rows_n])) < 2e-4


def test_no_dropout_and_dropout_on_first_layer():
    """--dropout_prop 0 (no mask anywhere) and nlayers 2/3 (Dropout directly after the layer-1 Dense)."""
    for nlayers, drop_p in [(2, 0.5), (3, 0.25), (4, 0.0)]:
        x, y, p, rng = make_problem(48, 90, 64, nlayers, seed=nlayers)
        net = build_net(x, y, p, drop_p=drop_p)
        idx = rng.choice(48, 32, replace=False)
        mask_np = (rng.random((32, 64)) >= drop_p).astype(np.uint8)
        loss = torch.zeros(1, device="cuda")
        rows = torch.from_numpy(idx.astype(np.int32)).cuda()
        net.train_step(rows, 32, 1, torch.from_numpy(mask_np).cuda() if drop_p > 0 else None, loss)
        torch.cuda.synchronize()
        pr = O.copy_params(p)
        m, v = O.zeros_like_trainable(pr), O.zeros_like_trainable(pr)
        ref = O.train_step(pr, m, v, 1, 1e-3, x[idx], y[idx], mask_np, drop_p)
        assert abs(loss.item() - ref) < 2e-5
        assert max(params_err(net.export_params(), pr).values()) < 1e-5, nlayers


def test_predict_zero_rows_and_non_multiple_of_32():
    x, y, p, rng = make_problem(70, 128, 64, 4)
    net = build_net(x, y, p)
    yhat = torch.zeros((70, 2), device="cuda")
    net.predict_rows(torch.arange(70, dtype=torch.int32, device="cuda"), 0, yhat)      # no-op, no error
    net.predict_rows(torch.arange(70, dtype=torch.int32, device="cuda"), 70, yhat)
    torch.cuda.synchronize()
    assert maxerr(yhat.cpu().numpy(), O.predict(p, x)) < 2e-5


def test_cli_without_samples_to_predict(tmp_path):
    """Every sample has a known location: the reference would call model.predict on an empty array
    (SURVEY Q11); here predlocs.txt is written with the header only and the run completes."""
    import os
    import pandas as pd
    from locator_amd import locator as L
    rng = np.random.default_rng(0)
    n, K = 60, 200
    af = rng.uniform(0.1, 0.9, K)
    g = rng.binomial(2, af, (n, K))
    mat = tmp_path / "m.txt"
    df = pd.DataFrame(g, columns=[f"s{i}" for i in range(K)])
    df.insert(0, "sampleID", [f"id{i}" for i in range(n)])
    df.to_csv(mat, sep="\t", index=False)
    sd = tmp_path / "s.txt"
    pd.DataFrame({"sampleID": [f"id{i}" for i in range(n)], "x": rng.uniform(0, 10, n),
                  "y": rng.uniform(0, 10, n)}).to_csv(sd, sep="\t", index=False)
    out = str(tmp_path / "o")
    assert L.main(["--matrix", str(mat), "--sample_data", str(sd), "--out", out, "--seed", "3", "--max_epochs", "3",
                   "--patience", "3", "--keras_verbose", "0", "--width", "32", "--nlayers", "4"]) == 0
    assert open(out + "_predlocs.txt").read().strip() == "x,y,sampleID"
    assert len(pd.read_csv(out + "_history.txt", sep="\t")) == 3
    assert os.path.exists(out + "_fitplot.pdf")


def test_unsupported_configurations_are_rejected_with_messages():
    from locator_amd import _lib
    from locator_amd.net import LocatorNet, upload_genotypes
    x = np.zeros((8, 40), np.uint8)
    X = upload_genotypes(x)
    Y = torch.zeros((8, 2), device="cuda")
    with pytest.raises(_lib.LocatorHipError, match="nlayers"):
        LocatorNet(X, Y, 40, 64, 0)
    LocatorNet(X, Y, 40, 64, 1)                                        # --nlayers 1 is accepted (round 3)
    with pytest.raises(_lib.LocatorHipError, match="width"):
        LocatorNet(X, Y, 40, 1025, 4)
    LocatorNet(X, Y, 40, 600, 4)                                       # --width above 512 is accepted (round 3)
    # --batch_size: 1..128; above 32 only on the fused-stack widths up to 256 and with Dropout after layer >= 2
    from locator_amd.train import EpochRunner
    tr, va = np.arange(6), np.arange(6, 8)
    with pytest.raises(ValueError, match="batch_size"):
        EpochRunner(LocatorNet(X, Y, 40, 64, 4), tr, va, 4097)
    EpochRunner(LocatorNet(X, Y, 40, 64, 4), tr, va, 129)              # above 128 is accepted since round 3
    with pytest.raises(ValueError, match="batch_size"):
        EpochRunner(LocatorNet(X, Y, 40, 64, 4), tr, va, 0)
    with pytest.raises(ValueError, match="width"):
        EpochRunner(LocatorNet(X, Y, 40, 512, 4), tr, va, 64)
    with pytest.raises(ValueError, match="width"):
        EpochRunner(LocatorNet(X, Y, 40, 32, 4), tr, va, 40)
    with pytest.raises(ValueError, match="nlayers"):
        EpochRunner(LocatorNet(X, Y, 40, 64, 3, 0.25), tr, va, 40)
    EpochRunner(LocatorNet(X, Y, 40, 64, 3, 0.0), tr, va, 40)          # no dropout: any depth >= 2 is fine
    with pytest.raises(ValueError, match="nlayers"):
        EpochRunner(LocatorNet(X, Y, 40, 64, 1, 0.0), tr, va, 40)      # more than 32 rows per step need a hidden stack


def _two_epochs(tuning):
    from locator_amd.train import EpochRunner
    x, y, p, rng = make_problem(70, 4096, 256, 4, seed=3)
    net = build_net(x, y, p, drop_p=0.25, seed=7, tuning=tuning)
    runner = EpochRunner(net, np.arange(60), np.arange(60, 70), 32, use_graph=False)
    out = [runner.run_epoch(np.random.default_rng(e).permutation(60)) for e in range(2)]
    torch.cuda.synchronize()
    return net.params.cpu().numpy().copy(), out


def test_speed_hints_do_not_change_a_single_bit():
    """loc_tuning (include/locator_hip.h): the cache policy of the layer-1 backward (l1b_nt_mask) and the XCD
    placement / L2 warm-up helpers of the hidden stack (stack_xcd_stride, stack_helpers) are speed hints passed
    explicitly in loc_net.tune - the library reads no environment.  Two epochs of training at width 256 (the shape
    the hints apply to) must leave bit-identical weights, losses and validation losses."""
    ref_w, ref_out = _two_epochs(None)
    for tuning in ({"l1b_nt_mask": -1}, {"l1b_nt_mask": 15}, {"l1b_nt_mask": 9},
                   {"stack_xcd_stride": 1, "stack_helpers": -1}, {"stack_helpers": 3}, {"stack_xcd_stride": 2}):
        w, out = _two_epochs(tuning)
        assert np.array_equal(w, ref_w) and out == ref_out, tuning


def test_row_block_backward_meets_the_single_step_parity_bar():
    """The bf16x3 row-block backward (normally used above 32 rows) pushed through the strictest parity bar of the
    default kernel: loc_tuning.l1b_rows = 1 routes <= 32-row steps of width 256 through it as one row block.  One
    training step within 1e-5 of the fp64 oracle on every weight and loss within 2e-5, three consecutive steps."""
    K, width, nlayers = 5830, 256, 10
    x, y, p, rng = make_problem(64, K, width, nlayers, seed=K)
    net = build_net(x, y, p, drop_p=0.25, tuning={"l1b_rows": 1})
    pr = O.copy_params(p)
    m, v = O.zeros_like_trainable(pr), O.zeros_like_trainable(pr)
    for t, n_b in enumerate((32, 21, 32), start=1):
        idx = rng.choice(64, n_b, replace=False)
        mask_np = (rng.random((32, width)) >= 0.25).astype(np.uint8)
        rows = np.zeros(32, np.int32)
        rows[:n_b] = idx
        loss = torch.zeros(1, device="cuda")
        net.train_step(torch.from_numpy(rows).cuda(), n_b, t, torch.from_numpy(mask_np).cuda(), loss)
        torch.cuda.synchronize()
        ref = O.train_step(pr, m, v, t, 1e-3, x[idx], y[idx], mask_np[:n_b, :width], 0.25)
        assert abs(loss.item() - ref) < 2e-5 * max(1, abs(ref))
        errs = params_err(net.export_params(), pr)
        assert max(errs.values()) < (1e-5 if t == 1 else 2e-5), (t, errs)
