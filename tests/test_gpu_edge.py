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
    assert maxerr(hist["loss"], href["loss"]) < 5e-4, (hist["loss"], href["loss"])
    assert maxerr(hist["val_loss"], href["val_loss"]) < 5e-4
    errs = params_err(net.export_params(), pref)
    assert max(errs.values()) < 1e-4, errs
    # padding of the width / SNP axes stayed exactly zero through training
    d, lay = net.d, net.lay
    flat = net.params.cpu().numpy()
    wh = flat[lay.wh:lay.wh + (d.L - 1) * d.Hp * d.Hp].reshape(d.L - 1, d.Hp, d.Hp)
    assert not wh[:, d.H:, :].any() and not wh[:, :, d.H:].any()
    assert not flat[lay.b1 + d.H:lay.b1 + d.Hp].any() and not flat[lay.gamma + d.K:lay.gamma + d.Kp].any()


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
        LocatorNet(X, Y, 40, 64, 1)
    with pytest.raises(_lib.LocatorHipError, match="width"):
        LocatorNet(X, Y, 40, 600, 4)
    # --batch_size: 1..128; above 32 only on the fused-stack widths up to 256 and with Dropout after layer >= 2
    from locator_amd.train import EpochRunner
    tr, va = np.arange(6), np.arange(6, 8)
    with pytest.raises(ValueError, match="batch_size"):
        EpochRunner(LocatorNet(X, Y, 40, 64, 4), tr, va, 129)
    with pytest.raises(ValueError, match="batch_size"):
        EpochRunner(LocatorNet(X, Y, 40, 64, 4), tr, va, 0)
    with pytest.raises(ValueError, match="width"):
        EpochRunner(LocatorNet(X, Y, 40, 512, 4), tr, va, 64)
    with pytest.raises(ValueError, match="width"):
        EpochRunner(LocatorNet(X, Y, 40, 32, 4), tr, va, 40)
    with pytest.raises(ValueError, match="nlayers"):
        EpochRunner(LocatorNet(X, Y, 40, 64, 3, 0.25), tr, va, 40)
    EpochRunner(LocatorNet(X, Y, 40, 64, 3, 0.0), tr, va, 40)          # no dropout: any depth >= 2 is fine


_KNOB_PROBE = r"""
import hashlib, numpy as np, torch
from tests.gpu_util import build_net, make_problem
from locator_amd.train import EpochRunner
x, y, p, rng = make_problem(70, 4096, 256, 4, seed=3)
net = build_net(x, y, p, drop_p=0.25, seed=7)
runner = EpochRunner(net, np.arange(60), np.arange(60, 70), 32, use_graph=False)
out = []
for e in range(2):
    out.append(runner.run_epoch(np.random.default_rng(e).permutation(60)))
torch.cuda.synchronize()
print("DIGEST", hashlib.sha1(net.params.cpu().numpy().tobytes()).hexdigest(), repr(out))
"""


def _probe(env):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", _KNOB_PROBE], cwd=root, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0]


def test_speed_knobs_do_not_change_a_single_bit():
    """Cache policy of the layer-1 backward (LOC_L1B_NT), XCD placement / L2 warm-up helpers of the hidden
    stack (LOC_STACK_XCD_STRIDE, LOC_STACK_HELPERS) are speed hints: two epochs of training (width 256, the
    shape the knobs apply to) must leave bit-identical weights, losses and validation losses."""
    ref = _probe({})
    assert _probe({"LOC_L1B_NT": "0"}) == ref
    assert _probe({"LOC_L1B_NT": "15"}) == ref
    assert _probe({"LOC_STACK_XCD_STRIDE": "1", "LOC_STACK_HELPERS": "0"}) == ref
    assert _probe({"LOC_STACK_HELPERS": "3"}) == ref
    # opt-in experiments kept in the tree: the side-stream overlap only reorders launches ->
    # identical bits; the split-K hidden stack changes the summation order -> same losses to 1e-5
    assert _probe({"LOC_SIDE_STREAM": "1"}) == ref
    split = _probe({"LOC_STACK_SPLIT": "4"})
    a = np.array(eval(ref.split(" ", 2)[2]))
    b = np.array(eval(split.split(" ", 2)[2]))
    assert a.shape == b.shape and np.max(np.abs(a - b)) < 1e-5, (a, b)


def test_row_block_backward_meets_the_single_step_parity_bar():
    """The bf16x3 row-block backward (normally used above 32 rows) pushed through the strictest parity tests of
    the default kernel: LOC_L1B_ROWS=1 routes <= 32-row steps of width 256 through it as one row block.  One
    training step within 1e-5 of the fp64 oracle on every weight, five steps, and the golden fixtures."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_golden.py", "-m", "gpu",
                        "-q", "-x", "-p", "no:cacheprovider", "-k", "one_training_step or five_steps or hip_path"],
                       cwd=root, env=dict(os.environ, LOC_L1B_ROWS="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
