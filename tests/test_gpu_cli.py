"""End-to-end runs of the `locator` command line on a GPU against the reference's own example
data (data/test_genotypes.vcf.gz, data/test_sample_data.txt; copied to tests/golden/ as fixtures):
output file set and formats, run-to-run determinism, and accuracy in the
neighbourhood of the reference README's printout (R2 ~ 0.95, mean error ~3.8 on a 0-50 range)."""
import json
import os

import numpy as np
import pandas as pd
import pytest

from locator_amd import genotypes as G
from locator_amd import locator as L

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
VCF = os.path.join(GOLD, "test_genotypes.vcf.gz")
SAMPLES = os.path.join(GOLD, "test_sample_data.txt")


def _run(argv):
    np.random.seed(None)
    assert L.main(argv) == 0


def test_single_run_writes_the_four_files_and_learns(tmp_path, capsys):
    out = str(tmp_path / "test")
    _run(["--vcf", VCF, "--sample_data", SAMPLES, "--out", out, "--seed", "12345", "--max_epochs", "150",
          "--patience", "20", "--keras_verbose", "0"])
    for suffix in ("_predlocs.txt", "_history.txt", "_params.json", "_fitplot.pdf"):
        assert os.path.exists(out + suffix), suffix
    assert not os.path.exists(out + ".weights.npz")
    pl = pd.read_csv(out + "_predlocs.txt")
    assert list(pl.columns) == ["x", "y", "sampleID"] and len(pl) == 50
    assert list(pl["sampleID"]) == [f"msp_{i}" for i in range(50)]
    assert pl["x"].between(-10, 60).all() and pl["y"].between(-10, 60).all()
    first = open(out + "_predlocs.txt").readline().strip()
    assert first == "x,y,sampleID"
    h = pd.read_csv(out + "_history.txt", sep="\t")
    assert list(h.columns) == ["loss", "val_loss", "learning_rate"] and 20 <= len(h) <= 150
    assert h["loss"].iloc[-1] < 0.5 * h["loss"].iloc[0]
    assert json.load(open(out + "_params.json"))["seed"] == 12345
    txt = capsys.readouterr().out
    assert "predicting locations..." in txt and "R2(x)=" in txt and "mean validation error" in txt
    r2x = float(txt.split("R2(x)=")[1].split("\n")[0])
    r2y = float(txt.split("R2(y)=")[1].split("\n")[0])
    err = float(txt.split("mean validation error ")[1].split("\n")[0])
    # README.md:147-155 (unseeded reference run): R2 0.948 / 0.960, mean error 3.76
    assert r2x > 0.85 and r2y > 0.85 and err < 8.0, (r2x, r2y, err)
    # truth for the 50 NA samples is not in the sample file; check the predictions are not collapsed
    assert pl["x"].std() > 5 and pl["y"].std() > 5


def test_no_chain_flag_runs_the_unchained_schedule_to_the_same_fit(tmp_path, capsys):
    """--no_chain (one layer-1 forward launch per step) against the chained default on the reference's example data: the
    same permutations and masks, fp32 round-off apart - ten epochs of predictions within 1e-3 of the coordinate span."""
    outs = []
    for flags in ([], ["--no_chain"]):
        out = str(tmp_path / ("c" + str(len(outs))))
        _run(["--vcf", VCF, "--sample_data", SAMPLES, "--out", out, "--seed", "12345", "--max_epochs", "10",
              "--patience", "10", "--keras_verbose", "0", "--plot_history", ""] + flags)
        outs.append(out)
    a, b = (pd.read_csv(o + "_predlocs.txt") for o in outs)
    assert list(a["sampleID"]) == list(b["sampleID"])
    assert np.abs(a[["x", "y"]].to_numpy() - b[["x", "y"]].to_numpy()).max() < 0.05          # coordinates span 0..50
    ha, hb = (pd.read_csv(o + "_history.txt", sep="\t") for o in outs)
    assert len(ha) == len(hb) == 10 and np.abs(ha["loss"] - hb["loss"]).max() < 1e-3
    assert json.load(open(outs[1] + "_params.json"))["no_chain"] is True


def test_bootstrap_outputs_and_run_to_run_determinism(tmp_path):
    outs = []
    for rep in range(2):
        out = str(tmp_path / f"b{rep}")
        _run(["--vcf", VCF, "--sample_data", SAMPLES, "--out", out, "--seed", "54321", "--bootstrap", "--nboots", "2",
              "--max_epochs", "4", "--patience", "4", "--keras_verbose", "0", "--plot_history", ""])
        outs.append(out)
    for b in ("FULL", "0", "1"):
        f0, f1 = (o + f"_boot{b}_predlocs.txt" for o in outs)
        assert os.path.exists(f0) and open(f0).read() == open(f1).read(), b
    a = pd.read_csv(outs[0] + "_bootFULL_predlocs.txt")
    b = pd.read_csv(outs[0] + "_boot0_predlocs.txt")
    assert len(a) == len(b) == 50 and not np.allclose(a["x"], b["x"])
    # history.txt holds the LAST replicate's history (every replicate overwrites it in the reference)
    h = pd.read_csv(outs[0] + "_history.txt", sep="\t")
    assert len(h) == 4 and not os.path.exists(outs[0] + "_fitplot.pdf")


def test_windows_on_zarr_with_reference_file_naming(tmp_path):
    v = G.read_vcf(VCF)
    store = str(tmp_path / "fix.zarr")
    # blosc / lz4 / shuffle chunks: what `allel.vcf_to_zarr` (scripts/vcf_to_zarr.py:12) writes by default
    G.write_callset_zarr(store, v["calldata/GT"], v["variants/POS"], v["samples"], chunk_variants=4096,
                         compressor="blosc")
    out = str(tmp_path / "w")
    _run(["--zarr", store, "--sample_data", SAMPLES, "--out", out, "--seed", "12345", "--windows",
          "--window_size", "1250000", "--max_epochs", "3", "--patience", "3", "--keras_verbose", "0"])
    size = 1250000
    for i in (0, size):
        stem = f"{out}_{i}-{i + size - 1}"
        # the reference appends the *flag* window to the already window-named stem
        assert os.path.exists(f"{stem}_0-{size - 1}_predlocs.txt"), os.listdir(tmp_path)
        assert len(pd.read_csv(f"{stem}_history.txt", sep="\t")) == 3
    assert os.path.exists(out + "_fitplot.pdf") and os.path.exists(out + "_params.json")


def test_jacknife_keep_weights_and_matrix_input(tmp_path):
    v = G.read_vcf(VCF)
    ac = G.filter_snps(v["calldata/GT"][:3000], 2, verbose=False)[:400]
    mat = str(tmp_path / "m.txt")
    df = pd.DataFrame(ac.T, columns=[f"s{i}" for i in range(ac.shape[0])])
    df.insert(0, "sampleID", v["samples"])
    df.to_csv(mat, sep="\t", index=False)
    out = str(tmp_path / "j")
    _run(["--matrix", mat, "--sample_data", SAMPLES, "--out", out, "--seed", "1", "--jacknife", "--nboots", "3",
          "--max_epochs", "3", "--patience", "3", "--keras_verbose", "0", "--keep_weights", "--width", "64",
          "--nlayers", "4", "--min_mac", "1"])
    for b in ("FULL", "0", "1", "2"):
        assert len(pd.read_csv(f"{out}_boot{b}_predlocs.txt")) == 50
    w = np.load(out + "_bootFULL.weights.npz")
    K = w["gamma"].shape[0]
    assert w["dense_0_kernel"].shape == (K, 64) and w["dense_3_kernel"].shape == (64, 64)
    assert w["dense_4_kernel"].shape == (64, 2) and w["dense_5_kernel"].shape == (2, 2)
    assert w["moving_variance"].shape == (K,)
    full = pd.read_csv(f"{out}_bootFULL_predlocs.txt")
    jk = pd.read_csv(f"{out}_boot0_predlocs.txt")
    assert not np.allclose(full["x"], jk["x"])          # 5 % of SNPs redrawn


def test_batch_size_64_and_rejected_batch_sizes(tmp_path, capsys):
    """--batch_size above the reference default: 64 rows per step (7 steps per epoch on the 405 training
    samples, last batch of 21) trains to the same neighbourhood; 200 (above the row-block kernels' 128: accepted since
    round 3, three steps per epoch) runs; 4097 is rejected with the limit in the message."""
    out = str(tmp_path / "b64")
    _run(["--vcf", VCF, "--sample_data", SAMPLES, "--out", out, "--seed", "12345", "--max_epochs", "200",
          "--patience", "30", "--keras_verbose", "0", "--batch_size", "64"])
    h = pd.read_csv(out + "_history.txt", sep="\t")
    assert 30 <= len(h) <= 200 and h["loss"].iloc[-1] < 0.5 * h["loss"].iloc[0]
    assert json.load(open(out + "_params.json"))["batch_size"] == 64
    txt = capsys.readouterr().out
    r2x = float(txt.split("R2(x)=")[1].split("\n")[0])
    err = float(txt.split("mean validation error ")[1].split("\n")[0])
    assert r2x > 0.85 and err < 8.0, (r2x, err)
    out = str(tmp_path / "b200")
    _run(["--vcf", VCF, "--sample_data", SAMPLES, "--out", out, "--seed", "12345", "--max_epochs", "120",
          "--patience", "30", "--keras_verbose", "0", "--batch_size", "200"])
    h = pd.read_csv(out + "_history.txt", sep="\t")
    assert 30 <= len(h) <= 120 and h["loss"].iloc[-1] < 0.6 * h["loss"].iloc[0]
    with pytest.raises(ValueError, match=r"1\.\.4096"):
        _run(["--vcf", VCF, "--sample_data", SAMPLES, "--out", str(tmp_path / "b4097"), "--seed", "1",
              "--max_epochs", "2", "--keras_verbose", "0", "--batch_size", "4097"])


def test_jacknife_is_one_batched_predict_and_matches_the_oracle_on_the_same_perturbed_matrices(tmp_path, monkeypatch):
    """--jacknife: the nboots perturbed copies of the prediction genotypes go through ONE many-row predict (here
    45 x 50 = 2250 rows: the image + GEMM first layer).  The draws are recorded as the CLI makes them, the perturbed
    matrices rebuilt on the host, and every {out}_boot{b}_predlocs.txt compared with oracle.predict on the weights
    the run kept (2e-5 on z-scored outputs, scaled to map units)."""
    from oracle import locator_oracle as O
    v = G.read_vcf(VCF)
    ac = G.filter_snps(v["calldata/GT"][:6000], 2, verbose=False)[:1500]
    mat = str(tmp_path / "m.txt")
    df = pd.DataFrame(ac.T, columns=[f"s{i}" for i in range(ac.shape[0])])
    df.insert(0, "sampleID", v["samples"])
    df.to_csv(mat, sep="\t", index=False)
    recorded = {}
    real_draws, real_predict = L.jacknife_draws, L.Model.predict

    def spy_draws(predgen, af, nboots, prop):
        recorded["predgen"] = np.array(predgen)
        recorded["draws"] = real_draws(predgen, af, nboots, prop)
        return recorded["draws"]

    def spy_predict(self, gen):
        recorded.setdefault("rows", []).append(gen.shape[0])
        return real_predict(self, gen)
    monkeypatch.setattr(L, "jacknife_draws", spy_draws)
    monkeypatch.setattr(L.Model, "predict", spy_predict)
    out = str(tmp_path / "j")
    nboots = 45
    _run(["--matrix", mat, "--sample_data", SAMPLES, "--out", out, "--seed", "7", "--jacknife", "--nboots", str(nboots),
          "--max_epochs", "3", "--patience", "3", "--keras_verbose", "0", "--keep_weights", "--min_mac", "1",
          "--plot_history", "", "--predict_mode", "exact"])        # the 2e-5 bar below is the exact mode's (auto: 1e-3 relative,
    #                                                                 tests/test_gpu_trained_predict.py)
    assert recorded["rows"][-1] == nboots * 50                  # one predict for all replicates
    p = O.cast_params(L.read_weights(out + "_bootFULL.weights.npz"), np.float64)
    sd = pd.read_csv(SAMPLES, sep="\t")
    mx, sx, my, sy = sd["x"].mean(), sd["x"].std(ddof=0), sd["y"].mean(), sd["y"].std(ddof=0)
    for b in (0, 17, nboots - 1):
        sites, vals = recorded["draws"][b]
        pg = recorded["predgen"].copy()
        pg[:, sites] = vals.T
        z = O.predict(p, pg)
        got = pd.read_csv(f"{out}_boot{b}_predlocs.txt")
        assert np.abs(got["x"].to_numpy() - (z[:, 0] * sx + mx)).max() < 2e-5 * sx + 1e-9
        assert np.abs(got["y"].to_numpy() - (z[:, 1] * sy + my)).max() < 2e-5 * sy + 1e-9


def test_load_weights_predicts_without_training_and_edited_callbacks_change_the_fit(tmp_path, monkeypatch):
    """--keep_weights then --load_weights: the second run trains nothing and writes the same predictions.  And the
    dicts load_callbacks returns are what train_network consumes: a shorter earlystop patience ends the fit earlier."""
    a = str(tmp_path / "a")
    common = ["--vcf", VCF, "--sample_data", SAMPLES, "--seed", "12345", "--keras_verbose", "0", "--plot_history", ""]
    _run(common + ["--out", a, "--max_epochs", "40", "--patience", "25", "--keep_weights"])
    n_a = len(pd.read_csv(a + "_history.txt", sep="\t"))
    b = str(tmp_path / "b")
    _run(common + ["--out", b, "--load_weights", a + ".weights.npz"])
    assert open(a + "_predlocs.txt").read() == open(b + "_predlocs.txt").read()
    assert not os.path.exists(b + "_history.txt")
    real = L.load_callbacks

    def short_patience(boot):
        ck, es, rl = real(boot)
        es["patience"] = 2
        return ck, es, rl
    monkeypatch.setattr(L, "load_callbacks", short_patience)
    c = str(tmp_path / "c")
    _run(common + ["--out", c, "--max_epochs", "40", "--patience", "25"])
    n_c = len(pd.read_csv(c + "_history.txt", sep="\t"))
    h = pd.read_csv(c + "_history.txt", sep="\t")
    assert n_c < n_a and n_c - 1 - int(h["val_loss"].idxmin()) == 2


def _metrics(txt):
    g = lambda key: float(txt.split(key)[1].split("\n")[0])
    return g("R2(x)="), g("R2(y)="), g("mean validation error "), g("median validation error ")


def test_full_default_run_lands_in_the_readme_band(tmp_path, capsys):
    """BASELINE.json configs[1]: the reference's example data with every default (patience 100, max_epochs 5000),
    --seed 12345.  README.md:147-155 prints R2 0.948 / 0.960 and mean error 3.76 for an unseeded reference run; the
    fp32 oracle's fits on the same split give R2 0.96-0.975 and 2.97-3.36 (tests/golden/oracle_fixture_fits.json)."""
    _run(["--vcf", VCF, "--sample_data", SAMPLES, "--out", str(tmp_path / "d"), "--seed", "12345",
          "--keras_verbose", "0", "--plot_history", ""])
    r2x, r2y, mean_err, med_err = _metrics(capsys.readouterr().out)
    assert r2x >= 0.93 and r2y >= 0.93 and mean_err <= 4.5, (r2x, r2y, mean_err)
    h = pd.read_csv(str(tmp_path / "d") + "_history.txt", sep="\t")
    assert 120 <= len(h) <= 600 and h["learning_rate"].iloc[-1] < 1e-3         # early stopping and the LR plateau acted


def test_statistical_parity_with_the_oracle_fit_distribution(tmp_path, capsys):
    """whole fits cannot be compared bit for bit (the reference never seeds TensorFlow), their
    DISTRIBUTION over seeds can.  Six fp32-oracle fits of the default run are committed
    (tests/golden/oracle_fixture_fits.json, make_statistical.py); five HIP fits with different --net_seed values must
    look like draws from the same population: mean validation error within 2.5 pooled standard errors of the oracle
    mean, every run inside the oracle range widened by 3 sigma, R2 and epoch counts in the oracle's neighbourhood."""
    ora = json.load(open(os.path.join(GOLD, "oracle_fixture_fits.json")))["fits"]
    o_err = np.array([f["mean_err"] for f in ora])
    o_r2 = np.array([[f["r2_x"], f["r2_y"]] for f in ora])
    o_ep = np.array([f["epochs"] for f in ora])
    runs = []
    for s in (11, 22, 33, 44, 55):
        out = str(tmp_path / f"s{s}")
        _run(["--vcf", VCF, "--sample_data", SAMPLES, "--out", out, "--seed", "12345", "--net_seed", str(s),
              "--keras_verbose", "0", "--plot_history", ""])
        r2x, r2y, mean_err, _ = _metrics(capsys.readouterr().out)
        runs.append((mean_err, r2x, r2y, len(pd.read_csv(out + "_history.txt", sep="\t"))))
    runs = np.array(runs)
    sig = max(o_err.std(ddof=1), 0.05)
    se = sig * np.sqrt(1 / len(o_err) + 1 / len(runs))
    assert abs(runs[:, 0].mean() - o_err.mean()) < 2.5 * se + 0.05, (runs[:, 0], o_err)
    assert runs[:, 0].min() > o_err.min() - 3 * sig and runs[:, 0].max() < o_err.max() + 3 * sig, (runs[:, 0], o_err)
    assert runs[:, 1:3].min() > o_r2.min() - 0.03, (runs[:, 1:3], o_r2)
    assert 0.5 * o_ep.min() <= np.median(runs[:, 3]) <= 2 * o_ep.max(), (runs[:, 3], o_ep)
    assert len(set(np.round(runs[:, 0], 6))) == 5                       # the net seed really changes the fit


def test_config3_windows_end_to_end_at_ag1000g_window_size(tmp_path, capsys):
    """BASELINE.json configs[3], two of its windows at full width: a synthetic `allel.vcf_to_zarr`-shaped store of 765
    samples x 2 windows x 150,000 variants (2 Mb each), `--windows --window_size 2000000`, callback-driven fits (two
    worker processes on the GPU).  Every window writes its predlocs / history under the reference's names, and the
    76 samples without coordinates are placed near where the generator put them (coordinates span 0..50)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_synth_zarr", os.path.join(os.path.dirname(GOLD), "..", "tools",
                                                                                  "make_synth_zarr.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    stem = str(tmp_path / "c3")
    xy, na = mk.make_store(stem, n=765, windows=2, window_size=2_000_000, per_window=150_000, seed=5)
    out = str(tmp_path / "win")
    _run(["--zarr", stem + ".zarr", "--sample_data", stem + "_samples.txt", "--out", out, "--windows",
          "--window_size", "2000000", "--seed", "12345", "--keras_verbose", "0", "--plot_history", ""])
    size = 2_000_000
    ids = np.array([f"AB{i:04d}" for i in range(765)])
    for w in (0, 1):
        stem_w = f"{out}_{w * size}-{(w + 1) * size - 1}"
        pl = pd.read_csv(f"{stem_w}_0-{size - 1}_predlocs.txt")            # the reference's doubled window name
        h = pd.read_csv(f"{stem_w}_history.txt", sep="\t")
        assert list(pl["sampleID"]) == list(ids[na]) and 20 <= len(h) <= 2000
        err = np.sqrt((pl["x"].to_numpy() - xy[na, 0]) ** 2 + (pl["y"].to_numpy() - xy[na, 1]) ** 2)
        assert err.mean() < 6.0, err.mean()                                 # full-size run: 2.7-3.8 per window


def test_config4_bootstrap_end_to_end_at_500k_snps(tmp_path):
    """BASELINE.json configs[4] at its matrix width, with few replicates and a bounded epoch count: 1000 samples x
    520,000 variants (> 500,000 SNPs survive the filters), `--bootstrap --nboots 3` (FULL + 3 resampled fits, two
    worker processes on the GPU, genotype rows handed over through shared memory, column resampling on the device).
    Every replicate writes its predlocs; resampled replicates differ from the full fit; the 100 samples without
    coordinates land near the truth already after 30 epochs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_synth_zarr", os.path.join(os.path.dirname(GOLD), "..", "tools",
                                                                                  "make_synth_zarr.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    stem = str(tmp_path / "c4")
    xy, na = mk.make_store(stem, n=1000, windows=1, window_size=50_000_000, per_window=520_000, seed=9)
    out = str(tmp_path / "boot")
    _run(["--zarr", stem + ".zarr", "--sample_data", stem + "_samples.txt", "--out", out, "--bootstrap", "--nboots", "3",
          "--seed", "12345", "--max_epochs", "30", "--patience", "30", "--keras_verbose", "0", "--plot_history", ""])
    preds = {b: pd.read_csv(f"{out}_boot{b}_predlocs.txt") for b in ("FULL", "0", "1", "2")}
    for b, pl in preds.items():
        assert len(pl) == 100
        err = np.sqrt((pl["x"].to_numpy() - xy[na, 0]) ** 2 + (pl["y"].to_numpy() - xy[na, 1]) ** 2)
        assert err.mean() < 8.0, (b, err.mean())
    assert not np.allclose(preds["FULL"]["x"], preds["0"]["x"]) and not np.allclose(preds["0"]["x"], preds["1"]["x"])
    assert len(pd.read_csv(out + "_history.txt", sep="\t")) == 30
