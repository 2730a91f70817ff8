"""Replicate summarisation on the device: loc_kde_peak_batch through the C ABI against the NumPy form of
locator_amd/summarize.py and against sklearn's KernelDensity - the estimator the reference itself calls
(/root/reference/locator_py/plot_locator.py:26-44)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host_scores(x, y, h):
    d2 = (x[:, None] - x[None, :]) ** 2 + (y[:, None] - y[None, :]) ** 2
    return np.log(np.exp(-d2 / (2.0 * h * h)).sum(axis=1))


def _bootstrap_like(rng, n_samples, n_rep, spread=0.3, outliers=True):
    truth = rng.uniform(0, 50, (n_samples, 2))
    P = truth[:, None, :] + rng.normal(0, spread, (n_samples, n_rep, 2))
    if outliers:                                   # a few replicates far from the rest, as failed fits produce
        for s in range(0, n_samples, 7):
            P[s, rng.integers(0, n_rep, 3)] += rng.normal(0, 20.0, (3, 2))
    return P


@pytest.mark.parametrize("n_samples,n_rep", [(100, 257), (3, 12), (5, 1), (2, 5000)])
def test_kde_peak_batch_equals_the_numpy_form_and_sklearn(n_samples, n_rep):
    """257 bootstrap predlocs x 100 samples (configs[4]'s --nboots 256 + the full fit), a short run, single replicates, and a
    sample beyond the kernel's LDS staging (5000 > 4096 points: read through L1).  Index-exact wherever the two best scores
    differ by more than float64 round-off; centroids to 1e-12 relative (NumPy sums pairwise, the kernel in a fixed tree)."""
    from sklearn.neighbors import KernelDensity
    from locator_amd import summarize as S
    rng = np.random.default_rng(7 + n_rep)
    P = _bootstrap_like(rng, n_samples, n_rep)
    idx, out = S.device_summaries([(P[s, :, 0], P[s, :, 1]) for s in range(n_samples)], 0.2)
    assert idx.shape == (n_samples,) and out.shape == (n_samples, 4)
    checked = 0
    for s in range(n_samples):
        x, y = P[s, :, 0], P[s, :, 1]
        sc = _host_scores(x, y, 0.2)
        order = np.argsort(-sc, kind="stable")
        gap = sc[order[0]] - sc[order[1]] if n_rep > 1 else np.inf
        if gap > 1e-9:
            assert idx[s] == order[0], (s, idx[s], order[:3], gap)
            kx, ky = S.kde_peak(x, y, 0.2)
            assert (out[s, 0], out[s, 1]) == (kx, ky)                      # the very point, bit for bit
            checked += 1
        else:                                                               # a near-tie: either of the tied points
            assert sc[idx[s]] >= sc[order[0]] - 1e-9
        assert (out[s, 0], out[s, 1]) == (x[idx[s]], y[idx[s]])
        gx, gy = S.centroid(x, y)
        assert abs(out[s, 2] - gx) <= 1e-12 * max(1.0, abs(gx)) and abs(out[s, 3] - gy) <= 1e-12 * max(1.0, abs(gy))
    assert checked >= n_samples - 2
    for s in range(min(n_samples, 4)):                                      # the reference's own estimator picks the same point
        if n_rep > 2000:
            break
        kd = KernelDensity(kernel="gaussian", bandwidth=0.2).fit(P[s])
        e = kd.score_samples(P[s])
        assert int(np.argwhere(e == np.amax(e)).tolist()[0][0]) == idx[s]


def test_kde_peak_ties_take_the_first_maximum_and_bad_samples_fall_back_to_the_mean():
    """np.argwhere(e == max)[0] is the reference's tie-break: an all-equal sample -> index 0; exact duplicate points -> the
    earlier one.  A non-finite coordinate has no density estimate (sklearn raises, plot_locator.py:35-37): index -1 and the
    mean, NaN included, as np.mean gives it.  Ragged replicate counts in one launch, an empty sample among them."""
    from locator_amd import summarize as S
    rng = np.random.default_rng(3)
    same = (np.full(64, 12.5), np.full(64, -3.25))
    dup_x = np.concatenate([rng.normal(10, 0.05, 40), [30.0, 30.0, 30.0]])
    dup_y = np.concatenate([rng.normal(5, 0.05, 40), [7.0, 7.0, 7.0]])
    dup_x[[5, 17]], dup_y[[5, 17]] = dup_x[2], dup_y[2]                     # three exact copies inside the dense cluster
    bad_x, bad_y = rng.normal(0, 1, 20), rng.normal(0, 1, 20)
    bad_y[4] = np.nan
    inf_x, inf_y = rng.normal(0, 1, 9), rng.normal(0, 1, 9)
    inf_x[0] = np.inf
    empty = (np.zeros(0), np.zeros(0))
    ragged = (rng.normal(0, 1, 300), rng.normal(0, 1, 300))
    groups = [same, (dup_x, dup_y), (bad_x, bad_y), empty, ragged, (inf_x, inf_y)]
    idx, out = S.device_summaries(groups, 0.2)
    assert idx[0] == 0 and (out[0, 0], out[0, 1]) == (12.5, -3.25) and (out[0, 2], out[0, 3]) == (12.5, -3.25)
    sc = _host_scores(dup_x, dup_y, 0.2)
    best = int(np.argmax(sc))
    assert idx[1] == best
    if best in (2, 5, 17):
        assert idx[1] == 2                                                   # the first of the identical points
    assert idx[2] == -1 and np.isnan(out[2, 1]) and np.isnan(out[2, 3]) and out[2, 0] == pytest.approx(np.mean(bad_x), rel=1e-12)
    assert idx[3] == -1 and np.isnan(out[3]).all()
    kx, ky = S.kde_peak(*ragged, 0.2)
    assert (out[4, 0], out[4, 1]) == (kx, ky)
    assert idx[5] == -1 and out[5, 0] == np.inf and out[5, 2] == np.inf


def test_summarize_cli_uses_the_device_and_writes_the_references_columns(tmp_path):
    """End to end: 70 bootstrap predlocs files of 30 samples -> {out}_centroids.txt, device launch = --host form."""
    import pandas as pd
    from locator_amd import summarize as S
    rng = np.random.default_rng(11)
    ids = [f"s{i}" for i in range(30)]
    P = _bootstrap_like(rng, 30, 70)
    for b in range(70):
        pd.DataFrame({"x": P[:, b, 0], "y": P[:, b, 1], "sampleID": ids}).to_csv(tmp_path / f"run_boot{b}_predlocs.txt", index=False)
    assert S.main(["--infile", str(tmp_path), "--out", str(tmp_path / "dev"), "--silence"]) == 0
    assert S.main(["--infile", str(tmp_path), "--out", str(tmp_path / "host"), "--silence", "--host"]) == 0
    d = pd.read_csv(str(tmp_path / "dev") + "_centroids.txt", sep="\t")
    h = pd.read_csv(str(tmp_path / "host") + "_centroids.txt", sep="\t")
    assert list(d.columns) == ["sampleID", "x", "y", "kd_x", "kd_y", "gc_x", "gc_y"]
    assert (d.sampleID == h.sampleID).all()
    assert np.array_equal(d[["kd_x", "kd_y"]].to_numpy(), h[["kd_x", "kd_y"]].to_numpy())
    assert np.allclose(d[["gc_x", "gc_y"]].to_numpy(), h[["gc_x", "gc_y"]].to_numpy(), rtol=1e-12, atol=0)
