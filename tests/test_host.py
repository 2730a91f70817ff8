"""Host logic that needs no GPU: ingest + filters against the reference's fixture, flag surface,
RNG-order parity of the replicate plans, and replicate sharding (incl. 2 ranks over gloo)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from locator_amd import genotypes as G
from locator_amd import locator as L
from locator_amd import replicates as R
from oracle import locator_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
VCF = os.path.join(GOLD, "test_genotypes.vcf.gz")
SAMPLES = os.path.join(GOLD, "test_sample_data.txt")


@pytest.fixture(scope="module")
def fixture_vcf():
    return G.read_vcf(VCF)


# ------------------------------------------------------------------ ingest
def test_vcf_fixture_shape_and_filters(fixture_vcf):
    """SURVEY.md §4 fixture characterisation: 11,527 records x 500 samples, allelism {1:5055, 2:6467, 3:5},
    K = 5,830 after biallelic + allele-1 count >= 2, values {0: 78.0 %, 1: 11.2 %, 2: 10.9 %}."""
    gt = fixture_vcf["calldata/GT"]
    assert gt.shape == (11527, 500, 2) and gt.dtype == np.int8
    assert list(fixture_vcf["samples"][:2]) == ["msp_0", "msp_1"] and fixture_vcf["samples"][-1] == "msp_499"
    pos = fixture_vcf["variants/POS"]
    assert pos[0] == 197 and pos[-1] == 2499926 and np.all(np.diff(pos) > 0)
    assert gt.min() == 0 and gt.max() == 2
    ac = G.count_alleles(gt)
    assert np.bincount((ac > 0).sum(1)).tolist() == [0, 5055, 6467, 5]
    a = G.filter_snps(gt, 2, verbose=False)
    assert a.shape == (5830, 500) and a.dtype == np.int8
    frac = np.bincount(a.ravel()) / a.size
    assert np.allclose(frac, [0.7775, 0.1116, 0.1109], atol=1e-4)     # SURVEY: 78.0 / 11.2 / 10.9 % (rounded)
    assert G.filter_snps(gt, 1, verbose=False).shape[0] == 6467          # min_mac == 1 skips the count filter


def test_vcf_parser_general_calls(tmp_path):
    p = tmp_path / "t.vcf"
    p.write_text("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2\ts3\n"
                 "1\t10\t.\tA\tT\t.\tPASS\t.\tGT\t0|1\t./.\t1/1\n"
                 "1\t20\t.\tA\tT,G\t.\tPASS\t.\tGT:DP\t0/2:5\t1|.:3\t10/1:9\n")
    v = G.read_vcf(str(p))
    assert v["calldata/GT"].tolist() == [[[0, 1], [-1, -1], [1, 1]], [[0, 2], [1, -1], [10, 1]]]
    assert v["variants/POS"].tolist() == [10, 20] and list(v["samples"]) == ["s1", "s2", "s3"]
    gt = v["calldata/GT"]
    assert G.is_missing(gt).tolist() == [[False, True, False], [False, True, False]]
    assert G.to_allele_counts_1(gt).tolist() == [[1, 0, 2], [0, 1, 1]]
    assert G.count_alleles(gt, 2).tolist() == [[1, 3, 0], [1, 2, 1]]


def test_impute_missing_consumes_rng_in_site_major_order():
    rng = np.random.default_rng(0)
    gt = rng.integers(0, 2, (6, 5, 2)).astype(np.int8)
    gt[1, 3] = -1
    gt[4, 0] = -1
    gt[4, 2, 1] = -1
    np.random.seed(7)
    ac = G.replace_md(gt.copy())
    dc = (gt == 1).reshape(6, -1).sum(1)
    nind = (~(gt < 0).any(2)).sum(1)
    np.random.seed(7)
    exp = [(1, 3, np.random.binomial(2, dc[1] / (2 * nind[1]))), (4, 0, np.random.binomial(2, dc[4] / (2 * nind[4]))),
           (4, 2, np.random.binomial(2, dc[4] / (2 * nind[4])))]
    for i, j, v in exp:
        assert ac[i, j] == v
    untouched = ~(gt < 0).any(2)
    assert np.array_equal(ac[untouched], (gt == 1).sum(2)[untouched])


@pytest.mark.parametrize("compressor", [None, "zlib"])
def test_zarr_v2_roundtrip_and_window_slices(tmp_path, compressor):
    rng = np.random.default_rng(1)
    gt = rng.integers(0, 2, (1000, 17, 2)).astype(np.int8)
    pos = np.sort(rng.choice(10**6, 1000, replace=False)).astype(np.int32)
    samples = np.array([f"s{i}" for i in range(17)])
    store = str(tmp_path / "c.zarr")
    G.write_callset_zarr(store, gt, pos, samples, chunk_variants=128, compressor=compressor)
    cs = G.open_group(store, mode="r")
    z = cs["calldata/GT"]
    assert z.shape == (1000, 17, 2)
    assert np.array_equal(z[:], gt) and np.array_equal(z[100:777, :, :], gt[100:777])
    assert np.array_equal(z[127:129], gt[127:129]) and z[5:5].shape == (0, 17, 2)
    assert np.array_equal(np.array(cs["variants/POS"]), pos)
    assert list(np.asarray(cs["samples"][:]).astype(str)) == list(samples)


def test_zarr_vlen_utf8_samples_and_unsupported_compressor(tmp_path):
    import struct
    d = tmp_path / "s"
    d.mkdir()
    names = ["AB0001", "x", "sample-3"]
    raw = struct.pack("<I", 3) + b"".join(struct.pack("<I", len(n)) + n.encode() for n in names)
    (d / "0").write_bytes(raw)
    (d / ".zarray").write_text(json.dumps({"zarr_format": 2, "shape": [3], "chunks": [3], "dtype": "|O",
                                           "compressor": None, "fill_value": 0, "order": "C",
                                           "filters": [{"id": "vlen-utf8"}]}))
    assert list(G.ZarrArray(str(d))[:]) == names
    (d / ".zarray").write_text(json.dumps({"zarr_format": 2, "shape": [3], "chunks": [3], "dtype": "<i4",
                                           "compressor": {"id": "lzma"}, "fill_value": 0,
                                           "order": "C", "filters": None}))
    with pytest.raises(ValueError, match="lzma.*not supported"):
        G.ZarrArray(str(d))


def test_matrix_reader_equals_count_semantics(tmp_path):
    p = tmp_path / "m.txt"
    p.write_text("sampleID\ta\tb\tc\nmsp1\t0\t1\t2\nmsp2\t2\t0\t1\n")
    gt, samples = G.read_matrix(str(p))
    assert list(samples) == ["msp1", "msp2"] and gt.shape == (3, 2, 2)
    assert G.to_allele_counts_1(gt).T.tolist() == [[0, 1, 2], [2, 0, 1]]


# ------------------------------------------------------------------ CLI surface
REFERENCE_FLAGS = {   # name: default — /root/reference/locator/locator.py:13-166, in declaration order
    "vcf": None, "zarr": None, "matrix": None, "sample_data": None, "train_split": 0.9, "windows": False,
    "window_start": 0, "window_stop": None, "window_size": 5e5, "bootstrap": False, "jacknife": False,
    "jacknife_prop": 0.05, "nboots": 50, "batch_size": 32, "max_epochs": 5000, "patience": 100, "min_mac": 2,
    "max_SNPs": None, "impute_missing": False, "dropout_prop": 0.25, "nlayers": 10, "width": 256, "out": None,
    "seed": None, "gpu_number": None, "plot_history": True, "keep_weights": False, "load_params": None,
    "keras_verbose": 1}


def test_flag_surface_matches_reference_and_params_json(tmp_path):
    ns = vars(L.build_parser().parse_args([]))
    keys = list(ns)
    assert keys[:len(REFERENCE_FLAGS)] == list(REFERENCE_FLAGS)          # same names, same order
    for k, v in REFERENCE_FLAGS.items():
        assert ns[k] == v, k
    # window flags stay untyped strings on the command line; plot_history is type=bool (Q5)
    ns2 = L.build_parser().parse_args(["--window_size", "2000000", "--plot_history", "False"])
    assert ns2.window_size == "2000000" and ns2.plot_history is True
    out = str(tmp_path / "run")
    a = L._setup(["--vcf", "x.vcf", "--sample_data", "s.txt", "--out", out, "--seed", "12345"])
    js = json.load(open(out + "_params.json"))
    assert list(js)[:len(REFERENCE_FLAGS)] == list(REFERENCE_FLAGS) and js["seed"] == 12345 and js["out"] == out
    # --load_params replaces every argument, including out; a reference-written json loads too
    ref_like = {k: js[k] for k in REFERENCE_FLAGS}
    ref_like["out"] = str(tmp_path / "other")
    json.dump(ref_like, open(str(tmp_path / "p.json"), "w"))
    b = L._setup(["--out", out, "--load_params", str(tmp_path / "p.json")])
    assert b.out == ref_like["out"] and b.gpus is None and os.path.exists(ref_like["out"] + "_params.json")


def test_pipeline_rng_order_matches_reference_known_answers(fixture_vcf, tmp_path):
    """sort_samples -> normalize_locs -> filter_snps -> split_train_test on the fixture with --seed 12345
    reproduces SURVEY.md §4's known answers (validation indices, bootstrap reseed, site order)."""
    L._setup(["--vcf", VCF, "--sample_data", SAMPLES, "--out", str(tmp_path / "o"), "--seed", "12345",
              "--bootstrap", "--nboots", "2"])
    gt, samples = fixture_vcf["calldata/GT"], fixture_vcf["samples"]
    sd, locs = L.sort_samples(samples, gt)
    assert np.isnan(locs[:50]).all() and not np.isnan(locs[50:]).any()
    ml, sl, ma, sa, nlocs = L.normalize_locs(locs)
    assert abs(sl - 14.3074) < 1e-3 and abs(sa - 14.0966) < 1e-3
    oml, osl, oma, osa, olocs = O.normalize_locs(locs)
    assert (ml, sl, ma, sa) == (oml, osl, oma, osa) and np.allclose(nlocs[50:], olocs[50:], rtol=0, atol=1e-15)
    ac = L.filter_snps(gt)
    train, test, tg, vg, tl, vl, pred, pg = L.split_train_test(ac, nlocs)
    assert list(test[:10]) == [465, 459, 233, 149, 429, 423, 454, 140, 489, 165]
    assert hashlib.sha1(test.astype("int64").tobytes()).hexdigest()[:16] == "144d3567059195b1"
    assert (len(train), len(test), len(pred)) == (405, 45, 50) and tg.shape == (405, 5830)
    units = L._bootstrap_units(tg.shape[1])
    assert [u["boot"] for u in units] == ["FULL", 0, 1] and [u["replicate"] for u in units] == [0, 1, 2]
    assert list(units[1]["site_order"][:8]) == [1092, 5266, 5679, 4778, 3551, 2418, 1321, 5729]
    # the oracle's restatement of the same chain agrees
    np.random.seed(12345)
    O.split_train_test(ac, nlocs, 0.9)
    chain = O.bootstrap_chain(2, 5830)
    assert chain[0][0] == 812135 and np.array_equal(chain[1][1], units[2]["site_order"])


# ------------------------------------------------------------------ replicate sharding
def _fake_fit(unit, device="cpu"):
    if unit.get("explode"):
        raise RuntimeError("boom")
    if unit.get("die"):                       # the worker process itself goes away (GPU fault / OOM kill stand-in)
        os.kill(os.getpid(), 9)
    if "big" in unit:                         # an array handed over through shared memory: read-only view, right content
        assert not unit["big"].flags.writeable and unit["big"].shape == (600, 500)
        return {"name": unit["name"], "value": float(unit["big"][unit["replicate"]].sum()), "seconds": 0.0,
                "args_out": unit["args"].out}
    v = float(np.sum(unit["payload"]) + unit["shared_bias"]) * (unit["replicate"] + 1)
    return {"name": unit["name"], "value": v, "seconds": 0.0, "args_out": unit["args"].out}


class _Args:
    out = "stem"


def _slow_fit(unit, device="cpu"):
    """0.4 s of 'fit'; the host part (0.4 s) must already have happened on the loader thread."""
    import time
    assert unit["host_done"], "host_prepare did not run before the fit"
    if unit.get("hang"):
        time.sleep(3600)
    if unit.get("die_after_host"):
        os.kill(os.getpid(), 9)
    t0 = time.time()
    time.sleep(0.4)
    return {"name": unit["name"], "value": unit["replicate"] * 2, "seconds": time.time() - t0, "pid": os.getpid()}


def _slow_host_prepare(unit, args):
    import time
    if unit.get("hang_host"):
        time.sleep(3600)
    time.sleep(0.4)
    return dict(unit, host_done=True, host_pid=os.getpid(), args_seen=args.out)


def _units(n):
    return [dict(name=f"u{i}", replicate=i, payload=np.arange(i + 3)) for i in range(n)]


def test_run_units_sequential_order_and_failure_isolation():
    units = _units(5)
    units[2]["explode"] = True
    logs = []
    res = R.run_units(units, _Args(), _fake_fit, n_gpus=1, shared={"shared_bias": 10.0}, log=logs.append)
    assert [r["unit_index"] for r in res] == [0, 1, 2, 3, 4]
    assert "error" in res[2] and "boom" in res[2]["error"] and any("FAILED" in l for l in logs)
    for i in (0, 1, 3, 4):
        assert res[i]["value"] == (sum(range(i + 3)) + 10.0) * (i + 1) and res[i]["args_out"] == "stem"


def test_run_units_spawned_workers_dynamic_queue_on_cpu():
    """The multi-worker path (spawned processes + bounded task queue + result queue): results come back in
    unit order whatever the completion order, a failing unit is reported and its siblings still run."""
    units = _units(9)
    units[4]["explode"] = True
    logs = []
    res = R.run_units(units, _Args(), _fake_fit, n_gpus=1, shared={"shared_bias": 2.0}, log=logs.append,
                      fits_per_gpu=3)
    assert [r["unit_index"] for r in res] == list(range(9))
    assert "error" in res[4] and sum("FAILED" in l for l in logs) == 1
    for i in range(9):
        if i != 4:
            assert res[i]["value"] == (sum(range(i + 3)) + 2.0) * (i + 1)


def test_run_units_survives_a_dead_worker_and_shares_big_arrays_through_shared_memory():
    """ADVICE r1 / VERDICT r1 #8: a worker killed mid-unit must not hang the run.  The parent notices the dead
    process, reports the unit it was holding as failed, starts a fresh worker and finishes the siblings.  The 2.4 MB
    `big` array travels once through multiprocessing.shared_memory (attached read-only by the workers, unlinked at
    the end)."""
    import glob
    big = np.arange(600 * 500, dtype=np.float64).reshape(600, 500)
    units = [dict(name=f"u{i}", replicate=i, payload=np.arange(2)) for i in range(8)]
    units[3]["die"] = True
    before = set(glob.glob("/dev/shm/psm_*"))
    logs = []
    res = R.run_units(units, _Args(), _fake_fit, n_gpus=1, shared={"shared_bias": 0.0, "big": big}, log=logs.append,
                      fits_per_gpu=2, poll_s=0.2)
    assert [r["unit_index"] for r in res] == list(range(8))
    assert "error" in res[3] and "died" in res[3]["error"]
    for i in range(8):
        if i != 3:
            assert "error" not in res[i] and res[i]["value"] == float(big[i].sum()), res[i]
    assert set(glob.glob("/dev/shm/psm_*")) == before              # nothing left behind in /dev/shm


def test_shard_static_partitions_units():
    for n, w in [(10, 4), (3, 8), (257, 8), (0, 2)]:
        parts = [R.shard_static(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _dist_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = R.run_units_distributed(_units(7), _Args(), _fake_fit, shared={"shared_bias": 1.0}, device="cpu")
    q.put((rank, [(r["unit_index"], r["gpu"], r["value"]) for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_run_units_distributed_gloo_world2_matches_single_process():
    """N > 1 path on CPU: 2 ranks over gloo, round-robin units, results gathered on every rank and equal
    to the 1-process run (per-unit results depend on the replicate index only, not on the rank)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = R.run_units(_units(7), _Args(), _fake_fit, n_gpus=1, shared={"shared_bias": 1.0})
    for rank in (0, 1):
        assert [v for _, _, v in got[rank]] == [r["value"] for r in single]
        assert [g for _, g, _ in got[rank]] == [i % 2 for i in range(7)]


# ------------------------------------------------------------------ replicate summarisation
def test_summarize_centroid_and_kernel_peak(tmp_path):
    import pandas as pd
    from sklearn.neighbors import KernelDensity
    from locator_amd import summarize as S
    rng = np.random.default_rng(0)
    ids = [f"s{i}" for i in range(5)]
    truth = rng.uniform(0, 50, (5, 2))
    preds = []
    for b in range(12):
        p = truth + rng.normal(0, 0.3, (5, 2))
        if b == 3:
            p[0] += 25.0                       # one outlier replicate for s0
        preds.append(p)
        pd.DataFrame({"x": p[:, 0], "y": p[:, 1], "sampleID": ids}).to_csv(tmp_path / f"run_boot{b}_predlocs.txt",
                                                                          index=False)
    sd = tmp_path / "samples.txt"
    pd.DataFrame({"sampleID": ids, "x": truth[:, 0], "y": truth[:, 1]}).to_csv(sd, sep="\t", index=False)
    bp = S.summarize(str(tmp_path), str(sd), str(tmp_path / "out"), silence=True, host=True)
    if not __import__("torch").cuda.is_available():      # the default is the device launch: no silent NumPy fallback
        with pytest.raises(SystemExit, match="no GPU visible"):
            S.summarize(str(tmp_path), str(sd), str(tmp_path / "out2"), silence=True)
    assert list(bp.columns) == ["sampleID", "x", "y", "kd_x", "kd_y", "gc_x", "gc_y"] and len(bp) == 5
    P = np.array(preds)                        # (12, 5, 2)
    assert np.allclose(bp[["gc_x", "gc_y"]].to_numpy(), P.mean(0))
    for i in range(5):                         # same point as sklearn's KernelDensity picks (the reference's estimator)
        kd = KernelDensity(kernel="gaussian", bandwidth=0.2).fit(P[:, i, :])
        j = int(np.argmax(kd.score_samples(P[:, i, :])))
        assert np.allclose(bp.loc[i, ["kd_x", "kd_y"]].to_numpy(dtype=float), P[j, i, :])
    # the density peak ignores the outlier, the centroid does not
    assert np.hypot(bp.kd_x[0] - truth[0, 0], bp.kd_y[0] - truth[0, 1]) < 1.0
    assert np.hypot(bp.gc_x[0] - truth[0, 0], bp.gc_y[0] - truth[0, 1]) > 2.0
    out = pd.read_csv(str(tmp_path / "out") + "_centroids.txt", sep="\t")
    assert list(out.columns) == list(bp.columns)


def test_windows_fast_prologue_keeps_the_rng_stream(tmp_path, fixture_vcf):
    """--windows skips the reference's discarded whole-store load (locator.py:508-516); the per-window splits
    must come out exactly as if it had been done (same global NumPy stream)."""
    gt, pos, samples = fixture_vcf["calldata/GT"][:3000], fixture_vcf["variants/POS"][:3000], fixture_vcf["samples"]
    store = str(tmp_path / "w.zarr")
    G.write_callset_zarr(store, gt, pos, samples, chunk_variants=1024)
    size = int(pos[-1] // 3 + 1)
    argv = ["--zarr", store, "--sample_data", SAMPLES, "--out", str(tmp_path / "o"), "--seed", "777", "--windows",
            "--window_size", str(size)]
    got = []
    for force_full in (True, False):
        L._setup(argv)
        smp, state = L._prologue(force_full=force_full)
        assert (state is None) == (not force_full)
        # eager units (parent slices + filters, the reference's order) vs lazy units (splits drawn up front,
        # slice + filter deferred to the worker)
        units = L._window_units(smp, lazy=not force_full)
        if not force_full:
            assert all("traingen" not in u for u in units)
            units = [L._load_window(dict(u, args=L.args)) for u in units]
        got.append(units)
    assert len(got[0]) == len(got[1]) == 3
    for a, b in zip(*got):
        assert a["name"] == b["name"] and np.array_equal(a["pred"], b["pred"])
        for k in ("traingen", "testgen", "predgen", "trainlocs", "testlocs"):
            assert np.array_equal(a[k], b[k]), k


def test_window_loader_falls_back_to_the_host_path_for_a_store_that_is_not_int8(tmp_path, fixture_vcf, monkeypatch):
    """the GPU worker's loader thread decodes gt[a:b] straight into an int8 pinned view, which only an
    int8 [variants][samples][ploidy] store allows.  A store with another dtype (allel writes int8, other tools may not)
    used to fail EVERY window; now it takes the host slice + filter path, with the same rows as an int8 copy of it."""
    import torch
    gt, pos, samples = fixture_vcf["calldata/GT"][:1500], fixture_vcf["variants/POS"][:1500], fixture_vcf["samples"]
    got = {}
    for name, dt in (("i8", np.int8), ("i16", np.int16)):
        store = str(tmp_path / f"{name}.zarr")
        G.write_callset_zarr(store, gt, pos, samples, chunk_variants=512)
        if dt is not np.int8:                   # rewrite calldata/GT in the other dtype
            import shutil
            shutil.rmtree(os.path.join(store, "calldata", "GT"))
            G.write_zarr_array(os.path.join(store, "calldata", "GT"), np.asarray(gt).astype(dt), (512,) + tuple(np.shape(gt)[1:]))
        L._setup(["--zarr", store, "--sample_data", SAMPLES, "--out", str(tmp_path / name), "--seed", "5", "--windows",
                  "--window_size", str(int(pos[-1] + 1))])
        smp, _ = L._prologue(force_full=False)
        units = L._window_units(smp, lazy=True)
        assert len(units) == 1
        monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
        if dt is np.int8:
            seen = []
            monkeypatch.setattr(L, "_read_window", lambda u: seen.append(u) or u)
            L._load_window_on_loader_thread(units[0], L.args)
            assert len(seen) == 1                                   # an int8 store goes the device way
            monkeypatch.undo()
            got[name] = L._load_window(dict(units[0], args=L.args))
        else:
            monkeypatch.setattr(L, "_read_window", lambda u: (_ for _ in ()).throw(AssertionError("int8-only path taken")))
            got[name] = L._load_window_on_loader_thread(units[0], L.args)
            monkeypatch.undo()
    for k in ("traingen", "testgen", "predgen"):
        assert got["i16"][k].dtype == got["i8"][k].dtype and np.array_equal(got["i8"][k], got["i16"][k]), k


def test_device_lock_is_reentrant_per_thread_and_released_drops_the_whole_hold():
    """train.DEVICE_LOCK (round 5): what keeps a HIP-graph capture apart from a sibling fit thread's set-up / read-back /
    tear-down.  Re-entrant for the thread that holds it; `released()` lets go of every nested hold for the duration of a
    block (FitLoop.run's epoch loop) and takes them back; another thread gets in exactly while it is released."""
    import threading
    import time
    from locator_amd.train import _DeviceLock
    lock = _DeviceLock()
    order = []

    def sibling():
        with lock:
            order.append("sibling in")
            time.sleep(0.05)
            order.append("sibling out")

    with lock:
        with lock:                                  # nested hold (locator._fit_unit -> FitLoop.finish)
            assert lock.held()
            t = threading.Thread(target=sibling)
            t.start()
            time.sleep(0.1)
            assert order == []                      # the sibling waits while this thread holds the lock
            with lock.released():
                assert not lock.held()
                t.join(2)
                assert order == ["sibling in", "sibling out"]
            assert lock.held()                      # ... and both holds are back
        assert lock.held()
    assert not lock.held()
    with lock.released():                           # a no-op for a thread that holds nothing
        assert not lock.held()
    got = []
    t = threading.Thread(target=lambda: (lock.__enter__(), got.append(1), lock.__exit__(None, None, None)))
    t.start()
    t.join(2)
    assert got == [1]


def test_fit_budget_admits_by_each_units_own_snp_count_and_reads_the_flag_from_the_unit(monkeypatch):
    """--fits_per_gpu 0 (the default since round 5): the pool starts three fit threads per GPU and locator._fit_unit admits
    as many units at a time as their SNP counts ask for - 3 up to 70,000 SNPs, 2 above (bench.py --replicates-per-gpu sweep:
    903k against 711k samples/s for 3 / 2 fits at 5,830 SNPs, 221k against 226k at 100,000), one large beside one small;
    an explicit flag wins.  Round 6: the rule is applied PER UNIT (a small first window no longer decides for
    the 150k-variant windows behind it) and the flag is read from the unit's own args - in a spawned worker the module
    global is still None when the first unit asks, so --fits_per_gpu 4 used to admit 3."""
    class A:
        fits_per_gpu = 0
    monkeypatch.setattr(L, "args", None, raising=False)              # what a fresh worker process has
    assert L._snps_hint({"window": (100, 1250)}) == 1150 and L._snps_hint({"gt_shape": (150016, 765, 2)}) == 150016
    assert L._snps_hint({"traingen": np.zeros((5, 77), np.uint8)}) == 77 and L._snps_hint({}) == 0

    def admitted_without_blocking(budget, Ks, a):
        import threading
        got = []
        for K in Ks:
            th = threading.Thread(target=lambda K=K: got.append(budget.acquire(K, a)), daemon=True)
            th.start()
            th.join(0.3)
        return len(got)

    for Ks, flag, want in (([5830] * 5, 0, 3), ([70_000] * 5, 0, 3), ([70_001] * 5, 0, 2), ([560_000] * 4, 0, 2),
                           ([150_000, 1150, 1150], 0, 2), ([1150, 1150, 150_000], 0, 2), ([1150, 150_000, 150_000], 0, 2),
                           ([5830] * 3, 1, 1), ([560_000] * 6, 4, 4)):
        monkeypatch.setattr(L, "_FIT_SLOTS", {})
        A.fits_per_gpu = flag
        budget = L._fit_slots("cuda:0")
        assert admitted_without_blocking(budget, Ks, A) == want, (Ks, flag)
        assert L._fit_slots("cuda:0") is budget                      # one budget per process and device
    # release lets the queued one in, and a lone fit is always admitted whatever its size
    monkeypatch.setattr(L, "_FIT_SLOTS", {})
    A.fits_per_gpu = 0
    budget = L._fit_slots("cuda:1")
    c1, c2 = budget.acquire(10 ** 6, A), budget.acquire(10 ** 6, A)
    assert budget.admitted() == 2
    import threading
    late = []
    th = threading.Thread(target=lambda: late.append(budget.acquire(10 ** 6, A)), daemon=True)
    th.start()
    th.join(0.2)
    assert not late
    budget.release(c1)
    th.join(2)
    assert late == [3] and budget.admitted() == 2
    budget.release(c2), budget.release(late[0])
    assert budget.admitted() == 0


def _queued_fit(unit, device="cpu"):
    """A fit function with in-process admission like locator._fit_unit: ONE fit at a time per process however many fit
    threads the worker has; ("start", i) is reported through unit["on_admitted"] when the unit is admitted."""
    import threading
    import time
    gate = _queued_fit.__dict__.setdefault("gate", threading.Lock())
    with gate:
        unit.pop("on_admitted")()
        time.sleep(0.8)
    return {"name": unit["name"], "value": unit["replicate"], "seconds": 0.8}


_queued_fit.reports_admission = True


def test_unit_timeout_counts_from_admission_not_from_the_wait_for_a_sibling_fit():
    """with three fit threads and fewer admission slots a unit may wait for a sibling's whole fit inside the
    worker.  Its --unit_timeout clock starts when the fit function reports admission (unit["on_admitted"]), so a healthy
    run with fit time < timeout < 2 x fit time completes; it used to lose the worker and two healthy fits."""
    units = [dict(name=f"q{i}", replicate=i) for i in range(6)]
    logs = []
    res = R.run_units(units, _Args(), _queued_fit, n_gpus=1, fits_per_gpu=3, procs_per_gpu=1, log=logs.append, poll_s=0.05,
                      unit_timeout=1.3)
    assert [r.get("value") for r in res] == list(range(6)), (res, logs)
    assert not any("exceeded --unit_timeout" in str(l) for l in logs)
    # a fit function that does not report admission itself is started by the worker, exactly once
    res = R.run_units(_units(3), _Args(), _fake_fit, n_gpus=1, fits_per_gpu=2, procs_per_gpu=1, shared={"shared_bias": 1.0},
                      log=logs.append, poll_s=0.05, isolate=True)
    assert all("error" not in r for r in res)


def test_bench_launches_its_own_ranks_without_torchrun(repo_root):
    """`python bench.py --gpus N` with no launcher: the parent starts N rank processes before touching a GPU, the
    ranks rendezvous on 127.0.0.1, barrier, take the max over ranks, and rank 0's single JSON line comes back through
    the parent (the GPU-free --selftest-launch leg of the same code path; the timed leg is tests/test_gpu_cli.py)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "3", "--selftest-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"selftest": "launch", "n_gpus": 3, "max_over_ranks": 3.0}
    # a rank count that disagrees with the environment is refused
    r = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="4", RANK="0"))
    assert r.returncode == 2


def test_bench_line_is_short_strict_json_with_the_contract_keys(repo_root):
    """Round 4's line grew to 24 KB and the driver's bounded stdout tail could not be parsed (BENCH_r04.parsed = null).  The
    line is built by bench.format_line: at most 4 KB, strict JSON (NaN / Infinity never appear), every contract key
    present; optional detail is dropped before the bound is broken, and a line that cannot fit raises."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo_root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def fail(c):
        raise AssertionError("non-finite constant %r in the bench line" % c)

    shape = {"us": 123.4, "frac": 0.4567, "incl_prep": 0.3456, "packed": {"us": 120.1, "frac": 0.47, "incl_prep": 0.36}}
    out = {"metric": "training samples/sec on 1000x100k-SNP matrix", "value": 184354.1, "unit": "samples/s", "n_gpus": 1,
           "steps": 20, "warmup": 5, "ms_per_step": 4.3937, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "w" * 160, "step": "s" * 120, "width": 256, "nlayers": 10, "graph": True, "chained_steps": True,
                      "epochs_in_flight": 2, "replicates_per_gpu": 1, "replicates": "single model"},
           "us_per_minibatch_step": 168.99, "whole_step_hbm_frac": 0.5518, "final_loss": float("nan"), "final_val_loss": float("inf"),
           "roofline": {"bound": "hbm", "kernel": "l1_bwd_adam_chain_kernel", "achieved": 5863.3, "peak": 8000.0, "unit": "GB/s",
                        "frac": 0.7329, "traffic": 678083488, "traffic_source": "t" * 90, "bytes_per_launch": 653914409,
                        "us_per_launch": 111.53, "launches_timed": 52},
           "l1_gemm": {"mode": "int8x2", "peak_tflops": 2500.0, "guard_median": 28.1, "guard_max": 77.0, "us_prep": 60.4,
                       "rows_1000": shape, "distinct_4096": shape, "distinct_16384": shape},
           "cpu_baseline": {"value": 359.7, "unit": "samples/s", "cores": 16, "kind": "port", "impl": "i" * 90, "sample": "x" * 340,
                            "last_loss": 0.5, "last_val_loss": 0.35, "numpy_port": {"value": 160.1, "unit": "samples/s", "sample": "y" * 120}}}
    line = bench.format_line(out)
    assert "\n" not in line and len(line.encode()) < 4096
    got = json.loads(line, parse_constant=fail)
    for k in bench.REQUIRED_KEYS:
        assert k in got, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in got["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    assert got["final_loss"] is None and got["final_val_loss"] is None          # non-finite numbers become null, never NaN
    assert "l1_gemm" in got                                                     # a normal line keeps its summary
    # bloat is shed from the optional detail first ...
    fat = dict(out, l1_gemm=dict(out["l1_gemm"], sweep=["r" * 300] * 40))
    line = bench.format_line(fat)
    got = json.loads(line, parse_constant=fail)
    assert len(line) <= bench.MAX_LINE_BYTES and "l1_gemm" not in got and "roofline" in got and "cpu_baseline" in got
    # ... and a line whose CONTRACT part cannot fit is an error rather than an unparseable record
    with pytest.raises(RuntimeError):
        bench.format_line(dict(out, config={"workload": "w" * 5000}))
    with pytest.raises(RuntimeError):
        bench.format_line({k: v for k, v in out.items() if k != "roofline"})


def test_jacknife_draws_reproduce_the_reference_loop_and_its_rng_stream():
    """locator.py:713-727 draws, per replicate, the redrawn sites and then one np.random.binomial(2, af[i], n_pred)
    PER SITE in a Python loop.  jacknife_draws issues one broadcast binomial per replicate: same values, and the
    global stream is left in the same state (the next draw agrees)."""
    rs = np.random.RandomState(3)
    n_pred, K, nboots, prop = 11, 600, 4, 0.05
    predgen = rs.randint(0, 3, (n_pred, K)).astype(np.uint8)
    af = rs.uniform(0.01, 0.99, K)
    np.random.seed(2024)
    ref = []
    for _ in range(nboots):                                      # the reference's loop, on a scratch copy
        pg = predgen.copy()
        sites = np.random.choice(pg.shape[1], int(pg.shape[1] * prop), replace=False)
        for i in sites:
            pg[:, i] = np.random.binomial(2, af[i], pg.shape[0])
        ref.append(pg)
    tail_ref = np.random.random_sample()
    np.random.seed(2024)
    got = L.jacknife_draws(predgen, af, nboots, prop)
    tail_got = np.random.random_sample()
    assert tail_got == tail_ref
    for (sites, vals), want in zip(got, ref):
        pg = predgen.copy()
        pg[:, sites] = vals.T
        assert np.array_equal(pg, want)


def test_callbacks_configuration_is_what_the_fit_consumes():
    """train.Callbacks takes its patience / LR patience / LR factor from what load_callbacks returns:
    a shorter earlystop patience stops earlier, a different LR patience / factor changes the schedule."""
    from locator_amd.train import Callbacks
    vals = [1.0, 0.9, 0.95, 0.96, 0.97, 0.98, 0.99, 1.0, 1.01]

    def run(**kw):
        cb, lrs = Callbacks(**kw), []
        for e, v in enumerate(vals):
            save, stop, lr = cb.on_epoch_end(e, v)
            lrs.append(lr)
            if stop:
                return e, lrs
        return len(vals) - 1, lrs
    assert run(patience=3)[0] == 4 and run(patience=6)[0] == 7
    assert run(patience=100, lr_patience=2)[1][4] == float(np.float32(5e-4))          # halved after 2 bad epochs
    assert run(patience=100, lr_patience=2, lr_factor=0.1)[1][4] == float(np.float32(1e-3) * np.float32(0.1))
    assert run(patience=12)[1][-1] == float(np.float32(1.25e-4)) and Callbacks(patience=12).lr_patience == 2   # int(12 / 6)


# ------------------------------------------------------------------ blosc-compressed zarr
def test_blosc_decoder_on_the_references_own_chunks(repo_root):
    """locator_py/map.zarr in the reference holds 2,128 arrays written by zarr + numcodecs with the default
    compressor (Blosc-1, cname lz4, shuffle 1) - the same format `allel.vcf_to_zarr` writes for genotype stores
    (scripts/vcf_to_zarr.py:12).  Three of its chunks are fixtures here (flags 0x21 split LZ4 streams + byte-shuffle,
    0x31 unsplit, 0x33 stored).  They are country outlines: the decoded float64 arrays must be finite longitudes /
    latitudes and closed rings (first vertex == last vertex), which no wrong decoding produces."""
    root = os.path.join(repo_root, "tests", "golden", "blosc_map_zarr")
    seen = set()
    for name in sorted(os.listdir(root)):
        a = G.ZarrArray(os.path.join(root, name))
        raw = open(os.path.join(root, name, "0.0"), "rb").read()
        seen.add(raw[2])
        xy = a[:]
        assert xy.shape == a.shape and xy.dtype == np.float64 and xy.shape[0] == 2
        assert np.isfinite(xy).all() and np.abs(xy[0]).max() <= 180 and np.abs(xy[1]).max() <= 90
        assert np.array_equal(xy[:, 0], xy[:, -1]) and np.ptp(xy[0]) > 0
    assert seen == {0x21, 0x31, 0x33}
    afg = G.ZarrArray(os.path.join(root, "Afghanistan"))[:]
    assert 60 < afg[0].min() < afg[0].max() < 75.5 and 29 < afg[1].min() < afg[1].max() < 39     # where Afghanistan is


def test_blosc_zarr_store_round_trip_and_unsupported_codecs(tmp_path):
    """A genotype store written the way `allel.vcf_to_zarr` writes it (blosc / lz4 / shuffle on calldata/GT int8 and
    variants/POS int32) reads back identically through the C decoder, sliced like the window loop slices it; truncated
    chunks and what is not built (blosclz, bit-shuffle) fail loudly."""
    rng = np.random.default_rng(8)
    gt = (rng.random((3000, 41, 2)) < 0.2).astype(np.int8)
    gt[rng.random(gt.shape) < 0.01] = -1
    pos = np.sort(rng.choice(10_000_000, 3000, replace=False)).astype(np.int32)
    samples = np.array([f"s{i}" for i in range(41)])
    store = str(tmp_path / "b.zarr")
    G.write_callset_zarr(store, gt, pos, samples, chunk_variants=1024, compressor="blosc")
    meta = json.load(open(os.path.join(store, "calldata", "GT", ".zarray")))
    assert meta["compressor"]["id"] == "blosc" and meta["compressor"]["cname"] == "lz4"
    cs = G.open_group(store, mode="r")
    assert np.array_equal(cs["calldata/GT"][:], gt) and np.array_equal(cs["variants/POS"][:], pos)
    assert np.array_equal(cs["calldata/GT"][1000:2100, :, :], gt[1000:2100])
    assert list(cs["samples"][:]) == list(samples)
    raw = open(os.path.join(store, "calldata", "GT", "0.0.0"), "rb").read()
    assert len(raw) < 0.6 * 1024 * 41 * 2                                   # it really is compressed
    for typesize in (1, 4, 8):                                               # stream splitting / shuffle paths
        data = np.repeat(rng.integers(0, 50, 5000), 3).astype({1: np.int8, 4: np.int32, 8: np.int64}[typesize]).tobytes()
        assert G.blosc_decompress(G.blosc_compress(data, typesize)) == data
    assert G.blosc_decompress(G.blosc_compress(b"", 1)) == b""
    with pytest.raises(ValueError, match="malformed"):
        G.blosc_decompress(raw[:len(raw) // 2])
    blosclz = bytearray(raw)
    blosclz[2] = (0 << 5) | (raw[2] & 0x1F)
    with pytest.raises(ValueError, match="unsupported codec"):
        G.blosc_decompress(bytes(blosclz))
    lz4_labelled_zstd = bytearray(raw)                 # LZ4 streams are not zstd frames: malformed, not a crash
    lz4_labelled_zstd[2] = (4 << 5) | (raw[2] & 0x1F)
    with pytest.raises((ValueError, RuntimeError), match="malformed|libzstd"):
        G.blosc_decompress(bytes(lz4_labelled_zstd))
    bitshuf = bytearray(raw)
    bitshuf[2] |= 0x04
    with pytest.raises(ValueError, match="bit-shuffle"):
        G.blosc_decompress(bytes(bitshuf))


def test_weights_from_another_network_are_refused_before_any_device_call(tmp_path):
    """--load_weights / LocatorNet.import_params: a .weights.npz from another SNP count, --width or
    --nlayers must raise on the host; the swizzle kernel would otherwise read it as K x H."""
    from locator_amd import _lib
    from locator_amd.net import LocatorNet
    rng = np.random.default_rng(0)
    p = O.init_params(40, 16, 3, rng)
    path = str(tmp_path / "w.npz")
    L.save_weights(path, {**p, "mov_mean": p["mov_mean"], "mov_var": p["mov_var"]})
    z = np.load(path)
    assert int(z["n_snps"]) == 40 and int(z["width"]) == 16 and int(z["nlayers"]) == 3
    back = L.read_weights(path)
    assert len(back["W"]) == 5 and back["W"][0].shape == (40, 16)
    net = LocatorNet.__new__(LocatorNet)                 # shapes only: no device, no library call
    net.d = _lib.Dims(K=40, Kp=64, H=16, Hp=32, L=3, n_pre=1)
    net.check_params(back)
    for K, H, Lh, what in ((41, 16, 3, "SNPs 41"), (40, 32, 3, "--width 32"), (40, 16, 4, "--nlayers 4")):
        net.d = _lib.Dims(K=K, Kp=64, H=H, Hp=32, L=Lh, n_pre=Lh // 2)
        with pytest.raises(ValueError, match=what):
            net.check_params(back)
    net.d = _lib.Dims(K=40, Kp=64, H=16, Hp=32, L=3, n_pre=1)
    bad = dict(back, gamma=np.ones(39))
    with pytest.raises(ValueError, match="gamma"):
        net.check_params(bad)


@pytest.mark.parametrize("compressor", ["blosc-zstd", "zstd"])
def test_zstd_zarr_chunks_through_the_systems_libzstd(tmp_path, compressor):
    """Published Ag1000G-style stores often carry zstd (Blosc(cname='zstd') or numcodecs.Zstd): read through
    libzstd.so.1, bound with dlopen by csrc/codecs.c (VERDICT r02 missing #5; reference: scripts/vcf_to_zarr.py:12,
    locator.py:188-194).  Where the library is absent the reader must say so instead of failing obscurely."""
    if not G.zstd_available():
        with pytest.raises(RuntimeError, match="libzstd"):
            G.zstd_decompress(b"\x28\xb5\x2f\xfd" + b"\0" * 8, 16)
        pytest.skip("libzstd.so.1 is not on this machine")
    rng = np.random.default_rng(9)
    gt = (rng.random((2500, 37, 2)) < 0.25).astype(np.int8)
    gt[rng.random(gt.shape) < 0.01] = -1
    pos = np.sort(rng.choice(10_000_000, 2500, replace=False)).astype(np.int32)
    samples = np.array([f"s{i}" for i in range(37)])
    store = str(tmp_path / "z.zarr")
    G.write_callset_zarr(store, gt, pos, samples, chunk_variants=1000, compressor=compressor)
    meta = json.load(open(os.path.join(store, "calldata", "GT", ".zarray")))
    assert (meta["compressor"]["id"], meta["compressor"].get("cname")) == (("blosc", "zstd") if compressor == "blosc-zstd"
                                                                           else ("zstd", None))
    cs = G.open_group(store, mode="r")
    assert np.array_equal(cs["calldata/GT"][:], gt) and np.array_equal(cs["variants/POS"][:], pos)
    assert np.array_equal(cs["calldata/GT"][900:2100, :, :], gt[900:2100])
    raw = open(os.path.join(store, "calldata", "GT", "0.0.0"), "rb").read()
    assert len(raw) < 0.5 * 1000 * 37 * 2
    if compressor == "blosc-zstd":
        assert raw[2] >> 5 == 4
        for typesize in (1, 4):
            data = np.repeat(rng.integers(0, 50, 40_000), 3).astype({1: np.int8, 4: np.int32}[typesize]).tobytes()
            assert G.blosc_decompress(G.blosc_compress(data, typesize, cname="zstd")) == data
        with pytest.raises(ValueError, match="malformed"):
            G.blosc_decompress(raw[:len(raw) // 2])
    else:
        assert raw[:4] == b"\x28\xb5\x2f\xfd"                         # the zstd frame magic
        with pytest.raises(ValueError, match="malformed"):
            G.zstd_decompress(raw[:len(raw) // 2], 1000 * 37 * 2)


# ------------------------------------------------------------------ replicate pool: VERDICT r02 "next" #3 (b), (c), (d) + ADVICE
def test_pool_overlaps_host_work_with_the_previous_fit_and_reports_a_timeline():
    """(b) + (a): with host_prepare the parent keeps two units in flight per worker and the worker's loader thread does
    unit i + 1's host work while unit i fits: 6 units x (0.4 s host + 0.4 s fit) on ONE worker pair take ~0.4 + 6 x 0.4 s
    each side instead of 6 x 0.8 s.  The summary carries the phases and an Amdahl projection."""
    import time
    units = [dict(name=f"w{i}", replicate=i) for i in range(6)]
    pool = R.ReplicatePool(_Args(), _slow_fit, n_gpus=1, fits_per_gpu=2, host_prepare=_slow_host_prepare,
                           log=lambda *_: None, poll_s=0.05)
    t0 = time.time()
    pool.start()
    res = pool.run(units)
    wall = time.time() - t0
    pool.close()
    assert [r["value"] for r in res] == [0, 2, 4, 6, 8, 10]
    assert all(abs(r["host_prepare_seconds"] - 0.4) < 0.2 for r in res)
    per_worker = {}
    for r in res:
        per_worker.setdefault(r["pid"], []).append(r)
    assert len(per_worker) == 2
    # from the first dispatch (workers ready) to the end: 3 units per worker; serial host + fit would be >= 2.4 s,
    # overlapped it is ~0.4 + 3 x 0.4 = 1.6 s
    loop = pool.timeline["run_finished"] - min(u["dispatched"] for u in pool.timeline["units"].values())
    assert loop < 2.2, loop
    s = pool.summary(res)
    assert s["units"] == 6 and s["workers"] == 2 and 2.0 < s["unit_work_seconds"] < 3.5
    assert abs(s["host_prepare_seconds_total"] - 2.4) < 0.6 and 0 <= s["serial_fraction"] < 1
    assert s["amdahl_projection_seconds"][8] < s["amdahl_projection_seconds"][1] and len(s["lines"]) == 4 and "forkserver start" in s["lines"][3]
    assert len(pool.timeline["workers"]) == 2 and all(w["ready"] >= w["spawned"] for w in pool.timeline["workers"])


def test_pool_started_before_the_prologue_hides_worker_startup():
    """(c): start() returns at once; by the time a 1.5 s 'prologue' is over the workers have imported torch and said
    ready, so the first unit starts immediately."""
    import time
    pool = R.ReplicatePool(_Args(), _fake_fit, n_gpus=1, fits_per_gpu=2, log=lambda *_: None, poll_s=0.05)
    t0 = time.time()
    pool.start()
    assert time.time() - t0 < 1.0                      # spawning does not wait for the children
    time.sleep(6.0)                                     # the parent's prologue (import torch in a child takes seconds)
    t1 = time.time()
    res = pool.run(_units(4), {"shared_bias": 1.0})
    dt = time.time() - t1
    pool.close()
    assert [r["unit_index"] for r in res] == [0, 1, 2, 3] and dt < 2.0, dt
    assert all(w["ready"] < t1 for w in pool.timeline["workers"])


def test_pool_kills_a_hung_worker_after_unit_timeout_and_requeues_what_it_had_only_prefetched():
    """(d) + ADVICE r02: unit 1 hangs forever.  Its worker is killed after unit_timeout (exact pid), unit 1 becomes an
    error record, and the unit that worker had merely PREFETCHED is not lost: it goes back in the queue and another
    (or a fresh) worker fits it.  Same requeue when a worker dies right after its host phase."""
    import time
    units = [dict(name=f"w{i}", replicate=i) for i in range(7)]
    units[1]["hang"] = True
    units[4]["die_after_host"] = True
    logs = []
    t0 = time.time()
    res = R.run_units(units, _Args(), _slow_fit, n_gpus=1, fits_per_gpu=2, host_prepare=_slow_host_prepare,
                      log=logs.append, poll_s=0.1, unit_timeout=1.5)
    assert time.time() - t0 < 60
    assert [r["unit_index"] for r in res] == list(range(7))
    assert "timed out" in res[1]["error"] and "died" in res[4]["error"]
    for i in (0, 2, 3, 5, 6):
        assert "error" not in res[i] and res[i]["value"] == 2 * i, res[i]
    assert any("exceeded --unit_timeout" in str(l) for l in logs)


def test_pool_kills_a_worker_whose_host_phase_hangs():
    """A unit whose zarr slice / filter (loader thread) never returns never reports "start": the idle worker is killed
    after unit_timeout all the same, the unit becomes an error record and the rest of the run completes."""
    import time
    units = [dict(name=f"h{i}", replicate=i) for i in range(5)]
    units[2]["hang_host"] = True
    logs = []
    t0 = time.time()
    res = R.run_units(units, _Args(), _slow_fit, n_gpus=1, fits_per_gpu=2, host_prepare=_slow_host_prepare,
                      log=logs.append, poll_s=0.1, unit_timeout=1.5)
    assert time.time() - t0 < 60
    assert "host work timed out" in res[2]["error"]
    for i in (0, 1, 3, 4):
        assert "error" not in res[i] and res[i]["value"] == 2 * i, res[i]
    assert any("host work exceeded --unit_timeout" in str(l) for l in logs)


def _die_once_fit(unit, device="cpu"):
    """A fit that takes its whole worker process down the FIRST time unit `die_once` runs (marker file), while a sibling
    fit thread of that process is in the middle of another unit."""
    import time
    if unit.get("die_once") and not os.path.exists(unit["die_once"]):
        time.sleep(0.3)                      # the sibling thread has started its own unit by now
        open(unit["die_once"], "w").close()
        os.kill(os.getpid(), 9)
    time.sleep(0.6)
    return {"name": unit["name"], "value": unit["replicate"] * 3, "seconds": 0.6, "pid": os.getpid()}


def test_pool_requeues_the_sibling_fit_of_a_crashed_worker_and_isolates_a_single_worker_under_unit_timeout(tmp_path):
    """ADVICE r04 (medium).  (1) One process x two fit threads (the default layout): a crash says nothing about WHICH fit
    aborted, so every unit the worker held is tried once more and fails only when a second worker is lost with it - the
    innocent sibling no longer becomes an error record (here even the culprit recovers: it dies only once).  (2) With
    --unit_timeout a single worker is still a separate PROCESS (the in-process path cannot kill a hung fit): the hung unit
    becomes an error record, its siblings are fitted by the replacement."""
    import time
    units = [dict(name=f"c{i}", replicate=i) for i in range(5)]
    units[1]["die_once"] = str(tmp_path / "died")
    logs = []
    res = R.run_units(units, _Args(), _die_once_fit, n_gpus=1, fits_per_gpu=2, procs_per_gpu=1, log=logs.append, poll_s=0.1,
                      isolate=True)
    assert [r["unit_index"] for r in res] == list(range(5))
    assert all("error" not in r and r["value"] == 3 * i for i, r in enumerate(res)), res
    assert len({r["pid"] for r in res}) >= 1 and os.getpid() not in {r["pid"] for r in res}
    # (2)
    pool = R.ReplicatePool(_Args(), _slow_fit, n_gpus=1, fits_per_gpu=2, procs_per_gpu=1, host_prepare=_slow_host_prepare,
                           log=logs.append, poll_s=0.1, unit_timeout=1.5)
    assert pool.n == 1 and pool.threads == 2 and pool.isolate
    units = [dict(name=f"t{i}", replicate=i) for i in range(5)]
    units[2]["hang"] = True
    t0 = time.time()
    res = pool.run(units)
    pool.close()
    assert time.time() - t0 < 60
    assert "timed out" in res[2]["error"]
    for i in (0, 1, 3, 4):
        assert "error" not in res[i] and res[i]["value"] == 2 * i and res[i]["pid"] != os.getpid(), res[i]
    # without a timeout the single worker stays in this process (no spawn, no pickling of the units)
    pool = R.ReplicatePool(_Args(), _slow_fit, n_gpus=1, fits_per_gpu=2, procs_per_gpu=1, host_prepare=_slow_host_prepare,
                           log=logs.append, poll_s=0.1)
    res = pool.run([dict(name=f"p{i}", replicate=i) for i in range(3)])
    pool.close()
    assert all(r["pid"] == os.getpid() for r in res)


def test_pool_reports_a_loader_thread_failure_instead_of_hanging(monkeypatch):
    """the shared-memory segment a worker is told to attach does not exist (gone, /dev/shm full).
    The loader thread's failure used to kill only that daemon thread - the process stayed alive, blocked on its queue,
    and with the default --unit_timeout 0 the run hung.  Now the worker reports ("dead", why) and exits; every unit
    ends as an error record (each is tried by at most two workers) and the run returns."""
    import time
    units = [dict(name=f"s{i}", replicate=i) for i in range(4)]
    monkeypatch.setattr(R, "_share", lambda shared: ({}, {"big": ("psm_locator_test_no_such_segment", (4,), "<f4")}, []))
    logs = []
    t0 = time.time()
    res = R.run_units(units, _Args(), _slow_fit, n_gpus=1, fits_per_gpu=2, log=logs.append, poll_s=0.1)
    assert time.time() - t0 < 120
    assert len(res) == 4 and all(r is not None and "error" in r for r in res), res
    assert any("loader thread failed" in str(l) for l in logs), logs


def test_zarr_read_into_equals_getitem(tmp_path):
    """ZarrArray.read_into (the --windows loader reads a slice straight into pinned memory): uncompressed chunks through
    file offsets, compressed chunks through the threaded decode, partial first / last chunks, a missing chunk."""
    rng = np.random.default_rng(3)
    gt = rng.integers(-1, 3, (1000, 17, 2)).astype(np.int8)
    for comp in (None, "blosc", "zlib"):
        p = str(tmp_path / ("gt_" + str(comp)))
        G.write_zarr_array(p, gt, (128, 17, 2), compressor=comp)
        za = G.ZarrArray(p)
        for a, b in ((0, 1000), (5, 130), (128, 256), (300, 301), (900, 1000), (0, 0)):
            out = np.full((b - a, 17, 2), 99, np.int8)
            za.read_into(out, a, b, threads=3)
            assert np.array_equal(out, gt[a:b]) and np.array_equal(za[a:b], gt[a:b])
        os.remove(os.path.join(p, "2.0.0"))
        out = np.full((384, 17, 2), 99, np.int8)
        za.read_into(out, 128, 512)
        assert np.array_equal(out[:128], gt[128:256]) and not out[128:256].any() and np.array_equal(out[256:], gt[384:512])
    with pytest.raises(ValueError):
        za.read_into(np.zeros((3, 17, 2), np.int16), 0, 3)


def _quick_fit(unit, device="cpu"):
    import threading
    import time
    t0 = time.time()
    time.sleep(0.4)
    if unit.get("explode"):
        raise RuntimeError("boom")
    return {"name": unit["name"], "value": unit["replicate"] * 2, "seconds": time.time() - t0, "pid": os.getpid(),
            "thread": threading.get_ident()}


def test_eight_gpu_layout_on_cpu_one_worker_per_gpu_two_fit_threads_each(monkeypatch):
    """The default 8-GPU layout (BASELINE configs[3] / [4]: replicates sharded over the GPUs of one node, no collective) with
    the device count faked - no 8-GPU node was available to any round, so this is what can be shown here: 8 worker processes
    (one per GPU index) x 2 fit threads = 16 concurrent fits; every unit fitted exactly once, records in unit order, every
    GPU index used, no more workers than needed for few units, and the timeline's Amdahl projection is there.  A unit that
    raises does not cost its worker; a worker that dies once does not cost its units."""
    import time
    monkeypatch.setattr(R, "visible_gpus", lambda: 8)
    units = [dict(name=f"g{i}", replicate=i) for i in range(120)]
    units[7]["explode"] = True
    logs = []
    pool = R.ReplicatePool(_Args(), _quick_fit, fits_per_gpu=2, procs_per_gpu=1, log=logs.append, poll_s=0.05,
                           max_workers=len(units))
    assert (pool.n_g, pool.n, pool.threads, pool.fits) == (8, 8, 2, 16)
    t0 = time.time()
    pool.start()
    res = pool.run(units)
    wall = time.time() - t0
    pool.close()
    assert [r["unit_index"] for r in res] == list(range(120))
    assert "error" in res[7] and "boom" in res[7]["error"]
    ok = [r for i, r in enumerate(res) if i != 7]
    assert all("error" not in r and r["value"] == 2 * r["unit_index"] for r in ok)
    # dynamic dispatch: a worker whose start-up is slow (eight `import torch` on this container's 8 CPUs) gets fewer units or,
    # when it is very late, none - so "most", not "all", of the GPU indices must show up
    assert len({r["gpu"] for r in res}) >= 6 and {r["gpu"] for r in res} <= set(range(8))
    assert len({r["pid"] for r in ok}) >= 6 and os.getpid() not in {r["pid"] for r in ok}
    by_pid = {}
    for r in ok:
        by_pid.setdefault(r["pid"], set()).add(r["thread"])
    assert max(len(t) for t in by_pid.values()) == 2 and sum(len(t) for t in by_pid.values()) >= 12   # fit threads really worked side by side
    # 120 units of 0.4 s on 16 concurrent fits: 8 rounds = 3.2 s of fits (48 s one at a time); the rest is the start-up of
    # eight worker processes, so the bound is loose
    assert wall < 90, wall
    s = pool.summary(res)
    assert s["gpus"] == 8 and s["workers"] == 8 and s["fit_threads"] == 2 and s["units"] == 120
    assert set(s["amdahl_projection_seconds"]) == {1, 2, 4, 8}
    # few units: never more concurrent fits (and workers) than units
    small = R.ReplicatePool(_Args(), _quick_fit, fits_per_gpu=2, procs_per_gpu=1, max_workers=3)
    assert small.n * small.threads >= 3 and small.n <= 3


def test_fit_threads_in_one_process_and_in_worker_processes():
    """VERDICT r03 next #4c: the concurrent fits of a GPU run as threads of ONE process (each on its own stream on a GPU;
    here the scheduling only).  One process x 2 threads (this process drives them) and 2 processes x 2 threads: every unit
    once, results in unit order, fits really overlap, a failing unit does not stop its sibling thread, and a worker that dies
    takes BOTH of its running units with it while the rest is rerouted."""
    import time
    import torch  # noqa: F401  (its first import takes seconds; not part of what is timed below)
    units = [dict(name=f"t{i}", replicate=i) for i in range(8)]
    units[3]["explode"] = True
    t0 = time.time()
    res = R.run_units(units, _Args(), _quick_fit, n_gpus=1, fits_per_gpu=2, procs_per_gpu=1, log=lambda *_: None)
    dt = time.time() - t0
    assert dt < 8 * 0.4 * 0.8, dt                                     # two at a time: ~1.6 s, not 3.2 s
    assert [r["unit_index"] for r in res] == list(range(8)) and "boom" in res[3]["error"]
    ok = [r for i, r in enumerate(res) if i != 3]
    assert all(r["value"] == 2 * r["unit_index"] for r in ok)
    assert len({r["pid"] for r in ok}) == 1 and len({r["thread"] for r in ok}) == 2
    pool = R.ReplicatePool(_Args(), _quick_fit, n_gpus=1, fits_per_gpu=4, procs_per_gpu=2, log=lambda *_: None, poll_s=0.1)
    assert (pool.n, pool.threads, pool.fits) == (2, 2, 4)
    t0 = time.time()
    res = pool.run(units)
    dt = time.time() - t0
    pool.close()
    assert [r["unit_index"] for r in res] == list(range(8)) and "boom" in res[3]["error"]
    ok = [r for i, r in enumerate(res) if i != 3]
    assert len({r["pid"] for r in ok}) == 2 and len({(r["pid"], r["thread"]) for r in ok}) == 4
    assert pool.summary(res)["fit_threads"] == 2
    # never more concurrent fits than units
    assert R.ReplicatePool(_Args(), _quick_fit, n_gpus=1, fits_per_gpu=2, procs_per_gpu=1, max_workers=1).threads == 1
    p3 = R.ReplicatePool(_Args(), _quick_fit, n_gpus=1, fits_per_gpu=4, procs_per_gpu=2, max_workers=3)
    assert p3.n * p3.threads >= 3 and p3.n <= 2


def test_native_host_filter_and_transposes_equal_the_numpy_restatement():
    """csrc/codecs.c loc_snp_flags / loc_snp_allele_counts / loc_rows_transposed (the parent's prologue of --bootstrap and the
    eager --windows path) against the NumPy spelling of allel's count_alleles -> is_biallelic -> to_allele_counts()[:, :, 1]
    (genotypes.filter_snps(native=False)) and `ac[:, rows].T`: missing calls, tri-allelic and monomorphic sites, alleles up
    to 5, min_mac 1 / 2 / 7, haploid and triploid calls, odd sizes around the 64-SNP blocks of the transpose."""
    rng = np.random.default_rng(11)
    for V, N, P in ((1000, 37, 2), (4099, 130, 2), (257, 513, 2), (300, 20, 1), (300, 20, 3), (63, 9, 2), (1, 5, 2)):
        af = rng.beta(0.3, 0.9, V)
        af[rng.random(V) < 0.1] = 0.0
        gt = (rng.random((V, N, P)) < af[:, None, None]).astype(np.int8)
        gt[rng.random(V) < 0.05, rng.integers(0, N), 0] = rng.integers(2, 6)
        gt[rng.random((V, N, P)) < 0.02] = -1
        for mm in (1, 2, 7):
            ref = G.filter_snps(gt, mm, verbose=False, native=False)
            got = G.filter_snps(gt, mm, verbose=False, native=True)
            assert got.dtype == ref.dtype and got.shape == ref.shape and np.array_equal(got, ref), (V, N, P, mm)
        rows = rng.permutation(N)[: max(1, N - 2)]
        assert np.array_equal(G.rows_transposed(ref, rows), np.ascontiguousarray(ref[:, rows].T))
        assert G.rows_transposed(ref, np.zeros(0, np.int64)).shape == (0, ref.shape[0])
    v = G.read_vcf(VCF)["calldata/GT"]                         # the reference's example: 5,830 SNPs either way
    assert np.array_equal(G.filter_snps(v, 2, verbose=False), G.filter_snps(v, 2, verbose=False, native=False))
    assert G.filter_snps(v, 2, verbose=False).shape[0] == 5830


def test_worker_gpu_binding_maps_the_parents_visible_list_and_counting_gpus_needs_no_torch(monkeypatch, tmp_path):
    """replicates._bind_worker_to_gpu (one process per GPU through HIP_VISIBLE_DEVICES): worker g of a
    parent that sees everything gets "g"; of a parent restricted by --gpu_number / HIP_VISIBLE_DEVICES the g-th entry of that
    list - which travels explicitly, a forked worker inherits the fork server's environment; CUDA_VISIBLE_DEVICES is folded in
    and removed.  visible_gpus() reads the KFD topology (nodes with simd_count > 0), narrowed by the same variables."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    R._bind_worker_to_gpu(3, {})
    assert os.environ["HIP_VISIBLE_DEVICES"] == "3" and "CUDA_VISIBLE_DEVICES" not in os.environ
    R._bind_worker_to_gpu(1, {"HIP_VISIBLE_DEVICES": "4,6,7", "CUDA_VISIBLE_DEVICES": None})
    assert os.environ["HIP_VISIBLE_DEVICES"] == "6"
    R._bind_worker_to_gpu(0, {"HIP_VISIBLE_DEVICES": None, "CUDA_VISIBLE_DEVICES": "5"})
    assert os.environ["HIP_VISIBLE_DEVICES"] == "5" and "CUDA_VISIBLE_DEVICES" not in os.environ
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,3")
    assert R._visible_env()["HIP_VISIBLE_DEVICES"] == "2,3"
    # a fake KFD topology: one CPU node, three GPU nodes
    import glob as _glob
    nodes = []
    for i, simd in enumerate((0, 1024, 1024, 1024)):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
        nodes.append(str(d / "properties"))
    monkeypatch.setattr(_glob, "glob", lambda pat: nodes if "kfd" in pat else [])
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert R.visible_gpus() == 3
    # a container that sees the host's topology but was given only some render nodes: a GPU counts when its device file opens
    for i, minor in ((1, 128), (2, 129), (3, 130)):
        with open(nodes[i], "a") as fh:
            fh.write(f"drm_render_minor {minor}\n")
    real_access = os.access
    monkeypatch.setattr(os, "access", lambda path, mode: path.endswith("renderD129") if "/dev/dri/" in str(path) else real_access(path, mode))
    assert R.visible_gpus() == 1
    monkeypatch.setattr(os, "access", lambda path, mode: True if "/dev/dri/" in str(path) else real_access(path, mode))
    assert R.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert R.visible_gpus() == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert R.visible_gpus() == 1
