#!/usr/bin/env python3
"""`locator` command line on MI355X — drop-in for /root/reference/locator/locator.py.

Same flags (locator.py:12-167), same pipeline (main, locator.py:487-749), same output files
(`{out}_predlocs.txt`, `{out}_history.txt`, `{out}_params.json`, `{out}_fitplot.pdf`), but the
Keras model / fit / predict are replaced by hand-written HIP kernels for gfx950 driven through
liblocator_hip.so, and the --windows / --bootstrap replicate loops are sharded over the GPUs of the
node (one model per GPU, no collectives).

Function names and signatures follow the reference so that its call sites read the same:
  load_genotypes, sort_samples, replace_md, filter_snps, normalize_locs, split_train_test,
  load_network, load_callbacks, train_network, predict_locs, plot_history, main.

Deliberate, documented deviations:
  * flags are parsed in main(), not at import; params.json is therefore written once, not twice
    (same content);
  * --keep_weights writes `{stem}.weights.npz` (NumPy archive, Keras tensor orientation) because
    h5py is not available; the best-epoch snapshot itself lives in HBM, not on disk;
  * --batch_size is limited to 4096 rows (the reference default is 32; above 32 it needs a --width that pads to 64,
    128 or 256 and --nlayers >= 4 with dropout; above 128 the step is correct but not tuned) and --width to 1024 (above 512: per-layer kernels);
  * extra flags --gpus / --fits_per_gpu / --unit_timeout / --no_graph / --no_chain / --net_seed / --load_weights / --predict_mode / --predict_packed / --predict_pieces (recorded at
    the end of params.json).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

from . import genotypes as G


def build_parser():
    p = argparse.ArgumentParser(prog="locator")
    p.add_argument("--vcf", help="VCF (optionally gzipped) holding the SNPs of every sample")
    p.add_argument("--zarr", help="zarr-v2 directory store with calldata/GT, variants/POS and samples")
    p.add_argument("--matrix", help="tab-delimited table: a 'sampleID' column followed by one 0/1/2 "
                                    "allele-count column per site")
    p.add_argument("--sample_data", help="tab-delimited table with columns sampleID, x, y; x and y are NA "
                                         "for the samples whose location is to be predicted")
    p.add_argument("--train_split", default=0.9, type=float,
                   help="fraction of located samples used for training (default 0.9)")
    p.add_argument("--windows", default=False, action="store_true",
                   help="fit one model per genomic window of a single chromosome (needs --zarr)")
    p.add_argument("--window_start", default=0, help="first window start (default 0)")
    p.add_argument("--window_stop", default=None, help="end of the last window (default: largest position)")
    p.add_argument("--window_size", default=5e5, help="window length in bp (default 500000)")
    p.add_argument("--bootstrap", default=False, action="store_true",
                   help="refit on SNP-resampled copies of the data (--nboots replicates after a full fit)")
    p.add_argument("--jacknife", default=False, action="store_true",
                   help="quick uncertainty heuristic: re-predict with a fraction of SNPs redrawn")
    p.add_argument("--jacknife_prop", default=0.05, type=float,
                   help="fraction of SNPs redrawn per jacknife replicate (default 0.05)")
    p.add_argument("--nboots", default=50, type=int, help="number of bootstrap / jacknife replicates (default 50)")
    p.add_argument("--batch_size", default=32, type=int,
                   help="minibatch size (default 32; 1..4096 here; above 32 needs a --width that pads to 64, 128 or 256)")
    p.add_argument("--max_epochs", default=5000, type=int, help="upper bound on training epochs (default 5000)")
    p.add_argument("--patience", type=int, default=100,
                   help="epochs without validation improvement before training stops (default 100)")
    p.add_argument("--min_mac", default=2, type=int, help="minimum allele-1 count for a SNP to be kept (default 2)")
    p.add_argument("--max_SNPs", default=None, type=int, help="use a random subset of this many SNPs (default all)")
    p.add_argument("--impute_missing", default=False, action="store_true",
                   help="draw missing calls from Binomial(2, site frequency) instead of counting them as 0")
    p.add_argument("--dropout_prop", default=0.25, type=float, help="dropout rate of the middle layer (default 0.25)")
    p.add_argument("--nlayers", default=10, type=int, help="number of hidden layers (default 10)")
    p.add_argument("--width", default=256, type=int, help="units per hidden layer (default 256; at most 1024 here)")
    p.add_argument("--out", help="stem of every output file")
    p.add_argument("--seed", default=None, type=int, help="NumPy seed for the train/validation split and SNP draws")
    p.add_argument("--gpu_number", default=None, type=str, help="restrict the run to this GPU index")
    p.add_argument("--plot_history", default=True, type=bool, help="write {out}_fitplot.pdf (default True)")
    p.add_argument("--keep_weights", default=False, action="store_true", help="keep the best weights on disk")
    p.add_argument("--load_params", default=None, type=str,
                   help="path of a _params.json from an earlier run; replaces every command-line value")
    p.add_argument("--keras_verbose", default=1, type=int,
                   help="0 silent, 1 or 2 one line per epoch (default 1)")
    # --- additions of this implementation (kept last so the reference's keys keep their order)
    p.add_argument("--gpus", default=None, type=int,
                   help="GPUs used to shard --windows / --bootstrap replicates (default: all visible)")
    p.add_argument("--fits_per_gpu", default=0, type=int,
                   help="concurrent replicate fits per GPU for --windows / --bootstrap (default 0 = by SNP count: 3 up to "
                        "70,000 SNPs per fit, 2 above - measured 903k / 711k samples/s for 3 / 2 fits at 5,830 SNPs, 221k / 226k "
                        "at 100,000)")
    p.add_argument("--procs_per_gpu", default=1, type=int,
                   help="worker processes per GPU for --windows / --bootstrap (default 1): the --fits_per_gpu concurrent fits "
                        "of a GPU run as that many threads of ONE process, each on its own stream - one device context, "
                        "one start-up; --procs_per_gpu 2 with --fits_per_gpu 2 is the rounds 1-3 layout (one fit per process)")
    p.add_argument("--host_filter", default=False, action="store_true",
                   help="--windows: filter each window's SNPs on the host (NumPy) as rounds 1-3 did, instead of uploading "
                        "the raw calls and filtering on the device (same rows bit for bit; measurement / fallback switch)")
    p.add_argument("--unit_timeout", default=0, type=float,
                   help="seconds one --windows / --bootstrap replicate may take inside a worker before that worker is "
                        "killed and replaced and the replicate reported as failed (default 0: no limit)")
    p.add_argument("--worker_start", default="forkserver", choices=("forkserver", "spawn"),
                   help="how --windows / --bootstrap worker processes start: forked from a server that imported torch once and "
                        "never touches a GPU (default: a worker costs a fork + one device context), or a fresh interpreter each")
    p.add_argument("--in_process", default=False, action="store_true",
                   help="--windows / --bootstrap on ONE GPU: run the replicate fits on threads of this process instead of a "
                        "worker process (saves the worker's start-up; no crash isolation, and --unit_timeout then still "
                        "forces a worker process: only a process can be killed)")
    p.add_argument("--no_graph", default=False, action="store_true", help="do not capture epochs into HIP graphs")
    p.add_argument("--no_chain", default=False, action="store_true",
                   help="one layer-1 forward launch per minibatch step instead of chaining it into the previous step's "
                        "layer-1 backward (the round-2 schedule; same results up to fp32 round-off, about 10 %% slower)")
    p.add_argument("--load_weights", default=None, type=str,
                   help="a .weights.npz written by --keep_weights: skip training and predict with these weights")
    p.add_argument("--predict_mode", default="auto", choices=("auto", "exact", "fast"),
                   help="first-layer arithmetic of many-row predictions on the int8 matrix pipe: auto (default) = 16-bit "
                        "fixed point per weight while the dynamic range of the trained weights allows it (largest / rms "
                        "weight per unit: median <= 64, worst <= 512 - predictions within 1e-3 relative, measured 5e-5 on a "
                        "converged fit, tests/test_gpu_trained_predict.py), else as exact; exact = 24-bit (as accurate as "
                        "fp32 accumulation; falls back to exactly-split bf16 when a unit's range exceeds 1024); fast = "
                        "16-bit unconditionally")
    p.add_argument("--predict_packed", default=False, action="store_true",
                   help="keep a 2-bit packed copy of every genotype matrix that is predicted from over 3072 rows or more "
                        "(values 0..3 only): such predictions read a quarter of the genotype bytes, with identical results. "
                        "Off by default since round 6: the one extra pass over the matrix costs as much as the packed form "
                        "saves in about eight such predictions, and the CLI predicts from a matrix once or twice")
    p.add_argument("--no_predict_packed", dest="predict_packed", action="store_false",
                   help="(default) never build the packed copy")
    p.add_argument("--predict_pieces", default=None, type=int,
                   help="force the bf16 matrix pipe with this many pieces per first-layer weight instead: 3 = "
                        "fp32-exact products, 1 or 2 = faster, approximate (default: bf16 x 3 only where the int8 "
                        "form does not apply - few rows, genotype values above 127)")
    p.add_argument("--net_seed", default=None, type=int,
                   help="seed of weight init / shuffling / dropout (the reference leaves these unseeded); "
                        "default: --seed, else entropy")
    return p


args = None          # module-level namespace, as in the reference (locator.py:167)


def _setup(argv=None):
    """Parse flags, honour --seed / --gpu_number / --load_params and record the run in {out}_params.json
    (what the reference does at import time, locator.py:169-184, and again at the top of main, :490-505)."""
    global args
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.load_params is not None:            # every value comes from the file; keys it lacks keep their defaults
        merged = vars(parser.parse_args([]))
        with open(args.load_params) as fh:
            merged.update(json.load(fh))
        args = argparse.Namespace(**merged)
    if args.seed is not None:
        np.random.seed(args.seed)
    if args.gpu_number is not None:
        for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            os.environ[var] = args.gpu_number
    _write_atomic(args.out + "_params.json", lambda fh: json.dump(vars(args), fh, indent=2))
    seed = args.net_seed if args.net_seed is not None else args.seed
    args._net_seed = seed if seed is not None else int.from_bytes(os.urandom(4), "little")
    return args


def _write_atomic(path, writer, mode="w"):
    """Write through a temporary file in the same directory and rename it into place: concurrent replicate workers
    (and a reader of a half-finished run) never see a truncated file."""
    import threading
    tmp = f"{path}.tmp{os.getpid()}_{threading.get_ident()}"     # replicate fits may share a process (one thread each)
    with open(tmp, mode) as fh:
        writer(fh)
    os.replace(tmp, path)


# ------------------------------------------------------------------ ingest (locator.py:187-308)
def load_genotypes():
    """(variants, samples, 2) int8 calls and the sample IDs from whichever of --zarr / --vcf / --matrix was given
    (locator.py:187-228); the readers live in genotypes.py (no scikit-allel / zarr dependency)."""
    if args.zarr is not None:
        print("reading zarr")
        callset = G.open_group(args.zarr, mode="r")
        za = callset["calldata/GT"]
        if za.dtype == np.int8 and za.ndim == 3:
            gt = np.empty(za.shape, np.int8)            # chunks decoded / read on a few threads, straight into place
            za.read_into(gt, 0, za.shape[0], threads=G.HOST_THREADS)
        else:
            gt = np.asarray(za[:], dtype=np.int8)
        return gt, np.asarray(callset["samples"][:])
    if args.vcf is not None:
        print("reading VCF")
        vcf = G.read_vcf(args.vcf)
        return vcf["calldata/GT"], vcf["samples"]
    if args.matrix is not None:
        return G.read_matrix(args.matrix)
    raise SystemExit("one of --zarr, --vcf or --matrix is required")


def sort_samples(samples, genotypes):
    """Coordinates in genotype-sample order (locator.py:231-247): the sample table is looked up by sampleID for
    every genotype sample; an ID missing from the table aborts the run like the reference does."""
    import pandas as pd
    table = pd.read_csv(args.sample_data, sep="\t")
    ids = np.asarray(samples).astype(str)
    sample_data = table.assign(sampleID2=table["sampleID"]).set_index("sampleID").reindex(ids)
    if not np.array_equal(sample_data["sampleID2"].to_numpy(dtype=object).astype(str), ids):
        print("sample ordering failed! Check that sample IDs match the VCF.")
        sys.exit()
    locs = sample_data[["x", "y"]].to_numpy()
    print("loaded " + str(np.shape(genotypes)) + " genotypes\n\n")
    return sample_data, locs


def replace_md(genotypes):
    print("imputing missing data")
    return G.replace_md(genotypes)


def filter_snps(genotypes):
    return G.filter_snps(genotypes, min_mac=args.min_mac, max_snps=args.max_SNPs,
                         impute_missing=args.impute_missing)


def normalize_locs(locs):
    """z-score each coordinate over the located samples (locator.py:284-292; NaN rows stay NaN)."""
    (meanlong, sdlong), (meanlat, sdlat) = [(np.nanmean(col), np.nanstd(col)) for col in (locs[:, 0], locs[:, 1])]
    scaled = np.column_stack([(locs[:, 0] - meanlong) / sdlong, (locs[:, 1] - meanlat) / sdlat])
    return meanlong, sdlong, meanlat, sdlat, scaled


def split_indices(locs, train_split):
    """The index part of split_train_test (locator.py:296-302): one np.random.choice from the global stream."""
    train = np.argwhere(~np.isnan(locs[:, 0]))[:, 0]
    known = set(train.tolist())
    pred = np.array([x for x in range(len(locs)) if x not in known], dtype=np.int64)
    test = np.random.choice(train, round((1 - train_split) * len(train)), replace=False)
    tset = set(test.tolist())
    train = np.array([x for x in train if x not in tset])
    return train, test, pred


def split_train_test(ac, locs):
    """locator.py:295-308: row sets from split_indices, genotypes as sample-major matrices."""
    train, test, pred = split_indices(locs, args.train_split)
    rows = lambda idx: G.rows_transposed(ac, idx) if len(idx) else np.zeros((0, ac.shape[0]), ac.dtype)
    return train, test, rows(train), rows(test), locs[train], locs[test], pred, rows(pred)


# ------------------------------------------------------------------ model (locator.py:311-470)
class DeviceRows:
    """Rows [start, start+n) of a genotype matrix already resident in HBM; stands in for the NumPy
    traingen / testgen / predgen of the reference so replicates need no host copies."""

    def __init__(self, X, start, n, K):
        self.X, self.start, self.n = X, int(start), int(n)
        self.shape = (int(n), int(K))


class Model:
    """What `load_network` returns in place of a compiled keras.Sequential: the hyper-parameters now,
    the device-resident LocatorNet once train_network has seen the data."""

    def __init__(self, n_snps, dropout_prop, width, nlayers, seed, replicate=0, device="cuda:0"):
        self.n_snps, self.dropout_prop, self.width, self.nlayers = n_snps, dropout_prop, width, nlayers
        self.seed, self.replicate, self.device = seed, replicate, device
        self.net = None
        self.n_train = self.n_val = 0
        self._val_key = None

    def _build(self, X, Y):
        from .net import LocatorNet
        self.net = LocatorNet(X, Y, self.n_snps, self.width, self.nlayers, self.dropout_prop, seed=self.seed,
                              replicate=self.replicate, device=self.device,
                              **predict_settings(args))
        return self.net

    def predict(self, gen):
        """model.predict(x) (locator.py:414, :441): inference-mode forward, float32 (n, 2)."""
        import torch
        from .net import upload_genotypes
        n = gen.shape[0]
        if n == 0:
            return np.zeros((0, 2), np.float32)
        net = self.net
        keepX = net.X
        if isinstance(gen, DeviceRows):
            net.X, start = gen.X, gen.start
        else:
            net.X, start = upload_genotypes(np.asarray(gen), self.device), 0
        net.auto_pack = bool(getattr(args, "predict_packed", False))  # opt-in; the packed copy is cached on the matrix itself
        net.cnet()
        rows = torch.arange(start, start + n, dtype=torch.int32, device=self.device)
        yhat = torch.zeros((n, 2), dtype=torch.float32, device=self.device)
        net.predict_rows(rows, n, yhat)
        torch.cuda.current_stream().synchronize()     # this fit's stream only: a device-wide wait would break into the graph
        #                                               capture of another fit running on another thread of this process
        net.X = keepX
        net.cnet()
        return yhat.cpu().numpy()

    def weights_dict(self):
        return self.net.export_params()


def predict_settings(a):
    """--predict_mode / --predict_pieces -> LocatorNet(predict_pieces=, predict_digits=)."""
    forced = getattr(a, "predict_pieces", None)
    if forced is not None:
        return {"predict_pieces": int(forced), "predict_digits": -1}
    return {"predict_pieces": 3, "predict_digits": {"auto": 0, "exact": 3, "fast": 2}[getattr(a, "predict_mode", "auto")]}


def load_network(traingen, dropout_prop, replicate=0, device="cuda:0"):
    # the reference ignores its dropout_prop argument and reads args.dropout_prop
    return Model(traingen.shape[1], args.dropout_prop, args.width, args.nlayers, args._net_seed, replicate, device)


_TLS = __import__("threading").local()


def _out():
    """Output stem of the fit this thread is running: a replicate unit's own stem while _fit_unit runs it (several units
    may be fitting at once in one process, each on its own thread and stream), args.out otherwise."""
    return getattr(_TLS, "out", None) or args.out


def _weights_path(boot):
    if args.bootstrap or args.jacknife:
        return _out() + "_boot" + str(boot) + ".weights.npz"
    return _out() + ".weights.npz"


def load_callbacks(boot):
    """locator.py:330-362.  The three Keras callbacks are state machines on val_loss (train.Callbacks); what is
    returned here, in the reference's order, is their configuration - and train_network takes it from here: editing
    earlystop["patience"], reducelr["patience"] / ["factor"] or checkpointer["filepath"] changes the fit."""
    checkpointer = {"callback": "ModelCheckpoint", "filepath": _weights_path(boot), "save_best_only": True,
                    "save_weights_only": True, "monitor": "val_loss"}
    earlystop = {"callback": "EarlyStopping", "monitor": "val_loss", "min_delta": 0, "patience": args.patience}
    reducelr = {"callback": "ReduceLROnPlateau", "monitor": "val_loss", "factor": 0.5,
                "patience": int(args.patience / 6), "min_delta": 0, "cooldown": 0, "min_lr": 0}
    return checkpointer, earlystop, reducelr


WEIGHT_KEYS = ("gamma", "beta", "moving_mean", "moving_variance")


def save_weights(path, w):
    """--keep_weights artefact: NumPy archive in Keras tensor orientation (the reference keeps a Keras HDF5 file,
    locator.py:332-348; h5py is not available here)."""
    flat = {"gamma": w["gamma"], "beta": w["beta"], "moving_mean": w["mov_mean"], "moving_variance": w["mov_var"]}
    for i, (k, b) in enumerate(zip(w["W"], w["b"])):
        flat[f"dense_{i}_kernel"], flat[f"dense_{i}_bias"] = k, b
    flat["n_snps"], flat["width"], flat["nlayers"] = (np.int64(w["W"][0].shape[0]), np.int64(w["W"][0].shape[1]),
                                                      np.int64(len(w["W"]) - 2))
    _write_atomic(path, lambda fh: np.savez(fh, **flat), mode="wb")


def read_weights(path):
    """Inverse of save_weights: the oracle-format dict LocatorNet.import_params takes."""
    z = np.load(path)
    n_dense = sum(1 for k in z.files if k.endswith("_kernel"))
    return {"gamma": z["gamma"], "beta": z["beta"], "mov_mean": z["moving_mean"], "mov_var": z["moving_variance"],
            "W": [z[f"dense_{i}_kernel"] for i in range(n_dense)], "b": [z[f"dense_{i}_bias"] for i in range(n_dense)]}


def _device_matrix(model, traingen, testgen, trainlocs, testlocs):
    """Training + validation rows as ONE uint8 matrix in HBM with their z-scored targets; returns (X, Y, train row
    range start, validation row range start)."""
    import torch
    from .net import upload_genotypes
    ntr, nva = traingen.shape[0], testgen.shape[0]
    if isinstance(traingen, DeviceRows):
        assert isinstance(testgen, DeviceRows) and testgen.X is traingen.X
        X, tr0, va0 = traingen.X, traingen.start, testgen.start
    else:
        X = upload_genotypes(np.concatenate([np.asarray(traingen), np.asarray(testgen)], axis=0), model.device)
        tr0, va0 = 0, ntr
    yh = np.zeros((X.shape[0], 2), np.float32)
    yh[tr0:tr0 + ntr] = trainlocs
    yh[va0:va0 + nva] = testlocs
    return X, torch.from_numpy(yh).to(model.device), tr0, va0


GRAPH_MAX_SNPS_IN_FIT_THREADS = 65536      # fit threads sharing a process capture epoch graphs up to this many SNPs (train_network)


def train_network(model, traingen, testgen, trainlocs, testlocs, callbacks, boot=0):
    """model.fit with the three callbacks, then the best-val_loss weights back in the model (locator.py:365-394)."""
    from .train import History, fit
    start = time.time()
    checkpointer, earlystop, reducelr = callbacks
    ntr, nva = traingen.shape[0], testgen.shape[0]
    if nva == 0:
        raise SystemExit("no validation samples: --train_split leaves round((1 - split) * n_located) = 0 of them, "
                         "and every callback of the fit monitors val_loss")
    X, Y, tr0, va0 = _device_matrix(model, traingen, testgen, trainlocs, testlocs)
    model._build(X, Y)
    model.n_train, model.n_val = ntr, nva
    if getattr(args, "load_weights", None):
        try:
            model.net.import_params(read_weights(args.load_weights))     # predict-only: nothing is trained
        except (ValueError, KeyError) as e:
            raise SystemExit(f"--load_weights {args.load_weights}: {e} - the file must come from a run with the same "
                             "genotypes and filters (--min_mac, --max_SNPs, --impute_missing), --width and --nlayers")
        history = History()
    else:
        # Round 5: fit threads that share a process capture their epoch graphs too - train.DEVICE_LOCK keeps a capture apart
        # from a sibling's set-up / read-back / tear-down (round 4 launched them eagerly) - where replay pays: with two fits
        # interleaving on one GPU an eagerly launched epoch costs host time, and that is hidden behind the device once a step's
        # kernels are long.  Measured (profiles/r05_readme_windows.json, r05_config4_workers.txt): 1,150 SNPs per window
        # 0.23 s per fit with graphs against 0.26-0.44 s eager; 150,000 SNPs 1.19 s against 1.17 s (two captures per fit cost
        # more than replay saves).  A fit alone in its process always replays (bench.py, single runs).
        from . import replicates
        use_graph = not args.no_graph and (replicates.fit_threads_in_process() <= 1 or model.net.d.K <= GRAPH_MAX_SNPS_IN_FIT_THREADS)
        history = fit(model.net, np.arange(tr0, tr0 + ntr), np.arange(va0, va0 + nva), batch_size=args.batch_size,
                      max_epochs=args.max_epochs, patience=earlystop["patience"], lr_patience=reducelr["patience"],
                      lr_factor=reducelr["factor"], use_graph=use_graph, verbose=args.keras_verbose,
                      chain=False if getattr(args, "no_chain", False) else None)
    if args.keep_weights:
        w = model.weights_dict()                     # device read-back: under the lock
        with _host_io():
            save_weights(checkpointer["filepath"], w)
    print("run time " + str((time.time() - start) / 60) + " minutes")
    return history, model


def _predlocs_path(boot):
    if args.bootstrap or args.jacknife:
        return f"{_out()}_boot{boot}_predlocs.txt"
    if args.windows:
        # the reference appends the *flag* window to an --out that already carries the real window
        w0, wsize = int(args.window_start), int(args.window_size)
        return f"{_out()}_{w0}-{w0 + wsize - 1}_predlocs.txt"
    return _out() + "_predlocs.txt"


def _to_map_units(z, sdlong, meanlong, sdlat, meanlat):
    """z-scored (n, 2) -> original coordinates, float64 like the reference's Python-float arithmetic."""
    z = np.asarray(z, dtype=np.float64).reshape(-1, 2)
    return z * np.array([sdlong, sdlat], dtype=np.float64) + np.array([meanlong, meanlat], dtype=np.float64)


def write_predlocs(path, xy, sample_ids):
    """`x,y,sampleID` CSV (locator.py:418-433), written atomically."""
    import pandas as pd
    frame = pd.DataFrame({"x": xy[:, 0], "y": xy[:, 1], "sampleID": np.asarray(sample_ids, dtype=object)})
    _write_atomic(path, lambda fh: frame.to_csv(fh, index=False))


def predict_locs(model, predgen, sdlong, meanlong, sdlat, meanlat, testlocs, pred, samples, testgen, history,
                 boot=0, verbose=True):
    """Predict the unknown samples, report the fit on the validation samples in map units, write
    {out}..._predlocs.txt and {out}_history.txt (locator.py:397-470)."""
    import pandas as pd
    if verbose:
        print("predicting locations...")
    # the two device passes first (under the device lock when fits share a process), then the host work without it
    z_pred, z_val = model.predict(predgen), model.predict(testgen)
    with _host_io():
        xy = _to_map_units(z_pred, sdlong, meanlong, sdlat, meanlat)
        write_predlocs(_predlocs_path(boot), xy, np.asarray(samples)[pred] if len(pred) else [])
        truth = _to_map_units(testlocs, sdlong, meanlong, sdlat, meanlat)
        fitted = _to_map_units(z_val, sdlong, meanlong, sdlat, meanlat)
        dists = np.sqrt(((fitted - truth) ** 2).sum(axis=1)).tolist()
        if verbose:
            r2 = [np.corrcoef(fitted[:, a], truth[:, a])[0][1] ** 2 for a in (0, 1)]
            print(f"R2(x)={r2[0]}\nR2(y)={r2[1]}\nmean validation error {np.mean(dists)}\n"
                  f"median validation error {np.median(dists)}\n")
        if len(history.history.get("loss", [])):
            table = pd.DataFrame(history.history)
            _write_atomic(_out() + "_history.txt", lambda fh: table.to_csv(fh, sep="\t", index=False))
    return dists


def plot_history(history, dists):
    """{out}_fitplot.pdf: validation and training loss from the fourth epoch on (locator.py:473-484)."""
    if not args.plot_history or not len(history.history.get("loss", [])):
        return
    import matplotlib
    matplotlib.use("agg")
    from matplotlib import pyplot as plt
    with plt.rc_context({"font.size": 7}):
        fig = plt.figure(dpi=200, figsize=(4, 1.5))
        for left, key, label in ((0.0, "val_loss", "Validation Loss"), (0.55, "loss", "Training Loss")):
            ax = fig.add_axes([left, 0, 0.4, 1])
            ax.plot(history.history[key][3:], "-", color="black", lw=0.5)
            ax.set_xlabel(label)
        fig.savefig(args.out + "_fitplot.pdf", bbox_inches="tight")
        plt.close(fig)


# ------------------------------------------------------------------ replicate units (windows / bootstrap)
_BASE_CACHE = {}      # per worker process: the shared genotype rows, uploaded once per device
_BASE_LOCK = __import__("threading").Lock()


AUTO_FITS_MAX = 3                # fit threads per GPU under --fits_per_gpu 0 ...
AUTO_FITS_SNPS = 70_000          # ... of which three fit at a time up to this many SNPs per fit, two above: the hidden stack of
#                                  a fit is latency-bound on a few compute units and a third fit fills them while layer 1 is
#                                  short (bench.py --replicates-per-gpu 1 / 2 / 3 / 4, samples/s: 5,830 SNPs 421k / 711k / 903k /
#                                  699k; 20,000 SNPs 354k / 529k / 620k / 513k; 100,000 SNPs 190k / 226k / 221k / 212k; 2 against 3
#                                  fits at 35,000 / 50,000 / 65,000 / 80,000 SNPs: 430k / 490k, 356k / 402k, 307k / 324k, 269k / 270k)
_FIT_SLOTS = {}
_FIT_SLOTS_LOCK = __import__("threading").Lock()


class _FitBudget:
    """Admission of the concurrent fits of this process on one device, sized PER UNIT: the device has AUTO_FITS_MAX * 2
    credits; under --fits_per_gpu 0 a fit of up to AUTO_FITS_SNPS SNPs takes 2 of them (three such fits at a time) and a
    larger one 3 (two at a time; one large beside one small), so a --windows run whose windows differ in size admits each
    window by its own SNP count - not by the first window's.  An explicit --fits_per_gpu n admits n fits whatever their
    size.  The flag comes from the unit's own args (in a worker process the module-global `args` is still None when the
    first unit asks)."""

    TOTAL = 2 * AUTO_FITS_MAX

    def __init__(self):
        import threading
        self._cv = threading.Condition()
        self._used = 0          # credits out under the automatic rule
        self._fits = 0          # fits admitted right now (any rule)

    def cost(self, K, a):
        n = int(getattr(a, "fits_per_gpu", 0) or 0)
        if n > 0:
            return 0, max(1, n)                             # (credits, cap on concurrent fits)
        return (2 if K <= AUTO_FITS_SNPS else 3), AUTO_FITS_MAX

    def acquire(self, K, a):
        c, cap = self.cost(K, a)
        with self._cv:
            # (a lone fit is always admitted: credits can never deadlock)
            while self._fits > 0 and (self._fits >= cap or self._used + c > self.TOTAL):
                self._cv.wait()
            self._fits += 1
            self._used += c
        return c

    def release(self, c):
        with self._cv:
            self._fits -= 1
            self._used -= c
            self._cv.notify_all()

    def admitted(self):
        with self._cv:
            return self._fits


def _fit_slots(device):
    """The admission budget of this process's fits on `device` (created at the first unit)."""
    with _FIT_SLOTS_LOCK:
        if device not in _FIT_SLOTS:
            _FIT_SLOTS[device] = _FitBudget()
        return _FIT_SLOTS[device]


def _snps_hint(unit):
    """SNP count of a unit's fit before anything has been filtered: a bootstrap unit's matrix is there, a window unit knows
    its variant range (an upper bound: the filters only remove)."""
    if "traingen" in unit:
        return int(np.shape(unit["traingen"])[1])
    if "gt_shape" in unit:
        return int(unit["gt_shape"][0])
    if "window" in unit:
        return int(unit["window"][1] - unit["window"][0])
    return 0


def _fit_unit(unit, device="cuda:0", reraise=False):
    """_fit_unit_body under the process-wide device lock (train.DEVICE_LOCK): when several fits share a process, one thread
    and stream each, everything a fit does on the device outside its epoch loop - upload, net construction, read-backs,
    predict launches, and the destruction of its graphs / events / buffers when the body returns OR RAISES - is kept apart
    from a sibling's HIP-graph capture.  The epoch loop itself runs with the lock released (FitLoop.run), and so does the
    unit's pure host work (output files, plots: _host_io).  Before that the unit waits for admission (_FitBudget:
    --fits_per_gpu, or 3 / 2 at a time by the unit's own SNP count) - without holding the lock - and only then reports
    ("start", index) to the pool through unit["on_admitted"], so --unit_timeout times the fit, not the wait for a sibling.
    A failed fit comes back as an error record (what the pool collects); reraise=True (the single fit of a plain run) lets
    the exception through instead - after the same teardown under the lock."""
    import traceback
    from .train import DEVICE_LOCK
    budget = _fit_slots(device)
    credits = budget.acquire(_snps_hint(unit), unit.get("args"))
    try:
        admitted = unit.pop("on_admitted", None) if isinstance(unit, dict) else None
        if admitted is not None:
            admitted()
        with DEVICE_LOCK:
            try:
                return _fit_unit_body(unit, device)
            except Exception as e:                      # noqa: BLE001 - becomes the unit's error record (as replicates._run_one's)
                # The traceback's frames hold the model, its graphs, events and device buffers: they must go away HERE,
                # under the lock, not later in a handler that runs while a sibling has a capture open.
                rec = {"name": unit.get("name", "?"), "error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()}
                traceback.clear_frames(e.__traceback__)
                if reraise:
                    raise
                del e
                return rec
    finally:
        budget.release(credits)


_fit_unit.reports_admission = True      # replicates._worker: this fit function sends the unit's ("start", i) itself


def _host_io():
    """Context for a fit's pure host work (files, plots): the device lock is dropped for its duration."""
    from .train import DEVICE_LOCK
    return DEVICE_LOCK.released()


def _fit_unit_body(unit, device="cuda:0"):
    """One replicate fit + predict on `device` — the body of the reference's window / bootstrap loops
    (locator.py:546-571, :656-676).  `unit` carries everything that depended on the NumPy stream.
    The unit's train / validation / prediction rows are uploaded once as one uint8 matrix; a
    bootstrap unit resamples its SNP columns on the device (loc_gather_columns) instead of the
    reference's three host fancy-index copies (locator.py:651-653)."""
    global args
    from .net import gather_columns, upload_genotypes
    args = unit["args"]
    t_unit = time.time()
    phases = {}
    if "window" in unit and "gt_pin" not in unit:
        with _host_io():
            _load_window(unit)      # a no-op when the worker's loader thread has already done it (host_prepare)
    phases["load"] = time.time() - t_unit
    X = None
    if "gt_pin" in unit:
        # the window's raw calls are in pinned memory: upload them as they are, filter + split + transpose on the device
        import torch
        from .net import filter_snps_device
        shape = unit["gt_shape"]
        nbytes = int(np.prod(shape, dtype=np.int64))
        pin = unit.pop("gt_pin")
        gt_dev = pin[:nbytes].to(device, non_blocking=True).view(torch.int8).view(*shape)
        train, test, pred = unit["train"], unit["test"], unit["pred"]
        order = np.concatenate([np.asarray(train), np.asarray(test), np.asarray(pred)]).astype(np.int32)
        X, K = filter_snps_device(gt_dev, order, args.min_mac)
        torch.cuda.current_stream().synchronize()
        _pin_give(pin)
        del gt_dev
        if K < 1:
            raise ValueError(f"{unit['name']}: no SNP passes the filters (biallelic, allele-1 count >= {args.min_mac})")
        ntr, nva, npr = len(train), len(test), len(pred)
        unit.update(trainlocs=unit["locs"][train], testlocs=unit["locs"][test])
        key = None
    else:
        tg, vg, pg = unit["traingen"], unit["testgen"], unit["predgen"]
        ntr, nva, npr, K = tg.shape[0], vg.shape[0], pg.shape[0], tg.shape[1]
        key = (device, id(tg), id(vg), id(pg)) if unit.get("cache_base") else None
    if X is None:
        with _BASE_LOCK:                    # two fit threads of one process share the uploaded base matrix
            X = _BASE_CACHE.get(key) if key else None
            if X is None:
                X = upload_genotypes(np.concatenate([np.asarray(tg), np.asarray(vg), np.asarray(pg).reshape(npr, K)], axis=0),
                                     device)
                if key:
                    _BASE_CACHE.clear()
                    _BASE_CACHE[key] = X
    if unit.get("site_order") is not None:
        X = gather_columns(X, unit["site_order"], K)
    phases["upload"] = time.time() - t_unit - phases["load"]
    traingen, testgen = DeviceRows(X, 0, ntr, K), DeviceRows(X, ntr, nva, K)
    predgen = DeviceRows(X, ntr + nva, npr, K)
    model = load_network(traingen, args.dropout_prop, replicate=unit["replicate"], device=device)
    # every file of this unit hangs off the unit's own stem (for a window: {out}_{start}-{end}; the reference swaps
    # args.out only after training, so its --keep_weights file is overwritten by every window)
    _TLS.out = unit["out"]              # thread-local: another unit may be fitting on another thread / stream of this process
    try:
        callbacks = load_callbacks(unit["boot"])
        t1 = time.time()
        history, model = train_network(model, traingen, testgen, unit["trainlocs"], unit["testlocs"], callbacks,
                                       unit["boot"])
        phases["fit"] = time.time() - t1
        t1 = time.time()
        dists = predict_locs(model, predgen, unit["sdlong"], unit["meanlong"], unit["sdlat"], unit["meanlat"],
                             unit["testlocs"], unit["pred"], unit["samples"], testgen, history, unit["boot"])
        phases["predict"] = time.time() - t1
    finally:
        _TLS.out = None
    return {"name": unit["name"], "history": history.history, "dists": dists, "seconds": time.time() - t_unit,
            "phases": phases, "epochs": len(history.history.get("loss", []))}


_PIN_POOL = []          # per worker process: pinned staging buffers for window slices (torch uint8 tensors), reused
_PIN_LOCK = __import__("threading").Lock()


def _pin_take(nbytes):
    import torch
    with _PIN_LOCK:
        for i, t in enumerate(_PIN_POOL):
            if t.numel() >= nbytes:
                return _PIN_POOL.pop(i)
        _PIN_POOL.clear()                               # too small for this window: let them go, allocate with head room
    # (a pinned allocation is a runtime call the loader thread makes while fit threads may be capturing graphs: it waits for
    # an open capture like every other out-of-loop device call, train.DEVICE_LOCK)
    from .train import DEVICE_LOCK
    with DEVICE_LOCK:
        return torch.empty(int(nbytes * 1.1) + 4096, dtype=torch.uint8).pin_memory()


def _pin_give(t):
    with _PIN_LOCK:
        if len(_PIN_POOL) < 3:
            _PIN_POOL.append(t)


def _read_window(unit):
    """Loader-thread half of a window on a GPU worker: ONLY the zarr slice gt[a:b] (locator.py:539), decoded straight into
    pinned host memory.  Filters (locator.py:265-273) and the split's transposes (:295-308) then run on the device
    (net.filter_snps_device in _fit_unit): the host phase of a 150,000-variant window drops from 0.68 s (NumPy allele counts +
    boolean indexing + three transposed copies of a 230 MB slice) to the read itself."""
    a, b = unit["window"]
    gt = G.open_group(unit["zarr"], mode="r")["calldata/GT"]
    shape = (max(min(b, gt.shape[0]) - a, 0),) + tuple(gt.shape[1:])
    nbytes = int(np.prod(shape, dtype=np.int64))
    pin = _pin_take(nbytes)
    view = pin.numpy()[:nbytes].view(np.int8).reshape(shape)
    gt.read_into(view, a, a + shape[0])
    unit["gt_pin"], unit["gt_shape"] = pin, shape
    return unit


def _load_window_on_loader_thread(unit, a):
    """ReplicatePool host_prepare hook, run by the worker's loader thread while the previous window is still fitting
    (slice + filters used to sit on every fit's critical path, locator.py:539-545): on a GPU worker the zarr slice only
    (_read_window; the filters run on the device), otherwise - scheduler tests without a GPU - the host slice + filters."""
    import torch
    u = dict(unit)
    u["args"] = a
    if torch.cuda.is_available() and not getattr(a, "host_filter", False):
        gt = G.open_group(u["zarr"], mode="r")["calldata/GT"]
        # the device filter takes the store's calls as they are: int8 [variants][samples][ploidy].  Any other dtype or
        # rank goes the host way (np.asarray(..., dtype=int8) as load_genotypes does) instead of failing every window
        if np.dtype(gt.dtype) == np.int8 and len(gt.shape) == 3:
            return _read_window(u)
    return _load_window(u)


def _window_bounds():
    """(window start, size, first SNP index, last SNP index) for every window of the run (locator.py:519-537)."""
    positions = np.asarray(G.open_group(args.zarr, mode="r")["variants/POS"][:])
    first = int(args.window_start)
    width = int(args.window_size)
    last = positions.max() if args.window_stop is None else int(args.window_stop)
    bounds = []
    for lo in range(first, int(last), width):
        inside = np.flatnonzero((positions >= lo) & (positions < lo + width))
        bounds.append((lo, width, int(inside.min()), int(inside.max())))   # an empty window raises, as in the reference
    return bounds


def _window_units(samples, lazy=None):
    """Host prologue of the window loop, in reference order (locator.py:519-545).

    Everything that advances the global NumPy stream happens here, sequentially.  Without
    --impute_missing / --max_SNPs the only draw per window is the split's np.random.choice, which depends
    on the sample file alone — so the splits are drawn up front and the expensive part (zarr slice +
    filters, ~2 s per 150k-variant window) is deferred to the worker that fits the window
    (`_load_window`), otherwise the serial prologue would cap multi-GPU scaling.  With either flag the
    draws depend on the genotypes and the whole prologue runs here, exactly as the reference does."""
    if lazy is None:
        lazy = not args.impute_missing and args.max_SNPs is None
    units = []
    for n, (i, size, a, b) in enumerate(_window_bounds()):
        print(f"\nProcessing window {i}-{i + size}")
        print(f"SNPs {a}-{b}")
        unit = dict(name=f"window {i}-{i + size - 1}", replicate=n, boot=None, out=f"{args.out}_{i}-{i + size - 1}",
                    samples=samples)
        if lazy:
            class _Shape:
                shape = (b - a, len(samples), 2)
            sample_data, locs = sort_samples(samples, _Shape())
            meanlong, sdlong, meanlat, sdlat, locs = normalize_locs(locs)
            train, test, pred = split_indices(locs, args.train_split)
            unit.update(window=(a, b), zarr=args.zarr, train=train, test=test, pred=pred, locs=locs,
                        sdlong=sdlong, meanlong=meanlong, sdlat=sdlat, meanlat=meanlat)
        else:
            unit.update(window=(a, b), zarr=args.zarr)
            _load_window(unit, draw_split=True)
        units.append(unit)
    return units


def _load_window(unit, draw_split=False):
    """Slice the store (gt[a:b]: excludes SNP b, as the reference does), filter, and cut the
    train / validation / prediction rows.  Runs in the worker for lazy units."""
    if "traingen" in unit:
        return unit
    a, b = unit["window"]
    gt = G.open_group(unit["zarr"], mode="r")["calldata/GT"]
    genotypes = np.asarray(gt[a:b, :, :], dtype=np.int8)
    if draw_split:                                   # eager path: reference order sort -> normalise -> filter -> split
        sample_data, locs = sort_samples(unit["samples"], genotypes)
        meanlong, sdlong, meanlat, sdlat, locs = normalize_locs(locs)
        ac = filter_snps(genotypes)
        train, test, pred = split_indices(locs, args.train_split)
        unit.update(sdlong=sdlong, meanlong=meanlong, sdlat=sdlat, meanlat=meanlat)
    else:
        ac = G.filter_snps(genotypes, min_mac=unit["args"].min_mac if "args" in unit else args.min_mac, verbose=False)
        train, test, pred, locs = unit["train"], unit["test"], unit["pred"], unit["locs"]
    unit.update(traingen=G.rows_transposed(ac, train), testgen=G.rows_transposed(ac, test),
                predgen=(G.rows_transposed(ac, pred) if len(pred) else np.zeros((0, ac.shape[0]), ac.dtype)),
                trainlocs=locs[train], testlocs=locs[test], pred=pred)
    return unit


def _bootstrap_units(n_sites):
    """FULL fit + nboots resamples; the reseed / site_order chain is drawn here, sequentially and in
    reference order (locator.py:635-650).  Units are small: the genotype rows travel once as `shared`."""
    units = [dict(name="boot FULL", replicate=0, boot="FULL", site_order=None, cache_base=True)]
    for boot in range(args.nboots):
        # the reference's `np.random.seed(np.random.choice(range(int(1e6)), 1))` (locator.py:637) without building a
        # million-element array per replicate: for an int population the legacy generator makes the very same draw
        # (randint(0, 10^6, 1); value = index) - same seeds, same stream (tests/test_host.py, tests/test_oracle.py known
        # answers), 80 ms -> 0.1 ms per replicate: 20 s of a 256-replicate parent prologue
        np.random.seed(np.random.choice(int(1e6), 1))
        site_order = np.random.choice(n_sites, n_sites, replace=True).astype(np.int32)
        units.append(dict(name=f"boot {boot}", replicate=boot + 1, boot=boot, site_order=site_order,
                          cache_base=True))
    return units


def _prologue(force_full=False):
    """locator.py:507-516.  Returns (samples, state) where state is None in the windows fast path or the
    tuple (meanlong, sdlong, meanlat, sdlat, ac, train, test, traingen, testgen, trainlocs, testlocs, pred, predgen)."""
    if args.windows and not args.impute_missing and args.max_SNPs is None and not force_full:
        # The reference loads and filters the WHOLE store here (locator.py:508-516) and then discards the
        # result: the window loop re-slices, re-filters and re-splits.  All that survives is the state of
        # the global NumPy stream, and without --impute_missing / --max_SNPs the only draw is the split's
        # np.random.choice, which depends on the sample file alone.  Make that draw, skip the 5.7 GB read.
        print("reading zarr")
        callset = G.open_group(args.zarr, mode="r")
        samples = np.asarray(callset["samples"][:])

        class _Shape:
            shape = tuple(callset["calldata/GT"].shape)
        sample_data, locs = sort_samples(samples, _Shape())
        meanlong, sdlong, meanlat, sdlat, locs = normalize_locs(locs)
        split_indices(locs, args.train_split)
        return samples, None
    genotypes, samples = load_genotypes()
    sample_data, locs = sort_samples(samples, genotypes)
    meanlong, sdlong, meanlat, sdlat, locs = normalize_locs(locs)
    ac = filter_snps(genotypes)
    train, test, traingen, testgen, trainlocs, testlocs, pred, predgen = split_train_test(ac, locs)
    return samples, (meanlong, sdlong, meanlat, sdlat, ac, train, test, traingen, testgen, trainlocs, testlocs,
                     pred, predgen)


# ------------------------------------------------------------------ main (locator.py:487-749)
def _print_replicate_summary(pool, results, t_program):
    """Per-phase means over the units, the scheduler's timeline and its Amdahl projection (replicates.ReplicatePool)."""
    ok = [r for r in results if r is not None and "error" not in r and "phases" in r]
    if ok:
        mean = lambda k: sum(r["phases"].get(k, 0.0) for r in ok) / len(ok)
        host = sum(r.get("host_prepare_seconds", 0.0) for r in ok) / len(ok)
        print(f"replicate phases, mean of {len(ok)} units: host slice+filter {host:.2f} s (loader thread), load-in-fit "
              f"{mean('load'):.2f} s, upload {mean('upload'):.2f} s, fit {mean('fit'):.3f} s, predict {mean('predict'):.3f} s")
    for line in pool.summary(results, program_started=t_program)["lines"]:
        print(line)


def _unit_count_bound():
    """Cheap upper bound on the number of replicate units, known before the prologue: the pool never spawns more worker
    processes (each one imports torch, opens a device context and loads the HIP library) than there are units."""
    if args.bootstrap:
        return int(args.nboots) + 1                                   # FULL fit + replicates (locator.py:612-676)
    if args.windows and args.zarr:
        try:
            if args.window_stop is not None:
                last = int(args.window_stop)
            else:
                last = int(np.asarray(G.open_group(args.zarr, mode="r")["variants/POS"][:]).max())
            return max(1, -(-(last - int(args.window_start)) // int(args.window_size)))
        except Exception:                                             # noqa: BLE001 - the prologue reports the real problem
            return None
    return None


def main(argv=None):
    t_program = time.time()
    from . import replicates
    raw = sys.argv[1:] if argv is None else list(argv)
    if (("--windows" in raw or "--bootstrap" in raw) and "--in_process" not in raw
            and not any(a == "spawn" or a == "--worker_start=spawn" for a in raw)):
        # replicate runs: the fork server starts FIRST - its `import torch` overlaps this process's own parsing, imports and
        # prologue, and every worker is then a fork + one device context (replicates.warm_start)
        replicates.warm_start("forkserver")
    _setup(argv)

    pool = None
    if args.windows or args.bootstrap:
        # the worker processes start NOW: spawn + `import torch` + device context + library load overlap the parent's
        # own prologue instead of following it
        lazy_windows = args.windows and not args.impute_missing and args.max_SNPs is None
        # --fits_per_gpu 0 (default): the pool gets AUTO_FITS_MAX fit threads per GPU and _fit_unit admits as many of them at a
        # time as the SNP count of the fits asks for (the pool exists before the genotypes have been read)
        pool = replicates.ReplicatePool(args, _fit_unit, n_gpus=args.gpus, fits_per_gpu=args.fits_per_gpu or AUTO_FITS_MAX,
                                        host_prepare=_load_window_on_loader_thread if lazy_windows else None,
                                        unit_timeout=getattr(args, "unit_timeout", 0),
                                        max_workers=_unit_count_bound(),
                                        procs_per_gpu=getattr(args, "procs_per_gpu", 1),
                                        isolate=True if not getattr(args, "in_process", False) else None).start()
    try:
        return _main_body(pool, t_program)
    finally:
        if pool is not None:
            pool.close()


def _main_body(pool, t_program):
    samples, state = _prologue()
    if state is not None:
        (meanlong, sdlong, meanlat, sdlat, ac, train, test, traingen, testgen, trainlocs, testlocs, pred,
         predgen) = state

    failed = 0
    if args.windows:
        units = _window_units(samples)
        results = pool.run(units)
        _print_replicate_summary(pool, results, t_program)
        for r in results:
            if "error" in r:
                print(f"{r['name']}: FAILED: {r['error']}")
            else:
                print(f"{r['name']}: run time {r['seconds'] / 60:.2f} minutes")
        failed = sum("error" in r for r in results)
        results = [r for r in results if "error" not in r]
        if results:                             # fitplot is overwritten per window in the reference: last one wins
            plot_history(_H(results[-1]["history"]), results[-1]["dists"])
    elif not args.bootstrap and not args.jacknife:
        unit = dict(name="single", replicate=0, boot=0, out=args.out, traingen=traingen, testgen=testgen,
                    predgen=predgen, trainlocs=trainlocs, testlocs=testlocs, pred=pred, samples=samples,
                    sdlong=sdlong, meanlong=meanlong, sdlat=sdlat, meanlat=meanlat, args=args)
        r = _fit_unit(unit, reraise=True)
        plot_history(_H(r["history"]), r["dists"])
    elif args.bootstrap:
        units = _bootstrap_units(traingen.shape[1])
        shared = dict(traingen=np.ascontiguousarray(traingen), testgen=np.ascontiguousarray(testgen),
                      predgen=np.ascontiguousarray(predgen), trainlocs=trainlocs, testlocs=testlocs, pred=pred,
                      samples=samples, sdlong=sdlong, meanlong=meanlong, sdlat=sdlat, meanlat=meanlat, out=args.out)
        results = pool.run(units, shared)
        _print_replicate_summary(pool, results, t_program)
        failed = sum("error" in r for r in results)
        # {out}_history.txt / fitplot are overwritten by every replicate in the reference: last one wins
        results = [r for r in results if "error" not in r]
        if results:
            import pandas as pd
            last = pd.DataFrame(results[-1]["history"])
            _write_atomic(args.out + "_history.txt", lambda fh: last.to_csv(fh, sep="\t", index=False))
            plot_history(_H(results[-1]["history"]), results[-1]["dists"])
    elif args.jacknife:
        _jacknife(ac, traingen, testgen, trainlocs, testlocs, predgen, pred, samples, sdlong, meanlong, sdlat, meanlat)
    if failed:
        print(f"{failed} replicate fit(s) FAILED", file=sys.stderr)
        return 1
    return 0


def jacknife_draws(predgen, af, nboots, prop):
    """The reference's jacknife randomness (locator.py:713-727), replicate after replicate from the global NumPy
    stream: the sites to redraw (choice without replacement), then for each of them, in that order, one
    Binomial(2, site frequency) per prediction sample.  The per-site calls of the reference are issued as ONE
    broadcast call per replicate: the legacy generator fills a broadcast request element by element in row-major
    order through the same scalar routine, so the stream - and every value - is the same (tests/test_host.py).
    Returns [(sites, values[n_sites][n_pred])]."""
    n_pred, K = predgen.shape
    out = []
    for _ in range(nboots):
        sites = np.random.choice(K, int(K * prop), replace=False)
        vals = np.random.binomial(2, np.asarray(af)[sites][:, None], (len(sites), n_pred)) if len(sites) else \
            np.zeros((0, n_pred), np.int64)
        out.append((sites, vals))
    return out


def _jacknife(ac, traingen, testgen, trainlocs, testlocs, predgen, pred, samples, sdlong, meanlong, sdlat, meanlat):
    """--jacknife (locator.py:683-747): one fit, then nboots re-predictions of the unknown samples with a fraction of
    their SNPs redrawn from the site frequencies.  The reference predicts replicate by replicate; here the nboots
    perturbed copies form ONE matrix of nboots x n_pred rows that goes up once and through one many-row predict (the
    first layer as a large-M GEMM against a weight image converted once)."""
    t_fit = time.time()
    history, model = train_network(load_network(traingen, args.dropout_prop), traingen, testgen, trainlocs, testlocs,
                                   load_callbacks("FULL"), "FULL")
    plot_history(history, predict_locs(model, predgen, sdlong, meanlong, sdlat, meanlat, testlocs, pred, samples,
                                       testgen, history, "FULL"))
    print("run time " + str((time.time() - t_fit) / 60) + " minutes")
    print("starting jacknife resampling")
    base = np.ascontiguousarray(np.asarray(predgen)).astype(np.uint8, copy=False)
    n_pred = base.shape[0]
    if n_pred == 0 or args.nboots < 1:
        return
    af = ac.sum(axis=1) / (ac.shape[1] * 2)
    draws = jacknife_draws(base, af, args.nboots, args.jacknife_prop)
    ids = np.asarray(samples)[pred]
    group = max(1, int(2e9 // max(1, base.size)))          # replicates per upload: at most ~2 GB of host staging
    for b0 in range(0, args.nboots, group):
        nb = min(group, args.nboots - b0)
        stacked = np.tile(base, (nb, 1))
        for b in range(nb):
            sites, vals = draws[b0 + b]
            stacked[b * n_pred:(b + 1) * n_pred, sites] = vals.T
        xy = _to_map_units(model.predict(stacked), sdlong, meanlong, sdlat, meanlat)
        for b in range(nb):
            write_predlocs(_predlocs_path(b0 + b), xy[b * n_pred:(b + 1) * n_pred], ids)


class _H:
    def __init__(self, history):
        self.history = history


if __name__ == "__main__":
    sys.exit(main())
