#!/usr/bin/env python3
"""Generates tests/golden/oracle_fixture_fits.json: the distribution, over seeds, of the end-of-fit validation
metrics of the CPU oracle (oracle/locator_oracle.py, float32, callback-driven `fit` with best-weight reload) on the
reference's own example data (tests/golden/test_genotypes.vcf.gz + test_sample_data.txt: K = 5,830 after filtering,
405 / 45 / 50 split with --seed 12345, exactly the CLI's default run).  The reference never seeds TensorFlow
(locator.py:170-171), so bit parity of a whole fit with Keras is undefined; what CAN be compared is this
distribution (SURVEY.md §0.4 iii "statistical parity"): tests/test_gpu_cli.py checks that HIP fits with different
net seeds land inside it.  "Restated-oracle" data, NOT outputs of the reference (oracle header: parity unpinned).

    python tests/golden/make_statistical.py [n_seeds]        (~40 s per seed on 8 cores)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from locator_amd import genotypes as G  # noqa: E402
from oracle import locator_oracle as O  # noqa: E402


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    import pandas as pd
    v = G.read_vcf(os.path.join(HERE, "test_genotypes.vcf.gz"))
    sd = pd.read_csv(os.path.join(HERE, "test_sample_data.txt"), sep="\t").set_index("sampleID").reindex(v["samples"])
    locs = sd[["x", "y"]].to_numpy()
    meanlong, sdlong, meanlat, sdlat, z = O.normalize_locs(locs)
    ac = G.filter_snps(v["calldata/GT"], min_mac=2, verbose=False)
    np.random.seed(12345)                                    # the CLI's --seed 12345 split
    train, test, traingen, testgen, trainlocs, testlocs, pred, predgen = O.split_train_test(ac, z)
    rows = []
    for s in range(n_seeds):
        t0 = time.time()
        rng = np.random.default_rng(1000 + s)
        p = O.init_params(traingen.shape[1], 256, 10, rng, dtype=np.float32)
        hist, best = O.fit(p, traingen, trainlocs.astype(np.float32), testgen, testlocs.astype(np.float32),
                           batch_size=32, max_epochs=5000, patience=100, drop_p=0.25,
                           perm_fn=lambda e: rng.permutation(traingen.shape[0]),
                           mask_fn=lambda e, st, nb: rng.random((nb, 256)) >= 0.25)
        p2 = O.denormalize(O.predict(best, testgen), sdlong, meanlong, sdlat, meanlat)
        t2 = O.denormalize(testlocs, sdlong, meanlong, sdlat, meanlat)
        r2x, r2y, mean_d, med_d, _ = O.validation_metrics(p2, t2)
        rows.append({"seed": 1000 + s, "epochs": len(hist["loss"]), "best_val_loss": float(min(hist["val_loss"])),
                     "r2_x": float(r2x), "r2_y": float(r2y), "mean_err": mean_d, "median_err": med_d})
        print(rows[-1], f"{time.time() - t0:.0f} s", flush=True)
    out = {"source": "tests/golden/make_statistical.py: fp32 NumPy oracle fits (restated Keras semantics), not the reference",
           "data": "tests/golden/test_genotypes.vcf.gz, --seed 12345 split, defaults (width 256, nlayers 10, dropout "
                   "0.25, batch 32, patience 100)", "fits": rows}
    json.dump(out, open(os.path.join(HERE, "oracle_fixture_fits.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
