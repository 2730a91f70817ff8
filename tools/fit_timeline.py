"""Wall clock of one callback-driven fit next to the GPU time of its kernels (round 4, VERDICT r03 missing #4 / weak #5).

  python3 tools/fit_timeline.py --fixture            the reference's example data (configs[1]: 405/45 x 5,830 SNPs)
  python3 tools/fit_timeline.py --n 765 --snps 150016  one window-sized fit (configs[3])
Prints one JSON line {epochs, fit_s, ms_per_epoch, ...}.  Run it under `rocprofv3 --kernel-trace --stats` and feed the
kernel_stats CSV to --stats-csv of a second call (--report) to get the idle fraction = 1 - kernel time / wall.
"""
import argparse
import csv
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", action="store_true")
    ap.add_argument("--n", type=int, default=765)
    ap.add_argument("--snps", type=int, default=150_016)
    ap.add_argument("--max_epochs", type=int, default=5000)
    ap.add_argument("--patience", type=int, default=100)
    ap.add_argument("--sync", action="store_true", help="the synchronous per-epoch loop (fit(pipelined=False))")
    ap.add_argument("--report", default=None, help="JSON line of an earlier run: combine with --stats-csv")
    ap.add_argument("--stats-csv", default=None)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    if a.report:
        r = json.loads(open(a.report).read().strip().splitlines()[-1])
        tot_ns = 0
        rows = []
        for row in csv.DictReader(open(a.stats_csv)):
            tot_ns += int(row["TotalDurationNs"])
            rows.append((int(row["TotalDurationNs"]), row["Name"][:60], int(row["Calls"])))
        rows.sort(reverse=True)
        r["kernel_s_whole_process"] = tot_ns / 1e9
        r["gpu_idle_frac_of_fit"] = 1.0 - min(1.0, tot_ns / 1e9 / r["fit_s"])
        r["top_kernels"] = [{"name": n, "total_ms": t / 1e6, "calls": c} for t, n, c in rows[:6]]
        print(json.dumps(r))
        return
    import inspect

    import torch

    from locator_amd.net import LocatorNet, upload_genotypes
    from locator_amd.train import fit

    if a.fixture:
        from locator_amd import genotypes as G
        from locator_amd.synth import normalize_locs, split_indices
        import pandas as pd
        gold = os.path.join(ROOT, "tests", "golden")
        vcf = G.read_vcf(os.path.join(gold, "test_genotypes.vcf.gz"))
        gt, samples = vcf["calldata/GT"], vcf["samples"]
        sd = pd.read_csv(os.path.join(gold, "test_sample_data.txt"), sep="\t").set_index("sampleID").loc[list(samples)]
        locs = np.array(sd[["x", "y"]], dtype=np.float64)
        ac = G.filter_snps(gt, min_mac=2, verbose=False)
        x = np.ascontiguousarray(ac.T)
    else:
        from locator_amd.synth import normalize_locs, split_indices, synth_genotypes
        x, locs = synth_genotypes(a.n, a.snps, seed=20260104, n_na=a.n // 10)
    train, test, pred = split_indices(locs, 0.9, seed=12345)
    _, _, _, _, ynorm = normalize_locs(locs)
    X = upload_genotypes(x)
    Y = torch.from_numpy(np.nan_to_num(ynorm).astype(np.float32)).cuda()
    net = LocatorNet(X, Y, x.shape[1], 256, 10, 0.25, seed=12345)
    torch.cuda.synchronize()
    kw = {}
    if "pipelined" in inspect.signature(fit).parameters:
        kw["pipelined"] = not a.sync
    t0 = time.perf_counter()
    hist = fit(net, train, test, max_epochs=a.max_epochs, patience=a.patience, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ne = len(hist.history["loss"])
    steps = (len(train) + 31) // 32
    print(json.dumps({"tag": a.tag, "workload": "fixture 405/45 x %d" % x.shape[1] if a.fixture else f"{a.n} x {x.shape[1]}",
                      "mode": kw.get("pipelined", False) and "pipelined" or "synchronous", "epochs": ne, "fit_s": round(dt, 4),
                      "ms_per_epoch": round(1e3 * dt / ne, 4), "us_per_step_incl_validation": round(1e6 * dt / ne / steps, 2),
                      "steps_per_epoch": steps, "val_loss_best": min(hist.history["val_loss"])}))


if __name__ == "__main__":
    main()
