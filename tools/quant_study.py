"""How do the many-row predict modes hold their tolerance on TRAINED weights?  (round 4)

Trains the metric's 1000 x 100,000 synthetic fit to convergence on the GPU (callback-driven, the reference's defaults,
locator.py:330-376), exports the weights and evaluates, against the float64 forward of the oracle:
  * what the device's predict modes deliver (exact / fast / bf16 pieces),
  * NumPy emulations of fixed-point weight formats for the int8 matrix pipe, so a format can be judged before a kernel
    is written for it:  per-unit scale (what l1_gemm_i8.hip ships), and a rank-1 scale delta_h * c_k with an integer
    per-SNP factor c_k <= 127 / x_max carried by the GENOTYPE operand (a diploid genotype uses 2 of an int8's 7 bits).
Prints one JSON line per item; --out writes them to a file.  Measurement tool: imports the oracle as the checker.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pow2_delta(mx, bits):
    """delta = 2^e with mx / delta inside the signed range of `bits`-bit base-256 digits (l1_gemm_i8.hip, digit_delta)."""
    mx = np.asarray(mx, np.float64)
    lim = {16: 32639.0, 24: 8355711.0, 8: 127.0}[bits]
    f, x = np.frexp(np.where(mx > 0, mx, 1.0))
    e = x - (bits - 1)
    e = e + (np.ldexp(f, bits - 1) > lim - 1.0)
    return np.where(mx > 0, np.ldexp(1.0, e), 1.0)


def quant_per_unit(wp, bits):
    d = pow2_delta(np.abs(wp).max(axis=0), bits)
    return np.rint(wp / d) * d


def row_factors(wp, x_max, cmax=None):
    """Integer per-SNP factors c_k in 1..floor(127 / x_max): rows are levelled to the largest row's magnitude / cmax."""
    cmax = int(127 // max(1, x_max)) if cmax is None else cmax
    r = np.abs(wp).max(axis=1)
    R = r.max()
    c = np.clip(np.floor(r / (R / cmax)), 1, cmax)
    return c


def quant_rank1(wp, bits, x_max, cmax=None):
    c = row_factors(wp, x_max, cmax)
    ws = wp / c[:, None]
    d = pow2_delta(np.abs(ws).max(axis=0), bits)
    return np.rint(ws / d) * d * c[:, None], c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--snps", type=int, default=100_000)
    ap.add_argument("--max_epochs", type=int, default=5000)
    ap.add_argument("--patience", type=int, default=100)
    ap.add_argument("--out", default=None)
    ap.add_argument("--weights", default=None, help="read trained weights (npz from --save) instead of training")
    ap.add_argument("--save", default=None)
    a = ap.parse_args()
    import torch

    from locator_amd.net import LocatorNet, upload_genotypes
    from locator_amd.synth import normalize_locs, split_indices, synth_genotypes
    from locator_amd.train import fit
    from oracle import locator_oracle as O

    lines = []

    def emit(**kw):
        s = json.dumps(kw)
        print(s, flush=True)
        lines.append(s)

    x, locs = synth_genotypes(a.n, a.snps, seed=20260101, n_na=a.n // 10)
    train, test, pred = split_indices(locs, 0.9, seed=12345)
    _, _, _, _, ynorm = normalize_locs(locs)
    X = upload_genotypes(x)
    Y = torch.from_numpy(np.nan_to_num(ynorm).astype(np.float32)).cuda()
    K, H = a.snps, 256

    def make_net(**kw):
        return LocatorNet(X, Y, K, H, 10, 0.25, seed=12345, **kw)

    net = make_net()
    t0 = time.perf_counter()
    hist = fit(net, train, test, max_epochs=a.max_epochs, patience=a.patience)
    torch.cuda.synchronize()
    ne = len(hist.history["loss"])
    emit(item="fit", epochs=ne, seconds=round(time.perf_counter() - t0, 3), val_loss_best=min(hist.history["val_loss"]),
         loss_last=hist.history["loss"][-1], lr_last=hist.history["learning_rate"][-1])
    p32 = net.export_params()
    p = O.cast_params(p32, np.float64)

    # ---- the weight the int8 image carries: w'[k][h] = s_k W1[k][h]
    s = p["gamma"] / np.sqrt(p["mov_var"] + O.BN_EPS)
    t = p["beta"] - p["mov_mean"] * s
    W1 = p["W"][0]
    wp = W1 * s[:, None]
    colmax, colrms = np.abs(wp).max(0), np.sqrt((wp ** 2).mean(0))
    rowmax = np.abs(wp).max(1)
    emit(item="weights", s_min=float(s.min()), s_med=float(np.median(s)), s_max=float(s.max()),
         mov_var_min=float(p["mov_var"].min()), mov_var_med=float(np.median(p["mov_var"])),
         W1_absmax=float(np.abs(W1).max()), W1_rms=float(np.sqrt((W1 ** 2).mean())),
         glorot_limit=float(np.sqrt(6.0 / (K + H))),
         unit_max_over_rms_med=float(np.median(colmax / colrms)), unit_max_over_rms_max=float((colmax / colrms).max()),
         row_max_spread=float(rowmax.max() / np.median(rowmax)), row_max_p99_over_med=float(np.quantile(rowmax, 0.99) / np.median(rowmax)))

    def study(p, p32, tag):
        s = p["gamma"] / np.sqrt(p["mov_var"] + O.BN_EPS)
        W1 = p["W"][0]
        wp = W1 * s[:, None]
        ref = O.predict(p, x, batch=250)
        pmax = float(np.abs(ref).max())

        def report(name, yhat, **extra):
            dev = np.abs(np.asarray(yhat, np.float64) - ref)
            emit(item="predict", case=tag, mode=name, max_abs=float(dev.max()), rel_to_max_pred=float(dev.max() / pmax),
                 rms_abs=float(np.sqrt((dev ** 2).mean())), **extra)

        # the noise predictor a device pass could evaluate from W1 and the BatchNorm moving statistics alone:
        #   noise_h^2 = delta_h^2 / 12 * sum_k E[x_k^2],  signal_h^2 = sum_k w'_kh^2 Var[x_k]   (independent SNPs)
        ex2 = p["mov_var"] + p["mov_mean"] ** 2
        sig_ind = np.sqrt((wp ** 2 * p["mov_var"][:, None]).sum(0))
        xf = x.astype(np.float64)
        z_ref = xf @ wp
        sig_meas = z_ref.std(0)
        for bits in (16, 24):
            d = pow2_delta(np.abs(wp).max(axis=0), bits)
            noise = d / np.sqrt(12.0) * np.sqrt(ex2.sum())
            zq = xf @ (np.rint(wp / d) * d)
            noise_meas = np.sqrt(((zq - z_ref) ** 2).mean(0))
            emit(item="predictor", case=tag, bits=bits, rel_ind_max=float((noise / sig_ind).max()),
                 rel_ind_med=float(np.median(noise / sig_ind)), rel_meas_sig_max=float((noise / sig_meas).max()),
                 noise_pred_over_measured_med=float(np.median(noise / noise_meas)),
                 noise_pred_over_measured_min=float((noise / noise_meas).min()),
                 true_rel_max=float((noise_meas / sig_meas).max()), true_rel_med=float(np.median(noise_meas / sig_meas)))
        rows_t = torch.arange(a.n, dtype=torch.int32, device="cuda")
        for name, kw in [("dev int8x3 (exact)", {"predict_digits": 3}), ("dev int8x2 (fast)", {"predict_digits": 2}),
                         ("dev bf16x3", {"predict_digits": -1, "predict_pieces": 3}),
                         ("dev bf16x2", {"predict_digits": -1, "predict_pieces": 2})]:
            n2 = make_net(**kw)
            n2.import_params(p32)
            yh = torch.zeros((a.n, 2), device="cuda")
            n2.predict_rows(rows_t, a.n, yh)
            torch.cuda.synchronize()
            report(name, yh.cpu().numpy())
            del n2

    study(p, p32, "trained")
    # constructed heavy tails (VERDICT r03, next #1a): one SNP row per unit F x its trained value, 1 % of the SNPs with a
    # moving variance of 1e-3
    rng = np.random.default_rng(5)
    for F in (30.0, 100.0, 1000.0):
        q32 = {k: ([w.copy() for w in v] if isinstance(v, list) else v.copy()) for k, v in p32.items()}
        ks = rng.choice(K, H, replace=False)
        for h in range(H):
            q32["W"][0][ks[h], h] *= np.float32(F)
        rare = rng.choice(K, K // 100, replace=False)
        q32["mov_var"][rare] = np.float32(1e-3)
        study(O.cast_params(q32, np.float64), q32, f"outliers x{F:g}")

    rows_all = np.arange(a.n)
    ref = O.predict(p, x, batch=250)
    pmax = float(np.abs(ref).max())

    def report(name, yhat, **extra):
        dev = np.abs(np.asarray(yhat, np.float64) - ref)
        emit(item="predict", mode=name, max_abs=float(dev.max()), rel_to_max_pred=float(dev.max() / pmax),
             rms_abs=float(np.sqrt((dev ** 2).mean())), **extra)

    # ---- device modes
    rows_t = torch.arange(a.n, dtype=torch.int32, device="cuda")
    for name, kw in [("dev int8x3 (exact)", {"predict_digits": 3}), ("dev int8x2 (fast)", {"predict_digits": 2}),
                     ("dev bf16x3", {"predict_digits": -1, "predict_pieces": 3}),
                     ("dev bf16x2", {"predict_digits": -1, "predict_pieces": 2}),
                     ("dev bf16x1", {"predict_digits": -1, "predict_pieces": 1}),
                     ("dev fp32 32-row kernels", {"predict_digits": -1, "predict_pieces": -1})]:
        n2 = make_net(**kw)
        n2.import_params(p32)
        yh = torch.zeros((a.n, 2), device="cuda")
        n2.predict_rows(rows_t, a.n, yh)
        torch.cuda.synchronize()
        report(name, yh.cpu().numpy())
        del n2

    # ---- emulated formats: replace W1 by a quantised w'/s (so the oracle's forward multiplies by s again)
    xf = x.astype(np.float64)
    x_max = int(x.max())
    z_ref = xf @ wp

    def with_wq(wq, name, **extra):
        z = xf @ wq
        zerr = np.abs(z - z_ref)
        q = O.copy_params(p)
        q["W"][0] = wq / s[:, None]
        yh = O.predict(q, x, batch=250)
        report(name, yh, z1_max_abs=float(zerr.max()), z1_rms=float(np.sqrt((zerr ** 2).mean())),
               z1_ref_rms=float(np.sqrt((z_ref ** 2).mean())), **extra)

    with_wq(quant_per_unit(wp, 24), "emu per-unit 24-bit")
    with_wq(quant_per_unit(wp, 16), "emu per-unit 16-bit")
    for cmax in (None, 31, 15, 7):
        wq, c = quant_rank1(wp, 16, x_max, cmax)
        with_wq(wq, f"emu rank-1 16-bit cmax={cmax or 127 // x_max}", c_mean=float(c.mean()), c_med=float(np.median(c)),
                c_max=float(c.max()), sum_c=float(c.sum()))
    wq, c = quant_rank1(wp, 8, x_max)
    with_wq(wq, "emu rank-1 8-bit (ONE digit plane)")
    with_wq(quant_per_unit(wp, 8), "emu per-unit 8-bit (ONE digit plane)")
    # bf16 round-to-nearest weights for scale
    w32 = wp.astype(np.float32)
    bits = w32.view(np.uint32)
    bf = (((bits.astype(np.uint64) + 0x7FFF + ((bits >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
    with_wq(bf.astype(np.float64), "emu bf16 RN weights")

    if a.save:
        np.savez(a.save, **{f"W{i}": w for i, w in enumerate(p32["W"])}, **{f"b{i}": b for i, b in enumerate(p32["b"])},
                 gamma=p32["gamma"], beta=p32["beta"], mov_mean=p32["mov_mean"], mov_var=p32["mov_var"])
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
