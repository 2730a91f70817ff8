#!/usr/bin/env python3
"""Where do 2-bit packed genotypes pay in the int8 many-row layer-1 GEMM?  (the fixed "pack from 3072
rows" rule picked the slower variant at 4096 rows on the driver's box.)

    python tools/gemm_packed_crossover.py [--out gpurun_out/r06_gemm_packed_crossover.jsonl]

Per row count, bytes against packed INTERLEAVED replay by replay in one process (tools/l1_gemm_sweep._time_graphed_ab: medians of
9 replays of a 10-launch graph each), for rows that stream from HBM (distinct rows) and for rows that repeat a 1000-row matrix
(the batched --jacknife / a second predict: the genotype lines come from L2 / the Infinity Cache).  One JSON line per shape."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--snps", type=int, default=100_000)
    ap.add_argument("--rows", default="1000,2048,3072,4096,6144,8192,12288,16384")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--digits", type=int, default=2)
    a = ap.parse_args()
    import numpy as np
    import torch
    from locator_amd import _lib
    from locator_amd.net import LocatorNet, upload_genotypes
    from locator_amd.synth import normalize_locs, synth_genotypes
    from tools.l1_gemm_sweep import _time_graphed_ab, distinct_rows
    dev = "cuda:0"
    x, locs = synth_genotypes(1000, a.snps, seed=20260101, n_na=100)
    X = upload_genotypes(x, dev)
    Y = torch.from_numpy(np.nan_to_num(normalize_locs(locs)[4]).astype(np.float32)).to(dev)
    net = LocatorNet(X, Y, a.snps, 256, 10, 0.25, seed=12345, device=dev)
    lib, d, lay = net.lib, net.d, net.lay
    P = net.params.data_ptr()
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bn4 = torch.zeros(4 * d.Kp, device=dev)
    _lib.check(lib.loc_bn_infer_scale_shift(d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta, P + 4 * lay.mov_mean,
                                            P + 4 * lay.mov_var, bn4.data_ptr(), st()))
    partial = torch.empty(256 * 128 * d.Hp, device=dev)
    image = torch.empty(lib.loc_l1_image_i8_bytes(C.byref(d), a.digits), dtype=torch.uint8, device=dev)
    _lib.check(lib.loc_l1_image_i8_build(C.byref(d), bn4.data_ptr(), P + 4 * lay.w1, a.digits, image.data_ptr(), st()))
    xd = distinct_rows(dev, d.Kp, 16384)
    fout = open(a.out, "w") if a.out else None

    def pack(Xm):
        X2 = torch.zeros((Xm.shape[0], d.Kp // 4), dtype=torch.uint8, device=dev)
        _lib.check(lib.loc_pack_genotypes_2bit(Xm.data_ptr(), Xm.stride(0), Xm.shape[0], d.Kp, X2.data_ptr(), X2.stride(0), st()))
        return X2

    xd2, xm2 = pack(xd), pack(net.X)
    for n_rows in [int(v) for v in a.rows.split(",")]:
        for label, Xm, X2, n_src in (("distinct", xd, xd2, min(n_rows, 16384)), ("repeating_1000", net.X, xm2, 1000)):
            rows = (torch.arange(n_rows, dtype=torch.int32, device=dev) % n_src).contiguous()
            a1 = torch.empty(((n_rows + 127) // 128 * 128, d.Hp), device=dev)
            run = lambda: _lib.check(lib.loc_l1_forward_gemm_i8(Xm.data_ptr(), Xm.stride(0), rows.data_ptr(), n_rows, C.byref(d),
                                                                image.data_ptr(), a.digits, 2, P + 4 * lay.b1, partial.data_ptr(),
                                                                partial.numel(), a1.data_ptr(), 0, None, st()))
            runp = lambda: _lib.check(lib.loc_l1_forward_gemm_i8_packed(X2.data_ptr(), X2.stride(0), rows.data_ptr(), n_rows,
                                                                        C.byref(d), image.data_ptr(), a.digits, P + 4 * lay.b1,
                                                                        partial.data_ptr(), partial.numel(), a1.data_ptr(), 0,
                                                                        None, st()))
            sb, sp = _time_graphed_ab([run, runp], a.iters)
            fl = 2.0 * n_rows * d.K * d.H
            rec = {"rows": n_rows, "source": label, "bytes_us": round(sb["median"], 1), "bytes_min_max": [round(sb["min"], 1), round(sb["max"], 1)],
                   "packed_us": round(sp["median"], 1), "packed_min_max": [round(sp["min"], 1), round(sp["max"], 1)],
                   "packed_over_bytes": round(sp["median"] / sb["median"], 3),
                   "frac_bytes": round(fl / sb["median"] * 1e-6 / 2500.0, 4), "frac_packed": round(fl / sp["median"] * 1e-6 / 2500.0, 4)}
            line = json.dumps(rec)
            print(line, flush=True)
            if fout:
                fout.write(line + "\n")
    if fout:
        fout.close()


if __name__ == "__main__":
    main()
