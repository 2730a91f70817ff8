"""Times the large-M layer-1 forward (loc_l1_forward_rows) against the bf16-MFMA and HBM rooflines.

    python tools/rows_gemm_bench.py [--snps 100000] [--width 256] [--rows 1000,450,90] [--iters 50]

flops = 2*M*K*H (the contraction the reference's model.predict performs, counted once however many
bf16 pieces carry each weight); bytes = M*K (uint8 genotypes) + 4*K*H (fp32 weights read once).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locator_amd import _lib  # noqa: E402
from locator_amd.net import LocatorNet  # noqa: E402

BF16_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=100000)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--rows", default="1000,450,90")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--blocks", default="0")
    ap.add_argument("--gemm-only", action="store_true", help="only the image-based GEMMs (l1_gemm.hip, l1_gemm_i8.hip)")
    ap.add_argument("--i8-only", action="store_true", help="only the int8 GEMM (l1_gemm_i8.hip)")
    ap.add_argument("--matrix-rows", type=int, default=0,
                    help="size of the genotype matrix the rows are drawn from (rows = arange(n) %% matrix_rows): the batched "
                         "--jacknife re-reads a small matrix, so its lines come from L2 / MALL; 0 = every row distinct")
    ap.add_argument("--lib", default=None, help="another build of liblocator_hip.so (timing ablations)")
    ap.add_argument("--packed", action="store_true",
                    help="int8 GEMM on 2-bit packed genotypes (loc_pack_genotypes_2bit + loc_l1_forward_gemm_i8_packed); the "
                         "result is checked bit for bit against the unpacked call")
    ap.add_argument("--unit-tiles", type=int, default=0,
                    help="int8 GEMM: loc_tuning.gemm_i8_unit_tiles (1 = eight waves per workgroup, 2 = four waves with 512 registers)")
    a = ap.parse_args()
    if a.lib:
        _lib.use_library(os.path.abspath(a.lib))
    dev = torch.device("cuda:0")
    n_max = max(int(r) for r in a.rows.split(","))
    if a.matrix_rows:
        n_max = min(n_max, a.matrix_rows)
    g = torch.Generator(device="cpu").manual_seed(1)
    X = (torch.rand((n_max, (a.snps + 31) // 32 * 32), generator=g) < 0.3).to(torch.uint8).to(dev)
    Y = torch.zeros((n_max, 2), device=dev)
    net = LocatorNet(X, Y, a.snps, a.width, 10, 0.25, seed=1)
    lib, d, lay = net.lib, net.d, net.lay
    P = net.params.data_ptr()
    bn4 = torch.zeros(4 * d.Kp, device=dev)
    _lib.check(lib.loc_bn_infer_scale_shift(d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta, P + 4 * lay.mov_mean,
                                            P + 4 * lay.mov_var, bn4.data_ptr(), None))
    partial = torch.empty(512 * 128 * d.Hp, device=dev)
    out = []
    for n in ([] if a.gemm_only or a.i8_only else [int(r) for r in a.rows.split(",")]):
        rows = (torch.arange(n, dtype=torch.int32, device=dev) % n_max).contiguous()
        a1 = torch.empty(((n + 127) // 128 * 128, d.Hp), device=dev)
        for blocks in [int(b) for b in a.blocks.split(",")]:
            for pieces in (3, 2, 1):
                if not lib.loc_l1_rows_supported(d.Hp, pieces):
                    continue

                def run():
                    _lib.check(lib.loc_l1_forward_rows(X.data_ptr(), X.stride(0), rows.data_ptr(), n, C.byref(d),
                                                       bn4.data_ptr(), P + 4 * lay.w1, P + 4 * lay.b1,
                                                       partial.data_ptr(), partial.numel(), a1.data_ptr(), pieces,
                                                       blocks, None, None))
                for _ in range(5):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(a.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / a.iters
                flops = 2.0 * n * a.snps * a.width
                byts = n * a.snps + 4.0 * a.snps * a.width
                rec = {"rows": n, "snps": a.snps, "width": a.width, "pieces": pieces, "blocks": blocks,
                       "us": round(us, 2),
                       "tflops": round(flops / us * 1e-6, 1), "frac_bf16_peak": round(flops / us * 1e-6 / BF16_PEAK_TFLOPS, 4),
                       "mfma_issue_frac": round(pieces * flops / us * 1e-6 / BF16_PEAK_TFLOPS, 4),
                       "gbs": round(byts / us * 1e-3, 1), "frac_hbm_peak": round(byts / us * 1e-3 / HBM_PEAK_GBS, 4)}
                print(json.dumps(rec), flush=True)
                out.append(rec)
    # int8 image + GEMM (l1_gemm_i8.hip): digits = 3 exact (1.5 bf16-MFMA equivalents per product), 2 fast (1)
    for n in [int(r) for r in a.rows.split(",")]:
        rows = (torch.arange(n, dtype=torch.int32, device=dev) % n_max).contiguous()
        a1 = torch.empty(((n + 127) // 128 * 128, d.Hp), device=dev)
        for digits in (3, 2):
            if not lib.loc_l1_gemm_i8_supported(d.Hp, digits):
                continue
            image = torch.empty(lib.loc_l1_image_i8_bytes(C.byref(d), digits), dtype=torch.uint8, device=dev)

            def prep():
                _lib.check(lib.loc_l1_image_i8_build(C.byref(d), bn4.data_ptr(), P + 4 * lay.w1, digits,
                                                     image.data_ptr(), None))
            for blocks in [int(b) for b in a.blocks.split(",")]:
                def run_plain():
                    _lib.check(lib.loc_l1_forward_gemm_i8(X.data_ptr(), X.stride(0), rows.data_ptr(), n, C.byref(d),
                                                          image.data_ptr(), digits, 2, P + 4 * lay.b1,
                                                          partial.data_ptr(), partial.numel(), a1.data_ptr(), blocks,
                                                          C.byref(_lib.Tuning(gemm_i8_unit_tiles=a.unit_tiles)), None))
                run = run_plain
                if a.packed:
                    X2 = torch.zeros((X.shape[0], d.Kp // 4), dtype=torch.uint8, device=dev)
                    _lib.check(lib.loc_pack_genotypes_2bit(X.data_ptr(), X.stride(0), X.shape[0], d.Kp, X2.data_ptr(),
                                                           X2.stride(0), None))

                    def run():
                        _lib.check(lib.loc_l1_forward_gemm_i8_packed(X2.data_ptr(), X2.stride(0), rows.data_ptr(), n,
                                                                     C.byref(d), image.data_ptr(), digits, P + 4 * lay.b1,
                                                                     partial.data_ptr(), partial.numel(), a1.data_ptr(),
                                                                     blocks, C.byref(_lib.Tuning(gemm_i8_unit_tiles=a.unit_tiles)),
                                                                     None))
                    prep(); run_plain(); torch.cuda.synchronize()
                    ref = a1[:n].clone()
                    run(); torch.cuda.synchronize()
                    assert torch.equal(ref, a1[:n]), f"packed result differs: max {float((ref - a1[:n]).abs().max())}"
                t = {}
                for name, fn in (("prep", prep), ("gemm", run)):
                    for _ in range(5):
                        fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(a.iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    t[name] = e0.elapsed_time(e1) * 1e3 / a.iters
                flops = 2.0 * n * a.snps * a.width
                rec = {"kernel": "int8 image+gemm" + (", 2-bit packed genotypes" if a.packed else ""), "blocks": blocks, "rows": n, "snps": a.snps, "width": a.width,
                       "digits": digits, "unit_tiles": a.unit_tiles, "us_gemm": round(t["gemm"], 2), "us_prep": round(t["prep"], 2),
                       "tflops": round(flops / t["gemm"] * 1e-6, 1),
                       "frac_bf16_peak": round(flops / t["gemm"] * 1e-6 / BF16_PEAK_TFLOPS, 4),
                       "frac_bf16_peak_incl_prep": round(flops / (t["gemm"] + t["prep"]) * 1e-6 / BF16_PEAK_TFLOPS, 4),
                       "i8_mfma_issue_frac": round(digits * 0.5 * flops / t["gemm"] * 1e-6 / BF16_PEAK_TFLOPS, 4)}
                print(json.dumps(rec), flush=True)
                out.append(rec)
    # image-based GEMM (l1_gemm.hip): conversion once per sweep, then a pure matrix-pipe K loop
    for n in ([] if a.i8_only else [int(r) for r in a.rows.split(",")]):
        rows = (torch.arange(n, dtype=torch.int32, device=dev) % n_max).contiguous()
        a1 = torch.empty(((n + 127) // 128 * 128, d.Hp), device=dev)
        for pieces in (3, 2, 1):
            if not lib.loc_l1_gemm_supported(d.Hp, pieces):
                continue
            image = torch.empty(lib.loc_l1_image_bytes(C.byref(d), pieces), dtype=torch.uint8, device=dev)

            def prep():
                _lib.check(lib.loc_l1_image_build(C.byref(d), bn4.data_ptr(), P + 4 * lay.w1, pieces,
                                                  image.data_ptr(), None))
            for blocks in [int(b) for b in a.blocks.split(",")]:
                def run():
                    _lib.check(lib.loc_l1_forward_gemm(X.data_ptr(), X.stride(0), rows.data_ptr(), n, C.byref(d),
                                                       image.data_ptr(), pieces, P + 4 * lay.b1, partial.data_ptr(),
                                                       partial.numel(), a1.data_ptr(), blocks, None))
                t = {}
                for name, fn in (("prep", prep), ("gemm", run)):
                    for _ in range(5):
                        fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(a.iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    t[name] = e0.elapsed_time(e1) * 1e3 / a.iters
                flops = 2.0 * n * a.snps * a.width
                byts = n * a.snps + 2.0 * pieces * a.snps * a.width
                rec = {"kernel": "image+gemm", "blocks": blocks, "rows": n, "snps": a.snps, "width": a.width,
                       "pieces": pieces, "us_gemm": round(t["gemm"], 2), "us_prep": round(t["prep"], 2),
                       "tflops": round(flops / t["gemm"] * 1e-6, 1),
                       "frac_bf16_peak": round(flops / t["gemm"] * 1e-6 / BF16_PEAK_TFLOPS, 4),
                       "frac_bf16_peak_incl_prep": round(flops / (t["gemm"] + t["prep"]) * 1e-6 / BF16_PEAK_TFLOPS, 4),
                       "mfma_issue_frac": round(pieces * flops / t["gemm"] * 1e-6 / BF16_PEAK_TFLOPS, 4),
                       "gbs": round(byts / t["gemm"] * 1e-3, 1)}
                print(json.dumps(rec), flush=True)
                out.append(rec)
    return out


if __name__ == "__main__":
    main()
