#!/usr/bin/env python3
"""The full large-M layer-1 GEMM sweep (every predict mode x every shape), moved out of bench.py's headline line in round 5.

    python tools/l1_gemm_sweep.py [--out profiles/rNN_l1_gemm_sweep.json] [--snps 100000] [--n 1000]
    python bench.py --l1-gemm-full profiles/rNN_l1_gemm_sweep.json        (same sweep on the bench's own fitted net)

One record per shape x mode: the large-M first-layer genotype GEMM (model.predict over all 1000 rows; 4096 / 16384 rows of
the same matrix = the batched --jacknife; 4096 / 16384 DISTINCT rows streaming from HBM) as a fraction of the dense
bf16-MFMA peak, for every predict mode the CLI ships - int8 x 3 digit planes (exact), int8 x 2 (the default while the
dynamic-range guard allows it), bf16 x 3 / 2 / 1 pieces - plus the in-loop-conversion kernel used for few rows.  The
tolerance each mode's PREDICTIONS are tested to is written once, in PREDICT_MODE_INFO below and DESIGN.md §5.
bench.py's line carries only the dozen-number summary of the default mode (bench.py: l1_gemm_summary).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BF16_PEAK_TFLOPS = 2500.0    # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)


REPLAYS = 9     # timed replays per figure: every quoted time is the MEDIAN of this many (one replay of a kernel
#                 whose clock sags under matrix load is a sample, not a measurement)


def _capture(fn, iters):
    """A HIP graph of `iters` back-to-back fn() (which enqueues on the current stream) and the stream it replays on."""
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()                           # warm replay, untimed
        s.synchronize()
    return g, s


def _replay_us(g, s, iters):
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record(s)
        g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def _stats(samples):
    import statistics
    return {"median": statistics.median(samples), "min": min(samples), "max": max(samples), "replays": len(samples)}


def _time_graphed(fn, iters, replays=REPLAYS):
    """MEDIAN microseconds of fn() over `replays` timed replays of a captured HIP graph of `iters` back-to-back launches (a graph:
    host launch overhead does not pad kernel time; the sustained, clocked-down rate).  `_time_graphed.last` keeps
    {median, min, max, replays} of the call."""
    import torch
    g, s = _capture(fn, iters)
    st = _stats([_replay_us(g, s, iters) for _ in range(max(1, replays))])
    torch.cuda.current_stream().wait_stream(s)
    _time_graphed.last = st
    return st["median"]


def _time_graphed_ab(fns, iters, rounds=REPLAYS):
    """Interleaved A/B(/C...) timing of several variants in ONE process: a graph per variant, `rounds` rounds of one timed replay
    each in turn (clock state, box and time are shared, so the differences are the variants').  Returns one {median, min, max,
    replays} per variant."""
    import torch
    caps = [_capture(fn, iters) for fn in fns]
    samples = [[] for _ in fns]
    for _ in range(max(1, rounds)):
        for i, (g, s) in enumerate(caps):
            samples[i].append(_replay_us(g, s, iters))
    for _, s in caps:
        torch.cuda.current_stream().wait_stream(s)
    return [_stats(v) for v in samples]


def _time_burst(fn, n=3, idle_s=0.25):
    """Mean microseconds of fn() over a short burst of n launches after the GPU sat idle: the matrix-pipe kernels clock
    down within a few ms of back-to-back load (rocprofv3 trace: launch 1-12 of l1_gemm_kernel<3> 122 us, launch 20
    154 us), and a predict issues ONE such launch per 4096 rows, not twenty."""
    import torch
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        time.sleep(idle_s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        best = us if best is None else min(best, us)
    return best


# Predict modes the CLI ships (locator_amd/locator.py --predict_mode / --predict_pieces) and the bar each one is held to
# on the PREDICTIONS at this workload's K (tests/test_gpu_baseline_sizes.py::test_config2_predict_all_rows):
PREDICT_MODE_INFO = {
    "int8x3": "--predict_mode exact (and what the default takes when the dynamic-range guard refuses two planes): 24-bit fixed "
              "point per weight, predictions 2e-5 absolute; tests/test_gpu_trained_predict.py: 8e-7 on converged fits",
    "int8x2": "DEFAULT (--predict_mode auto) while the guard allows it (largest / typical scaled weight per unit: median <= 64, "
              "worst <= 512; converged metric fit: 28 / 77): 16-bit fixed point per weight, predictions <= 1e-3 relative "
              "(north_star bound), measured 5e-5 on the converged metric fit, 7e-5 on the reference's example fit "
              "(tests/test_gpu_trained_predict.py); also --predict_mode fast (unconditional)",
    "bf16x3": "--predict_pieces 3 (and the fallback for genotypes > 127): fp32-exact products, 2e-5 absolute",
    "bf16x2": "--predict_pieces 2: predictions <= 1e-3 relative",
    "bf16x1": "--predict_pieces 1: plain bf16 weights, predictions only within 2e-2 - OUTSIDE the north_star tolerance, "
              "listed for reference, not a figure against the MFMA target",
}


def l1_gemm_roofline(net, n_matrix, iters=20, x_distinct=None):
    """The only large-M contraction on the path (model.predict / --jacknife, locator.py:414, :441, :683-747):
    a1 = ELU(BN(x) W1 + b1) for M rows at once.  flops = 2*M*K*H counted ONCE, however many int8 digits or bf16 pieces
    carry each fp32 weight; the denominator is the dense bf16-MFMA peak for every mode (int8 digits run on the i8 pipe
    at twice the bf16 rate: 3 digits cost 1.5 bf16-MFMA equivalents per product, 2 digits 1).  Shapes: M = every row of
    the matrix (model.predict over all samples); M = 4096 and M = 16384 (= LOC_PREDICT_CHUNK, what loc_predict launches at
    a time) rows drawn from it (the batched --jacknife: nboots x n_pred perturbed rows of the same matrix in one predict, so
    genotype lines repeat and come from L2 / MALL); M = 4096 and M = 16384
    DISTINCT rows of a second synthetic matrix (every genotype byte streams from HBM once).  Per shape and mode: us =
    GEMM + its reduction, mean of `iters` back-to-back launches replayed from a graph (the sustained, clocked-down
    rate); us_prep = the once-per-predict weight conversion (not in us; frac_bf16_peak_incl_prep has it); burst_of_3 =
    three launches from idle; `in_loop_conversion` = loc_l1_forward_rows, which converts inside the K loop (few rows)."""
    import ctypes as C

    import torch
    from locator_amd import _lib
    lib, d, lay = net.lib, net.d, net.lay
    P = net.params.data_ptr()
    dev = net.params.device
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bn4 = torch.zeros(4 * d.Kp, device=dev)
    _lib.check(lib.loc_bn_infer_scale_shift(d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta, P + 4 * lay.mov_mean,
                                            P + 4 * lay.mov_var, bn4.data_ptr(), st()))
    partial = torch.empty(256 * 128 * d.Hp, device=dev)

    def shape(X, n_rows, n_src, in_loop=True):
        rows = (torch.arange(n_rows, dtype=torch.int32, device=dev) % n_src).contiguous()
        a1 = torch.empty(((n_rows + 127) // 128 * 128, d.Hp), device=dev)
        flops = 2.0 * n_rows * d.K * d.H
        out = {"rows": n_rows, "distinct_rows": min(n_rows, n_src), "flops": flops}

        def rec(us, mfma_equiv, byts):
            tf = flops / us * 1e-6
            return {"us": round(us, 1), "tflops": round(tf, 1), "frac_bf16_peak": round(tf / BF16_PEAK_TFLOPS, 4),
                    "mfma_issue_frac": round(mfma_equiv * tf / BF16_PEAK_TFLOPS, 4), "hbm_gbs": round(byts / us * 1e-3, 1)}

        def timed(prep, run, mfma_equiv, byts):
            prep()
            us_prep = _time_graphed(prep, 10)
            us = _time_graphed(run, iters)
            r = rec(us, mfma_equiv, byts)
            r["us_min_max"] = [round(_time_graphed.last["min"], 1), round(_time_graphed.last["max"], 1)]
            r["us_prep"] = round(us_prep, 1)
            us_b = _time_burst(run)
            r["burst_of_3"] = {"us": round(us_b, 1), "frac_bf16_peak": round(flops / us_b * 1e-6 / BF16_PEAK_TFLOPS, 4)}
            r["frac_bf16_peak_incl_prep"] = round(flops / (us + us_prep) * 1e-6 / BF16_PEAK_TFLOPS, 4)
            return r

        def product_path(Xm, pitch, packed, image, digits, us_with_reduce):
            """The alternative built and measured in round 4 (loc_tuning.gemm_reduce = 1; NOT the default, it is slower): the GEMM
            kernel without its reduction launch (loc_l1_forward_gemm_i8_partial) and the hidden-stack launch adding the group
            partial sums up in its input stage.  us_gemm_kernel = the GEMM alone; us_stack / us_stack_fused = the hidden-stack
            launch (what follows the GEMM in a predict) fed from a1 / from the partial sums; us_layer1 = us_gemm_kernel + what
            the fusion adds to the stack launch, to be read against us_with_reduce_launch (the default)."""
            groups, cv = C.c_int(0), C.c_void_p()
            yh = torch.empty((n_rows, 2), device=dev)
            runk = lambda: _lib.check(lib.loc_l1_forward_gemm_i8_partial(Xm.data_ptr(), pitch, packed, rows.data_ptr(), n_rows,
                                                                         C.byref(d), image.data_ptr(), digits, 2,
                                                                         partial.data_ptr(), partial.numel(), 0, None,
                                                                         C.byref(groups), C.byref(cv), st()))
            runk()
            mp = (n_rows + 127) // 128 * 128
            args_tail = (P + 4 * lay.wh, P + 4 * lay.bh, P + 4 * lay.wa, P + 4 * lay.ba, P + 4 * lay.wb, P + 4 * lay.bb, d.Hp, d.L,
                         n_rows, None, None, yh.data_ptr(), None, st)
            run_s = lambda: _lib.check(lib.loc_stack_forward_eval(a1.data_ptr(), *args_tail[:-1], st()))
            run_f = lambda: _lib.check(lib.loc_stack_forward_eval_partial(partial.data_ptr(), groups.value, mp * d.Hp, cv,
                                                                         P + 4 * lay.b1, *args_tail[:-1], 0, st()))
            us_k = _time_graphed(runk, iters)
            it_s = max(3, iters // 4)
            us_s, us_f = _time_graphed(run_s, it_s), _time_graphed(run_f, it_s)
            us_l1 = us_k + max(0.0, us_f - us_s)
            return {"us_gemm_kernel": round(us_k, 1), "groups": groups.value, "us_stack": round(us_s, 1),
                    "us_stack_fused": round(us_f, 1), "us_layer1": round(us_l1, 1),
                    "frac_bf16_peak": round(flops / us_l1 * 1e-6 / BF16_PEAK_TFLOPS, 4),
                    "us_with_reduce_launch": us_with_reduce}

        for digits in (3, 2):
            if lib.loc_l1_gemm_i8_supported(d.Hp, digits):
                image = torch.empty(lib.loc_l1_image_i8_bytes(C.byref(d), digits), dtype=torch.uint8, device=dev)
                prep = lambda: _lib.check(lib.loc_l1_image_i8_build(C.byref(d), bn4.data_ptr(), P + 4 * lay.w1, digits,
                                                                    image.data_ptr(), st()))
                run = lambda: _lib.check(lib.loc_l1_forward_gemm_i8(X.data_ptr(), X.stride(0), rows.data_ptr(), n_rows,
                                                                    C.byref(d), image.data_ptr(), digits, 2,
                                                                    P + 4 * lay.b1, partial.data_ptr(), partial.numel(),
                                                                    a1.data_ptr(), 0, None, st()))
                key = "int8x%d" % digits
                out[key] = timed(prep, run, 0.5 * digits, n_rows * d.K + 1.0 * digits * d.K * d.H)
                out[key]["reduce_fused_into_stack"] = product_path(X, X.stride(0), 0, image, digits, out[key]["us"])
                # the same GEMM reading a 2-bit packed copy of the matrix (--predict_packed / loc_net.X2; not the default:
                # packing costs one pass over the matrix, see us_pack): bit-identical activations
                X2 = torch.zeros((X.shape[0], d.Kp // 4), dtype=torch.uint8, device=dev)
                pack = lambda: _lib.check(lib.loc_pack_genotypes_2bit(X.data_ptr(), X.stride(0), X.shape[0], d.Kp,
                                                                      X2.data_ptr(), X2.stride(0), st()))
                pack()
                runp = lambda: _lib.check(lib.loc_l1_forward_gemm_i8_packed(X2.data_ptr(), X2.stride(0), rows.data_ptr(),
                                                                            n_rows, C.byref(d), image.data_ptr(), digits,
                                                                            P + 4 * lay.b1, partial.data_ptr(),
                                                                            partial.numel(), a1.data_ptr(), 0, None, st()))
                keyp = key + "_packed2bit"
                out[keyp] = timed(prep, runp, 0.5 * digits, n_rows * d.K / 4 + 1.0 * digits * d.K * d.H)
                out[keyp]["reduce_fused_into_stack"] = product_path(X2, X2.stride(0), 1, image, digits, out[keyp]["us"])
                out[keyp]["us_pack"] = round(_time_graphed(pack, 5), 1)
                del image, X2
        for pieces in (3, 2, 1):
            key = "bf16x%d" % pieces
            if lib.loc_l1_gemm_supported(d.Hp, pieces):
                image = torch.empty(lib.loc_l1_image_bytes(C.byref(d), pieces), dtype=torch.uint8, device=dev)
                prep = lambda: _lib.check(lib.loc_l1_image_build(C.byref(d), bn4.data_ptr(), P + 4 * lay.w1, pieces,
                                                                 image.data_ptr(), st()))
                run = lambda: _lib.check(lib.loc_l1_forward_gemm(X.data_ptr(), X.stride(0), rows.data_ptr(),
                                                                 n_rows, C.byref(d), image.data_ptr(), pieces,
                                                                 P + 4 * lay.b1, partial.data_ptr(), partial.numel(),
                                                                 a1.data_ptr(), 0, st()))
                out[key] = timed(prep, run, pieces, n_rows * d.K + 2.0 * pieces * d.K * d.H)
                del image
            if in_loop and pieces != 2 and lib.loc_l1_rows_supported(d.Hp, pieces):
                run = lambda: _lib.check(lib.loc_l1_forward_rows(X.data_ptr(), X.stride(0), rows.data_ptr(),
                                                                 n_rows, C.byref(d), bn4.data_ptr(), P + 4 * lay.w1,
                                                                 P + 4 * lay.b1, partial.data_ptr(), partial.numel(),
                                                                 a1.data_ptr(), pieces, 0, None, st()))
                out.setdefault("in_loop_conversion", {})[key] = rec(_time_graphed(run, iters), pieces,
                                                                    n_rows * d.K + 4.0 * d.K * d.H)
        return out

    res = shape(net.X, n_matrix, n_matrix)
    res["peak_tflops"] = BF16_PEAK_TFLOPS
    g = net.quant_guard()
    res["default_mode"] = {"flag": "--predict_mode auto (the CLI default)",
                           "guard": {"median_range": round(g[0], 1), "max_range": round(g[1], 1)},
                           "digit_planes": int(g[2]),
                           "takes": ("int8x%d" % int(g[2]) if g[2] > 0 else "bf16x3") + ", from the 2-bit packed matrix for chunks of "
                                    ">= 3072 rows (packed automatically when the genotypes are <= 3); the hidden stack that follows takes the "
                                    "fp32 matrix pipe from 3072 rows per chunk (us_stack)"}
    res["kernel"] = ("int8: l1_gemm_i8_kernel + l1_gemm_reduce_kernel (digit planes written once per predict by l1_scan_kernel "
                     "+ l1_image_i8_kernel); bf16: l1_gemm_kernel + l1_gemm_reduce_kernel (l1_image_kernel)")
    res["jacknife_shape_4096_rows"] = shape(net.X, 4096, n_matrix, in_loop=False)
    res["jacknife_shape_16384_rows"] = shape(net.X, 16384, n_matrix, in_loop=False)     # one LOC_PREDICT_CHUNK
    if x_distinct is not None:
        res["distinct_4096_rows"] = shape(x_distinct[:4096], 4096, 4096, in_loop=False)
        if x_distinct.shape[0] >= 16384:
            res["distinct_16384_rows"] = shape(x_distinct, 16384, 16384, in_loop=False)
    return res


def distinct_rows(dev, Kp, n_rows=16384):
    """n_rows DISTINCT synthetic genotype rows (0/1/2 drawn on the device, 4096 at a time) for the streaming shapes."""
    import torch
    g = torch.Generator(device=dev).manual_seed(4096)
    xd = torch.empty((n_rows, Kp), dtype=torch.uint8, device=dev)
    for r0 in range(0, n_rows, 4096):
        u = torch.rand((min(4096, n_rows - r0), Kp), device=dev, generator=g)
        xd[r0:r0 + u.shape[0]] = (u < 0.25).to(torch.uint8) + (u < 0.08).to(torch.uint8)
        del u
    return xd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None, help="write the sweep here (JSON) instead of stdout")
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--snps", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    import numpy as np
    import torch
    from locator_amd.net import LocatorNet, upload_genotypes
    from locator_amd.synth import normalize_locs, synth_genotypes
    dev = "cuda:0"
    x, locs = synth_genotypes(a.n, a.snps, seed=20260101, n_na=a.n // 10)
    _, _, _, _, ynorm = normalize_locs(locs)
    X = upload_genotypes(x, dev)
    Y = torch.from_numpy(np.nan_to_num(ynorm).astype(np.float32)).to(dev)
    net = LocatorNet(X, Y, a.snps, a.width, 10, 0.25, seed=12345, device=dev)
    res = l1_gemm_roofline(net, a.n, iters=a.iters, x_distinct=distinct_rows(dev, net.d.Kp))
    res["tolerances"] = PREDICT_MODE_INFO
    text = json.dumps(res, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    else:
        print(text)


if __name__ == "__main__":
    main()
