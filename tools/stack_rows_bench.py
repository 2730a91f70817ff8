#!/usr/bin/env python3
"""Times the hidden stack of a many-row predict in its three forms (loc_stack_forward_eval_form: -1 = 2 rows per workgroup on
the vector ALU, 1 = 32-row tiles on v_mfma_f32_32x32x2_f32, 2 = 16-row tiles on v_mfma_f32_16x16x4_f32) over row counts, from
a captured graph (mean of 20 back-to-back launches).   python tools/stack_rows_bench.py [--rows 512,1024,...]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from locator_amd import _lib  # noqa: E402
from locator_amd.net import LocatorNet  # noqa: E402
from tools.l1_gemm_sweep import _time_graphed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="512,1024,1536,2048,3072,4096,6144,8191,8192,12288,16384")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    K = 2048
    X = torch.zeros((64, K), dtype=torch.uint8, device=dev)
    Y = torch.zeros((64, 2), device=dev)
    net = LocatorNet(X, Y, K, 256, 10, 0.25, seed=1)
    lib, d, lay, P = net.lib, net.d, net.lay, net.params.data_ptr()
    for n in [int(r) for r in a.rows.split(",")]:
        a1 = torch.randn(((n + 127) // 128 * 128, d.Hp), device=dev) * 0.5
        yhat = torch.zeros((n, 2), device=dev)
        rec = {"rows": n}
        for form, name in ((-1, "valu_2rows"), (-2, "valu_4rows"), (-3, "valu_8rows"), (1, "mfma_32rows"), (2, "mfma_16rows"), (0, "default")):
            run = lambda: _lib.check(lib.loc_stack_forward_eval_form(a1.data_ptr(), P + 4 * lay.wh, P + 4 * lay.bh, P + 4 * lay.wa,
                                                                     P + 4 * lay.ba, P + 4 * lay.wb, P + 4 * lay.bb, d.Hp, d.L, n,
                                                                     None, None, yhat.data_ptr(), None, form,
                                                                     torch.cuda.current_stream().cuda_stream))
            rec[name + "_us"] = round(_time_graphed(run, a.iters), 1)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
