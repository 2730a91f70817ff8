"""Experiment driver for l1_gemm.hip: runs loc_l1_forward_gemm for each (pieces, variant) so that
`rocprofv3 --kernel-trace --stats` gives one line per kernel instantiation.

    rocprofv3 --kernel-trace --stats -d out -o k --output-format csv -- python3 tools/gemm_variants.py --variants 0,1
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locator_amd import _lib  # noqa: E402
from locator_amd.net import LocatorNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--snps", type=int, default=100000)
ap.add_argument("--rows", type=int, default=1000)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--pieces", default="1,3")
ap.add_argument("--variants", default="0")
ap.add_argument("--blocks", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
X = (torch.rand((a.rows, (a.snps + 31) // 32 * 32), generator=g) < 0.3).to(torch.uint8)
X = (X + (torch.rand(X.shape, generator=g) < 0.3).to(torch.uint8)).to(dev)
Y = torch.zeros((a.rows, 2), device=dev)
net = LocatorNet(X, Y, a.snps, 256, 10, 0.25, seed=1)
lib, d, lay = net.lib, net.d, net.lay
P = net.params.data_ptr()
bn4 = torch.zeros(4 * d.Kp, device=dev)
_lib.check(lib.loc_bn_infer_scale_shift(d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta, P + 4 * lay.mov_mean,
                                        P + 4 * lay.mov_var, bn4.data_ptr(), None))
partial = torch.empty(256 * 128 * d.Hp, device=dev)
rows = torch.arange(a.rows, dtype=torch.int32, device=dev)
a1 = torch.empty(((a.rows + 127) // 128 * 128, d.Hp), device=dev)
for pieces in [int(p) for p in a.pieces.split(",")]:
    image = torch.empty(lib.loc_l1_image_bytes(C.byref(d), pieces), dtype=torch.uint8, device=dev)
    for _ in range(3):
        _lib.check(lib.loc_l1_image_build(C.byref(d), bn4.data_ptr(), P + 4 * lay.w1, pieces, image.data_ptr(), None))
    ref = None
    for v in [int(x) for x in a.variants.split(",")]:
        a1.fill_(float("nan"))
        _lib.check(lib.loc_l1_forward_gemm(X.data_ptr(), X.stride(0), rows.data_ptr(), a.rows, C.byref(d),
                                           image.data_ptr(), pieces, P + 4 * lay.b1, partial.data_ptr(),
                                           partial.numel(), a1.data_ptr(), a.blocks | (v << 16), None))
        torch.cuda.synchronize()
        if ref is None:
            ref = a1.clone()
        print(f"pieces {pieces} variant {v}: max |a1 - a1(first variant)| = {(a1 - ref).abs().max().item():.3e}", flush=True)
        for _ in range(a.iters):
            _lib.check(lib.loc_l1_forward_gemm(X.data_ptr(), X.stride(0), rows.data_ptr(), a.rows, C.byref(d),
                                               image.data_ptr(), pieces, P + 4 * lay.b1, partial.data_ptr(),
                                               partial.numel(), a1.data_ptr(), a.blocks | (v << 16), None))
        torch.cuda.synchronize()
