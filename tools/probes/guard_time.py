"""Kernel times of the dynamic-range scan (l1_scan_kernel + l1_quant_guard_kernel; round 4: l1_colmax_kernel) in isolation: run under
    rocprofv3 --kernel-trace --stats -- python3 tools/probes/guard_time.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from locator_amd import _lib
from locator_amd.net import LocatorNet, _stream

K = 100000
Kp = (K + 31) // 32 * 32
X = torch.zeros((64, Kp), dtype=torch.uint8, device="cuda")
Y = torch.zeros((64, 2), device="cuda")
net = LocatorNet(X, Y, K, 256, 10, 0.25, seed=3)
net.quant_guard()
c = net.cnet()
for i in range(50):
    _lib.check(net.lib.loc_predict_scan(C.byref(c), _stream()), "scan")
    torch.cuda.synchronize()
print("done")
