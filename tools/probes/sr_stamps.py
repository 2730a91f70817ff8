"""Phase timing inside stack_rows_eval_kernel (measurement build: -DSR_STAMPS=<block>, see stack_rows.hip).
    hipcc ... -DSR_STAMPS=300 -c stack_rows.hip  ->  build/liblocator_hip_srstamps.so
    python3 tools/probes/sr_stamps.py build/liblocator_hip_srstamps.so [rows]
Prints, per layer and wave, the cycles spent in the MFMA loop, the epilogue and the barrier."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from locator_amd import _lib

_lib.use_library(sys.argv[1])
import torch

from locator_amd.net import LocatorNet

n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K = 20000
Kp = (K + 31) // 32 * 32
g = torch.Generator(device="cuda").manual_seed(1)
X = (torch.rand((n, Kp), device="cuda", generator=g) < 0.2).to(torch.uint8)
X[:, K:] = 0
Y = torch.zeros((n, 2), device="cuda")
net = LocatorNet(X, Y, K, 256, 10, 0.25, seed=3, predict_digits=2)
rows = torch.arange(n, dtype=torch.int32, device="cuda")
yhat = torch.zeros((n, 2), device="cuda")
for _ in range(3):
    net.predict_rows(rows, n, yhat)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
lib = C.CDLL(sys.argv[1])
assert lib.loc_debug_sr_stamps(buf) == 0
s = np.array(buf[:], dtype=np.uint64).reshape(8, 64).astype(np.int64)
t0 = s[:, 0].min()
print("wave: first stamp relative to the earliest wave:", (s[:, 0] - t0).tolist())
print("layer  mfma-loop  epilogue  barrier  (cycles of the counter; min..max over the 8 waves)")
for l in range(9):
    a, b, c, d = s[:, 4 * l], s[:, 4 * l + 1], s[:, 4 * l + 2], s[:, 4 * l + 3]
    print(l + 2, (b - a).min(), (b - a).max(), "|", (c - b).min(), (c - b).max(), "|", (d - c).min(), (d - c).max())
print("whole stack per wave:", (s[:, 35] - s[:, 0]).tolist())
