#!/usr/bin/env python3
"""Per-phase timestamps from inside stack_fused_kernel (workgroup 0, thread 0; wall_clock64 = 100 MHz)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from locator_amd import _lib
from locator_amd.net import LocatorNet, upload_genotypes

K, H = 100000, 256
rng = np.random.default_rng(0)
x = rng.integers(0, 3, (64, K)).astype(np.uint8)
y = rng.normal(0, 1, (64, 2)).astype(np.float32)
net = LocatorNet(upload_genotypes(x), torch.from_numpy(y).cuda(), K, H, 10, 0.25, seed=1)
lib = _lib.load()
dbg = torch.zeros(256, dtype=torch.int64, device="cuda")
rows = torch.arange(32, dtype=torch.int32, device="cuda")
mask = torch.ones(32 * 256, dtype=torch.uint8, device="cuda")
loss = torch.zeros(1, device="cuda")
for it in range(5):
    if it == 4:
        lib.loc_debug_set_buffer(dbg.data_ptr())
    net.train_step(rows, 32, it + 1, mask, loss)
torch.cuda.synchronize()
lib.loc_debug_set_buffer(None)
d = dbg.cpu().numpy()
d = d[d > 0]
dt = np.diff(d) * 10.0   # ns (100 MHz)
names = ["contract", "barrier", "epilogue", "barrier+next-prologue"]
print("stamps:", len(d))
for i in range(0, min(len(dt), 36), 4):
    print(f"pass {i//4}: " + "  ".join(f"{names[j]}={dt[i+j]:7.0f}ns" for j in range(4) if i + j < len(dt)))
print("total fwd (ns):", (d[-1] - d[0]) * 10)
