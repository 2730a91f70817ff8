"""Phase timing inside l1_bwd_adam_chain_kernel (measurement build: -DLOC_CHAIN_STAMPS=<workgroup> on the source with
tools/probes/l1_chain_probes.patch applied).
    bash tools/probes/build_chain_probe.sh chstamps -DLOC_CHAIN_STAMPS=100
    python3 tools/probes/chain_stamps.py build/liblocator_hip_chstamps.so
Prints, per k-tile iteration of one workgroup (min..max over its 8 waves, shader cycles): issue of the 12 prefetch loads, the
hand-counted wait, MFMA chain + reductions, Adam + 12 stores, transposes + staging, barrier, gamma / beta Adam + next forward."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from locator_amd import _lib

path = os.path.abspath(sys.argv[1])
_lib.use_library(path)
import torch

from locator_amd.net import LocatorNet, upload_genotypes
from locator_amd.synth import normalize_locs, split_indices, synth_genotypes
from locator_amd.train import FitLoop

K = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
n = 1000
x, locs = synth_genotypes(n, K, seed=20260101, n_na=n // 10)
ynorm = normalize_locs(locs)[4]
train, test, pred = split_indices(locs, seed=12345)
X = upload_genotypes(x, "cuda:0")
Y = torch.from_numpy(np.nan_to_num(np.asarray(ynorm)).astype(np.float32)).to("cuda:0")
net = LocatorNet(X, Y, K, 256, 10, 0.25, seed=12345, device="cuda:0")
rng = np.random.default_rng(99)
loop = FitLoop(net, train, test, batch_size=32, max_epochs=200, patience=10 ** 6, lr_patience=16, use_graph=True,
               perm_fn=lambda e: rng.permutation(len(train)), depth=2, xchain=True)
for _ in range(8):
    loop.submit(None)
    loop.collect(None)
loop.collect(0)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 16 * 8))()
lib = C.CDLL(path)
assert lib.loc_debug_chain_stamps(buf) == 0
s = np.array(buf[:], dtype=np.uint64).reshape(8, 16, 8).astype(np.int64)
names = ["issue 12 loads", "wait vmcnt(12)", "MFMA chain + dgamma", "Adam + 12 stores", "transpose + stage + fetch", "barrier",
         "gamma/beta Adam + forward"]
its = int((s[0, :, 0] > 0).sum())
print("iterations stamped:", its, " first stamp of each wave relative to the earliest:", (s[:, 0, 0] - s[:, 0, 0].min()).tolist())
print("it   total |", " | ".join(names))
tot = np.zeros(7)
for it in range(its):
    d = s[:, it, 1:] - s[:, it, :-1]
    nxt = (s[:, it + 1, 0] - s[:, it, 0]) if it + 1 < its else (s[:, it, 7] - s[:, it, 0])
    print(f"{it:2d} {int(np.median(nxt)):7d} |", " | ".join(f"{int(d[:, j].min())}..{int(d[:, j].max())}" for j in range(7)))
    tot += np.median(d, axis=0)
print("median cycles per phase over the launch:", [int(v) for v in tot], " sum", int(tot.sum()))
print("whole loop per wave:", (s[:, its - 1, 7] - s[:, 0, 0]).tolist())
