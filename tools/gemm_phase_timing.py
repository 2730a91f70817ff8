"""In-kernel s_memtime stamps of the two-group GEMM schedule (variant 192), workgroup 0, one lane per wave."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locator_amd import _lib  # noqa: E402
from locator_amd.net import LocatorNet  # noqa: E402

pieces = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 192
K, n = 100000, 1000
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
X = (torch.rand((n, K), generator=g) < 0.3).to(torch.uint8).to(dev)
net = LocatorNet(X, torch.zeros((n, 2), device=dev), K, 256, 10, 0.25, seed=1)
lib, d, lay = net.lib, net.d, net.lay
P = net.params.data_ptr()
bn4 = torch.zeros(4 * d.Kp, device=dev)
_lib.check(lib.loc_bn_infer_scale_shift(d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta, P + 4 * lay.mov_mean,
                                        P + 4 * lay.mov_var, bn4.data_ptr(), None))
partial = torch.empty(256 * 128 * d.Hp, device=dev)
rows = torch.arange(n, dtype=torch.int32, device=dev)
a1 = torch.empty((1024, d.Hp), device=dev)
image = torch.empty(lib.loc_l1_image_bytes(C.byref(d), pieces), dtype=torch.uint8, device=dev)
_lib.check(lib.loc_l1_image_build(C.byref(d), bn4.data_ptr(), P + 4 * lay.w1, pieces, image.data_ptr(), None))
for _ in range(3):
    _lib.check(lib.loc_l1_forward_gemm(X.data_ptr(), X.stride(0), rows.data_ptr(), n, C.byref(d), image.data_ptr(),
                                       pieces, P + 4 * lay.b1, partial.data_ptr(), partial.numel(), a1.data_ptr(),
                                       variant << 16, None))
torch.cuda.synchronize()
buf = np.zeros(8 * 1024, np.uint64)
assert lib.loc_l1_gemm_debug_read(buf.ctypes.data) == 0
buf = buf.reshape(8, 1024).astype(np.int64)
if variant == 320:
    for w in (0, 1, 4, 5):
        s = buf[w]
        nz = int((s > 0).sum()) // 5 * 5
        dd = s[:nz].reshape(-1, 5)
        print(f"wave {w}: tiles {len(dd)}; mean ticks [read->kk0 ready, kk1, kk2, kk3, last MFMAs issued] over tiles 5..40:", np.round(dd[5:40].mean(0), 1), "sum", round(float(dd[5:40].sum(1).mean()), 1))
    sys.exit(0)
if variant in (640, 644):
    for w in (0, 1, 4, 5):
        s = buf[w]
        nz = int((s > 0).sum()) // 3 * 3
        d = np.diff(s[:nz])
        dd = d[:len(d) // 3 * 3].reshape(-1, 3)
        print(f"wave {w}: iterations {len(dd)}; mean ticks [first part, second part, barrier] over iterations 5..40:", np.round(dd[5:40].mean(0), 1), "sum", round(float(dd[5:40].sum(1).mean()), 1))
    sys.exit(0)
# per iteration stamps: group 0: [after wait_vm (p==0 only)], before barrier1, after barrier1, before barrier2, after barrier2
for w in (0, 4):
    s = buf[w]
    nz = int((s > 0).sum())
    s = s[:nz] - s[0]
    print(f"wave {w}: {nz} stamps, total {s[-1]} ticks")
    per = 5 if (w == 0 and pieces == 1) else 5 if w == 4 else None
    if per:
        body = s[1:1 + (nz - 1) // per * per].reshape(-1, per)
        dd = np.diff(np.concatenate([[s[0]], body.reshape(-1)])).reshape(-1, per)
        print("  mean ticks per segment over iterations 5..40:", np.round(dd[5:40].mean(0), 1), " iteration:", round(float(dd[5:40].sum(1).mean()), 1))
        print("  first 3 iterations:", dd[:3].tolist())
