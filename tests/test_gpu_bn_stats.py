"""Epoch-level BatchNorm batch statistics (loc_bn_epoch_stats_only; reference: BatchNormalization in training mode, batch mean
and BIASED variance per SNP, /root/reference/locator/locator.py:318): the 16-SNPs-per-thread kernel (16-byte aligned rows)
against the 4-SNPs-per-thread one (any 4-byte aligned pitch) bit for bit, and both against NumPy."""
import ctypes as C

import numpy as np
import pytest
import torch

from locator_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,n,batch", [(1000, 70, 32), (100000, 200, 32), (5830, 150, 64), (31, 40, 7)])
def test_epoch_batch_statistics_wide_and_narrow_loads_agree_bit_for_bit(K, n, batch):
    lib = _lib.load()
    rng = np.random.default_rng(K + n)
    Kp = (K + 31) // 32 * 32
    x = rng.integers(0, 3, (n, Kp), dtype=np.uint8)
    x[:, K:] = 0
    x[rng.integers(0, n, 5), rng.integers(0, K, 5)] = 255                 # (the integer sums must hold any byte)
    perm = rng.permutation(n).astype(np.int32)
    n_steps = (n + batch - 1) // batch
    n_last = n - (n_steps - 1) * batch
    rows = torch.from_numpy(perm).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    Xa = torch.from_numpy(x).cuda()                                       # pitch Kp: rows 16-byte aligned -> the wide kernel
    Xb_store = torch.zeros((n, Kp + 4), dtype=torch.uint8, device="cuda")
    Xb_store[:, :Kp] = Xa
    out = []
    for X, pitch in ((Xa, Kp), (Xb_store, Kp + 4)):                       # pitch Kp + 4: only 4-byte aligned -> the narrow kernel
        stats = torch.full((n_steps * 2 * Kp,), -7.0, device="cuda")
        _lib.check(lib.loc_bn_epoch_stats_only(C.c_void_p(X.data_ptr()), C.c_int64(pitch), C.c_void_p(rows.data_ptr()), batch,
                                               n_last, n_steps, K, Kp, C.c_void_p(stats.data_ptr()), st), "stats")
        torch.cuda.synchronize()
        out.append(stats.cpu().numpy().reshape(n_steps, 2, Kp))
    assert np.array_equal(out[0], out[1])
    for j in range(n_steps):
        xb = x[perm[j * batch:(j + 1) * batch]].astype(np.float64)
        np.testing.assert_allclose(out[0][j, 0, :K], xb.mean(0)[:K], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(out[0][j, 1, :K], xb.var(0)[:K], rtol=1e-5, atol=1e-6)
        assert not out[0][j, :, K:].any()
