"""GPU parity tests of the large-M layer-1 forward (loc_l1_forward_rows, bf16 matrix pipe) and of
loc_predict routed through it, against the fp64 oracle.

Tolerances on a1 = ELU(z1), |z1| = O(1):
  3 pieces : exact products, fp32 accumulation -> 2e-5 absolute (same bar as the fp32-MFMA kernel)
  2 pieces : weights carry 16 significand bits  -> 2e-4 absolute
  1 piece  : plain bf16 weights (2^-9 relative) -> 2e-2 absolute and <= 1e-2 relative to max|z1|
             ("bf16-MFMA layer 1: <= 1e-2 relative on z1")
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import locator_oracle as O
from tests.gpu_util import build_net, make_problem, maxerr

pytestmark = pytest.mark.gpu


def _a1_reference(p, x):
    """ELU(BN_inference(x) W1 + b1) in float64 (oracle/locator_oracle.py forward, inference mode)."""
    xh = (x.astype(np.float64) - p["mov_mean"]) / np.sqrt(p["mov_var"] + 1e-3) * p["gamma"] + p["beta"]
    z = xh @ p["W"][0] + p["b"][0]
    return np.where(z > 0, z, np.expm1(z)), z


def _run_rows(net, rows, n, pieces, target_blocks=0, scratch_tiles=512):
    from locator_amd import _lib
    d, lay = net.d, net.lay
    P = net.params.data_ptr()
    bn4 = torch.zeros(4 * d.Kp, device="cuda")
    _lib.check(net.lib.loc_bn_infer_scale_shift(d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta,
                                                P + 4 * lay.mov_mean, P + 4 * lay.mov_var, bn4.data_ptr(), None))
    mp = (n + 127) // 128 * 128
    partial = torch.empty(scratch_tiles * 128 * d.Hp, device="cuda")
    a1 = torch.full((mp, d.Hp), float("nan"), device="cuda")
    _lib.check(net.lib.loc_l1_forward_rows(net.X.data_ptr(), net.X.stride(0), rows.data_ptr(), n, C.byref(d),
                                           bn4.data_ptr(), P + 4 * lay.w1, P + 4 * lay.b1, partial.data_ptr(),
                                           partial.numel(), a1.data_ptr(), pieces, target_blocks, None, None),
               "loc_l1_forward_rows")
    torch.cuda.synchronize()
    return a1.cpu().numpy()


@pytest.mark.parametrize("K,width,n", [(200, 64, 33), (5830, 256, 450), (97, 100, 1), (1000, 32, 128),
                                       (4096, 256, 129), (3000, 320, 300), (640, 160, 127)])
def test_rows_forward_three_pieces_is_fp32_exact(K, width, n):
    x, y, p, rng = make_problem(max(n, 8), K, width, 2, seed=K + n)
    net = build_net(x, y, p)
    assert net.lib.loc_l1_rows_supported(net.d.Hp, 3)
    r = rng.permutation(x.shape[0])[:n].astype(np.int32)
    a1 = _run_rows(net, torch.from_numpy(r).cuda(), n, 3)
    ref, _ = _a1_reference(p, x[r])
    assert maxerr(a1[:n, :width], ref) < 2e-5, maxerr(a1[:n, :width], ref)
    assert np.isfinite(a1).all()                       # padded rows / units hold finite values


@pytest.mark.parametrize("pieces,tol_abs", [(2, 2e-4), (1, 2e-2)])
def test_rows_forward_fewer_pieces_within_stated_tolerance(pieces, tol_abs):
    K, width, n = 5830, 256, 200
    x, y, p, rng = make_problem(n, K, width, 2, seed=pieces)
    net = build_net(x, y, p)
    r = np.arange(n, dtype=np.int32)
    a1 = _run_rows(net, torch.from_numpy(r).cuda(), n, pieces)
    ref, z = _a1_reference(p, x[r])
    err = maxerr(a1[:n, :width], ref)
    assert err < tol_abs, err
    if pieces == 1:
        assert err <= 1e-2 * np.abs(z).max()
        # and it really is the approximate path: plain bf16 weights cannot reach the fp32 bar
        assert err > 2e-5


def test_rows_forward_is_deterministic_and_independent_of_the_split():
    """Same launch twice -> identical bits; a different SNP-group count only changes summation order."""
    K, width, n = 5830, 256, 300
    x, y, p, rng = make_problem(n, K, width, 2, seed=5)
    net = build_net(x, y, p)
    r = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
    a = _run_rows(net, r, n, 3)
    b = _run_rows(net, r, n, 3)
    assert np.array_equal(a, b)
    c = _run_rows(net, r, n, 3, target_blocks=24, scratch_tiles=24)
    assert maxerr(a[:n], c[:n]) < 5e-6


def test_rows_forward_rejects_what_it_cannot_do():
    from locator_amd import _lib
    x, y, p, rng = make_problem(40, 256, 512, 2, seed=9)
    net = build_net(x, y, p)
    assert not net.lib.loc_l1_rows_supported(512, 3)            # 3 x bf16 tiles of width 512 exceed the LDS
    assert net.lib.loc_l1_rows_supported(512, 1)
    assert not net.lib.loc_l1_rows_supported(256, 4) and not net.lib.loc_l1_rows_supported(256, 0)
    with pytest.raises(_lib.LocatorHipError, match="does not fit"):
        _run_rows(net, torch.arange(40, dtype=torch.int32, device="cuda"), 40, 3)
    x, y, p, rng = make_problem(200, 256, 64, 2, seed=9)
    net = build_net(x, y, p)
    with pytest.raises(_lib.LocatorHipError, match="scratch too small"):
        _run_rows(net, torch.arange(200, dtype=torch.int32, device="cuda"), 200, 3, scratch_tiles=1)


@pytest.mark.parametrize("K,width,nlayers,n", [(5830, 256, 10, 450), (2000, 128, 4, 1500), (300, 64, 3, 33)])
def test_predict_large_m_matches_oracle_and_the_32_row_kernels(K, width, nlayers, n):
    """loc_predict over more than 32 rows (bf16x3 layer 1, then one stack launch per 16,384-row chunk) vs
    oracle.predict (2e-5 abs), and vs the same rows pushed through the 32-row fp32-MFMA kernels
    (predict_pieces = -1): both are fp32-exact contractions, so they agree to summation-order noise."""
    x, y, p, rng = make_problem(n, K, width, nlayers, seed=n)
    net = build_net(x, y, p)
    rows = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
    yhat, dist = torch.zeros((n, 2), device="cuda"), torch.zeros(n, device="cuda")
    net.predict_rows(rows, n, yhat, dist)
    torch.cuda.synchronize()
    r = rows.cpu().numpy()
    ref = O.predict(p, x[r])
    assert maxerr(yhat.cpu().numpy(), ref) < 2e-5, maxerr(yhat.cpu().numpy(), ref)
    assert maxerr(dist.cpu().numpy(), O.euclid(ref, y[r])) < 2e-5
    net.predict_pieces = -1
    net._net = None
    yhat2 = torch.zeros((n, 2), device="cuda")
    net.predict_rows(rows, n, yhat2)
    torch.cuda.synchronize()
    assert maxerr(yhat.cpu().numpy(), yhat2.cpu().numpy()) < 1e-5
