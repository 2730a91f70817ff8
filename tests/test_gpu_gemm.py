"""GPU parity tests of the image-based large-M layer-1 GEMM (loc_l1_image_build + loc_l1_forward_gemm,
locator_amd/csrc/l1_gemm.hip) against the fp64 oracle forward (oracle/locator_oracle.py, inference mode)
and against the in-loop-conversion kernel it supersedes for many rows (loc_l1_forward_rows).

Reference lines: model.predict, /root/reference/locator/locator.py:414, :441; --jacknife, :683-747.
Tolerances on a1 = ELU(z1), |z1| = O(1) (same as tests/test_gpu_rows.py):
  3 pieces : exact products, fp32 accumulation -> 2e-5 absolute
  2 pieces : 16 significand bits per weight     -> 2e-4 absolute
  1 piece  : plain bf16 weights                 -> 2e-2 absolute and <= 1e-2 relative to max|z1|
"""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.gpu_util import build_net, make_problem, maxerr

pytestmark = pytest.mark.gpu


def _a1_reference(p, x):
    xh = (x.astype(np.float64) - p["mov_mean"]) / np.sqrt(p["mov_var"] + 1e-3) * p["gamma"] + p["beta"]
    z = xh @ p["W"][0] + p["b"][0]
    return np.where(z > 0, z, np.expm1(z)), z


def run_gemm(net, rows, n, pieces, target_blocks=0, scratch_tiles=256):
    from locator_amd import _lib
    d, lay, lib = net.d, net.lay, net.lib
    P = net.params.data_ptr()
    bn4 = torch.zeros(4 * d.Kp, device="cuda")
    _lib.check(lib.loc_bn_infer_scale_shift(d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta, P + 4 * lay.mov_mean,
                                            P + 4 * lay.mov_var, bn4.data_ptr(), None))
    image = torch.empty(lib.loc_l1_image_bytes(C.byref(d), pieces), dtype=torch.uint8, device="cuda")
    _lib.check(lib.loc_l1_image_build(C.byref(d), bn4.data_ptr(), P + 4 * lay.w1, pieces, image.data_ptr(), None),
               "loc_l1_image_build")
    mp = (n + 127) // 128 * 128
    partial = torch.empty(scratch_tiles * 128 * d.Hp, device="cuda")
    a1 = torch.full((mp, d.Hp), float("nan"), device="cuda")
    _lib.check(lib.loc_l1_forward_gemm(net.X.data_ptr(), net.X.stride(0), rows.data_ptr(), n, C.byref(d),
                                       image.data_ptr(), pieces, P + 4 * lay.b1, partial.data_ptr(), partial.numel(),
                                       a1.data_ptr(), target_blocks, None), "loc_l1_forward_gemm")
    torch.cuda.synchronize()
    return a1.cpu().numpy()


@pytest.mark.parametrize("K,n", [(5830, 450), (64, 1), (97, 130), (4096, 129), (3000, 300), (8192, 1000),
                                 (100, 128), (32, 5), (20000, 257)])
def test_gemm_three_pieces_is_fp32_exact(K, n):
    """K not a multiple of 64 (zero tail of the image), K < one block, row counts around the 128-row tile edge,
    more row tiles than SNP groups allow at 256 workgroups."""
    width = 256
    x, y, p, rng = make_problem(max(n, 8), K, width, 2, seed=K + n)
    net = build_net(x, y, p)
    assert net.lib.loc_l1_gemm_supported(net.d.Hp, 3)
    r = rng.permutation(x.shape[0])[:n].astype(np.int32)
    a1 = run_gemm(net, torch.from_numpy(r).cuda(), n, 3)
    ref, _ = _a1_reference(p, x[r])
    assert maxerr(a1[:n, :width], ref) < 2e-5, maxerr(a1[:n, :width], ref)
    assert np.isfinite(a1).all()


def test_gemm_handles_any_uint8_genotype_and_padded_width():
    """Genotype bytes up to 255 are exact in bf16; width 250 pads to 256 with zero units."""
    K, n, width = 1000, 200, 250
    x, y, p, rng = make_problem(n, K, width, 2, seed=3)
    x = rng.integers(0, 256, x.shape).astype(np.uint8)
    net = build_net(x, y, p)
    r = np.arange(n, dtype=np.int32)
    a1 = run_gemm(net, torch.from_numpy(r).cuda(), n, 3)
    ref, z = _a1_reference(p, x)
    assert maxerr(a1[:n, :width], ref) < 2e-5 * max(1.0, np.abs(z).max())
    assert not a1[:n, width:].any()


@pytest.mark.parametrize("pieces,tol_abs", [(2, 2e-4), (1, 2e-2)])
def test_gemm_fewer_pieces_within_stated_tolerance(pieces, tol_abs):
    K, width, n = 5830, 256, 200
    x, y, p, rng = make_problem(n, K, width, 2, seed=pieces)
    net = build_net(x, y, p)
    r = np.arange(n, dtype=np.int32)
    a1 = run_gemm(net, torch.from_numpy(r).cuda(), n, pieces)
    ref, z = _a1_reference(p, x[r])
    err = maxerr(a1[:n, :width], ref)
    assert err < tol_abs, err
    if pieces == 1:
        assert err <= 1e-2 * np.abs(z).max() and err > 2e-5


def test_gemm_is_deterministic_and_agrees_with_the_in_loop_conversion_kernel():
    from tests.test_gpu_rows import _run_rows
    K, width, n = 5830, 256, 300
    x, y, p, rng = make_problem(n, K, width, 2, seed=5)
    net = build_net(x, y, p)
    r = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
    a = run_gemm(net, r, n, 3)
    b = run_gemm(net, r, n, 3)
    assert np.array_equal(a, b)
    c = run_gemm(net, r, n, 3, target_blocks=24, scratch_tiles=24)      # a different SNP-group split
    assert maxerr(a[:n], c[:n]) < 5e-6
    d = _run_rows(net, r, n, 3)                                         # same exact products, other summation order
    assert maxerr(a[:n], d[:n]) < 5e-6


@pytest.mark.parametrize("pieces,target_blocks", [(1, 0), (2, 0), (3, 0), (1, 48), (2, 24), (3, 8)])
def test_gemm_every_loop_shape_matches_the_three_piece_result_structure(pieces, target_blocks):
    """The unrolled body holds 6 SNP blocks (1 or 2 pieces) or 2 (3 pieces) and a remainder of 2 or 4 blocks follows it;
    the number of SNP groups decides how many blocks a workgroup walks.  K = 20,000 is 313 blocks (odd: one zero tile
    pads the last pair); 1000 rows at the default 256 workgroups = 32 groups of 8-10 blocks, and fewer workgroups
    = longer walks (48 -> 6 groups of 52-54 blocks, 24 -> 3 of 104-106, 8 -> 1 group of all 314).  Every row and unit
    is compared with the fp64 forward."""
    K, width, n = 20000, 256, 1000
    x, y, p, rng = make_problem(n, K, width, 2, seed=11 + pieces)
    net = build_net(x, y, p)
    r = rng.permutation(n).astype(np.int32)
    a1 = run_gemm(net, torch.from_numpy(r).cuda(), n, pieces, target_blocks=target_blocks)
    ref, z = _a1_reference(p, x[r])
    err = maxerr(a1[:n, :width], ref)
    assert err < {3: 3e-5, 2: 4e-4, 1: 4e-2}[pieces], err


def test_gemm_rejects_what_it_cannot_do():
    from locator_amd import _lib
    x, y, p, rng = make_problem(40, 256, 128, 2, seed=9)
    net = build_net(x, y, p)
    assert not net.lib.loc_l1_gemm_supported(128, 3) and not net.lib.loc_l1_gemm_supported(256, 4)
    assert net.lib.loc_l1_image_bytes(C.byref(net.d), 3) == 0
    x, y, p, rng = make_problem(300, 256, 256, 2, seed=9)
    net = build_net(x, y, p)
    with pytest.raises(_lib.LocatorHipError, match="scratch too small"):
        run_gemm(net, torch.arange(300, dtype=torch.int32, device="cuda"), 300, 3, scratch_tiles=1)
