"""The many-row forms of the hidden stack (stack_rows.hip: 32 rows per workgroup on v_mfma_f32_32x32x2_f32, and - round 5 -
16 rows per workgroup on v_mfma_f32_16x16x4_f32 where that needs fewer rounds of workgroups: up to 4096 rows per chunk) against
the row-parallel vector-ALU kernel (2, 4 or 8 rows per workgroup) they replace above 8 x compute units rows and against the
float64 forward of the oracle
(locator.py:319-325 layers 2..L + the two Dense(2) heads; model.predict, :414, :441)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import locator_oracle as O
from tests.gpu_util import build_net, make_problem, maxerr

pytestmark = pytest.mark.gpu


def _a1(p, x):
    """ELU output of layer 1 in inference mode, float64."""
    s = p["gamma"] / np.sqrt(p["mov_var"] + O.BN_EPS)
    xh = (x.astype(np.float64) - p["mov_mean"]) * s + p["beta"]
    return O.elu(xh @ p["W"][0] + p["b"][0])


def _rest(p, a):
    nl = len(p["W"]) - 2
    for l in range(1, nl):
        a = O.elu(a @ p["W"][l] + p["b"][l])
    y1 = a @ p["W"][nl] + p["b"][nl]
    return y1 @ p["W"][nl + 1] + p["b"][nl + 1]


def _run(net, a1_dev, n, form, with_dist):
    from locator_amd import _lib
    lay, d, P = net.lay, net.d, net.params.data_ptr()
    yhat = torch.full((n, 2), float("nan"), device="cuda")
    dist = torch.full((n,), float("nan"), device="cuda") if with_dist else None
    rows = torch.arange(n, dtype=torch.int32, device="cuda")
    _lib.check(net.lib.loc_stack_forward_eval_form(a1_dev.data_ptr(), P + 4 * lay.wh, P + 4 * lay.bh, P + 4 * lay.wa,
                                                   P + 4 * lay.ba, P + 4 * lay.wb, P + 4 * lay.bb, d.Hp, d.L, n,
                                                   rows.data_ptr() if with_dist else None, net.Y.data_ptr() if with_dist else None,
                                                   yhat.data_ptr(), dist.data_ptr() if with_dist else None, form, None))
    torch.cuda.synchronize()
    return yhat.cpu().numpy(), (dist.cpu().numpy() if with_dist else None)


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("n,width,nlayers", [(2, 256, 10), (15, 256, 3), (16, 256, 10), (17, 256, 2), (31, 256, 10), (32, 256, 2),
                                             (33, 250, 4), (100, 256, 10), (600, 256, 10), (1280, 256, 10), (3072, 256, 10), (3300, 256, 10), (4096, 230, 3),
                                             (8191, 256, 10)])
def test_matrix_pipe_form_equals_the_vector_alu_form_and_the_oracle(n, width, nlayers, form):
    """form 1 = 32-row tiles, form 2 = 16-row tiles (3072 / 4096 / 8191 rows among the shapes)."""
    x, y, p, rng = make_problem(n, 300, width, nlayers, seed=n + nlayers)
    net = build_net(x, y, p)
    assert net.lib.loc_stack_rows_supported(net.d.Hp, net.d.L)
    a1 = _a1(p, x)
    a1_dev = torch.zeros(((n + 127) // 128 * 128, net.d.Hp), device="cuda")
    a1_dev[:n, :width] = torch.from_numpy(a1.astype(np.float32)).cuda()
    ref = _rest(p, a1.astype(np.float32).astype(np.float64))
    got_m, dist_m = _run(net, a1_dev, n, form, True)
    got_v, dist_v = _run(net, a1_dev, n, -1, True)
    assert np.isfinite(got_m).all() and maxerr(got_m, ref) < 2e-5 and maxerr(got_v, ref) < 2e-5
    assert maxerr(got_m, got_v) < 1e-5
    assert maxerr(dist_m, O.euclid(ref, y)) < 2e-5 and maxerr(dist_m, dist_v) < 1e-5
    got_n, _ = _run(net, a1_dev, n, form, False)                  # without targets: same predictions
    assert np.array_equal(got_n, got_m)
    if form == 2 and 2049 <= n <= 4096:                           # what the default picks for this row count (256 compute units)
        got_d, _ = _run(net, a1_dev, n, 0, True)
        assert np.array_equal(got_d, got_m)
    if form == 1 and width == 256:
        # the vector-ALU form with 4 and 8 rows per workgroup (round 5: what the default takes for 513..1024 and 1025..2048
        # rows on 256 compute units): a row's sums do not depend on how many rows share its workgroup
        for rf in (-2, -3):
            got_r, dist_r = _run(net, a1_dev, n, rf, True)
            assert np.array_equal(got_r, got_v) and np.array_equal(dist_r, dist_v), rf
        if n <= 2048:
            got_d, _ = _run(net, a1_dev, n, 0, True)
            assert np.array_equal(got_d, got_v)


def test_only_width_256_takes_the_matrix_pipe_form():
    from locator_amd import _lib
    lib = _lib.load()
    assert lib.loc_stack_rows_supported(256, 10) and lib.loc_stack_rows_supported(256, 2)
    assert not lib.loc_stack_rows_supported(128, 10) and not lib.loc_stack_rows_supported(256, 1)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    assert lib.loc_stack_rows_min_rows() == 8 * ncu + 1         # up to 8 x CUs rows one round of vector-ALU workgroups covers them


@pytest.mark.parametrize("n", [3072, 5000])
def test_predict_through_both_forms_agrees(n):
    """loc_predict: the default takes a matrix-pipe form from 2049 rows per chunk; loc_tuning.stack_rows = -1 keeps the
    vector-ALU kernel.  Both against oracle.predict, int8 exact first layer with the group reduction fused into either."""
    x, y, p, rng = make_problem(700, 4000, 256, 10, seed=n)
    rows = torch.from_numpy((rng.permutation(n) % 700).astype(np.int32)).cuda()
    ref = O.predict(p, x[rows.cpu().numpy()])
    outs = []
    for tuning in ({}, {"stack_rows": -1}, {"stack_rows": 1, "gemm_reduce": 1}, {"stack_rows": 2, "gemm_reduce": 1}, {"stack_rows": 1}):
        net = build_net(x, y, p, tuning=tuning)
        yhat, dist = torch.zeros((n, 2), device="cuda"), torch.zeros(n, device="cuda")
        net.predict_rows(rows, n, yhat, dist)
        torch.cuda.synchronize()
        outs.append(yhat.cpu().numpy())
        assert maxerr(outs[-1], ref) < 2e-5
        assert maxerr(dist.cpu().numpy(), O.euclid(ref, y[rows.cpu().numpy()])) < 2e-5
    # the default = the 16-row form up to 4096 rows, the 32-row form at 5000; the group reduction fused into the stack launch
    # changes no bit of either
    assert np.array_equal(outs[3 if n <= 4096 else 4], outs[0])
    assert maxerr(outs[0], outs[1]) < 1e-5 and np.array_equal(outs[4], outs[2]) and maxerr(outs[3], outs[4]) < 1e-5
