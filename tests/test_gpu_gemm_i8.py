"""GPU parity tests of the int8 large-M layer-1 GEMM (loc_l1_image_i8_build + loc_l1_forward_gemm_i8,
locator_amd/csrc/l1_gemm_i8.hip) against the fp64 oracle forward (oracle/locator_oracle.py, inference mode).

Reference lines: model.predict, /root/reference/locator/locator.py:414, :441; --jacknife, :683-747.
Tolerances on a1 = ELU(z1), |z1| = O(1):
  3 digits : 24-bit fixed point against each unit's largest weight, exact integer accumulation -> 2e-5 absolute, the
             bar of the exactly-split bf16 x 3 kernel (tests/test_gpu_gemm.py) and of the fp32-MFMA kernel
  2 digits : 16-bit fixed point -> each weight is off by at most delta_h / 2 = max_k|w'| 2^-16 (asserted from the
             decoded image); on z1 that is a random walk over the non-zero genotypes of a row
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import locator_oracle as O
from tests.gpu_util import build_net, make_problem, maxerr

pytestmark = pytest.mark.gpu

G8_HP, G8_TILE = 256, 16384


def _a1_reference(p, x):
    xh = (x.astype(np.float64) - p["mov_mean"]) / np.sqrt(p["mov_var"] + 1e-3) * p["gamma"] + p["beta"]
    z = xh @ p["W"][0] + p["b"][0]
    return np.where(z > 0, z, np.expm1(z)), z


def _bn4(net):
    from locator_amd import _lib
    d, lay, lib = net.d, net.lay, net.lib
    P = net.params.data_ptr()
    bn4 = torch.zeros(4 * d.Kp, device="cuda")
    _lib.check(lib.loc_bn_infer_scale_shift(d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta, P + 4 * lay.mov_mean,
                                            P + 4 * lay.mov_var, bn4.data_ptr(), None))
    return bn4


def build_image(net, digits):
    from locator_amd import _lib
    d, lay, lib = net.d, net.lay, net.lib
    bn4 = _bn4(net)
    image = torch.zeros(lib.loc_l1_image_i8_bytes(C.byref(d), digits), dtype=torch.uint8, device="cuda")
    _lib.check(lib.loc_l1_image_i8_build(C.byref(d), bn4.data_ptr(), net.params.data_ptr() + 4 * lay.w1, digits,
                                         image.data_ptr(), None), "loc_l1_image_i8_build")
    torch.cuda.synchronize()
    return image, bn4


def run_gemm_i8(net, rows, n, digits, x_max=2, target_blocks=0, scratch_tiles=256, image=None, unit_tiles=0):
    from locator_amd import _lib
    d, lay, lib = net.d, net.lay, net.lib
    if image is None:
        image, _ = build_image(net, digits)
    mp = (n + 127) // 128 * 128
    partial = torch.empty(scratch_tiles * 128 * d.Hp, device="cuda")
    a1 = torch.full((mp, d.Hp), float("nan"), device="cuda")
    _lib.check(lib.loc_l1_forward_gemm_i8(net.X.data_ptr(), net.X.stride(0), rows.data_ptr(), n, C.byref(d),
                                          image.data_ptr(), digits, x_max, net.params.data_ptr() + 4 * lay.b1,
                                          partial.data_ptr(), partial.numel(), a1.data_ptr(), target_blocks,
                                          C.byref(_lib.Tuning(gemm_i8_unit_tiles=unit_tiles)), None),
               "loc_l1_forward_gemm_i8")
    torch.cuda.synchronize()
    return a1.cpu().numpy()


def decode_image(image, d, digits):
    """(delta[Hp], q[K-padded][Hp]) from the device image: the layout documented in l1_gemm_i8.hip."""
    raw = image.cpu().numpy()
    nkt = ((d.Kp + 63) // 64 + 1) & ~1
    delta = raw[8 * G8_HP * 4: 9 * G8_HP * 4].view(np.float32).copy()
    from locator_amd import _lib
    tiles_off = int(_lib.load().loc_l1_image_i8_tiles_offset(C.byref(d)))    # cvec8 | delta | colmax | guard | scan shares | tiles
    t = raw[tiles_off: tiles_off + nkt * digits * G8_TILE].view(np.int8).reshape(nkt, digits, 4, G8_HP, 16)
    q = np.zeros((nkt, 4, G8_HP, 16), np.int64)
    for p in range(digits):
        q = q * 256 + t[:, p].astype(np.int64)
    return delta, q.transpose(0, 1, 3, 2).reshape(nkt * 64, G8_HP)          # [k][unit]


@pytest.mark.parametrize("digits", [2, 3])
def test_digit_image_is_the_rounded_fixed_point_weight_bit_for_bit(digits):
    """q = rint(fp32(s_k W1[k][h]) / delta_h) with delta_h the power of two the max pass picks; every digit in
    [-128, 127]; the shift term equals sum_k t_k W1[k][h]."""
    K, width, n = 5830, 256, 64
    x, y, p, rng = make_problem(n, K, width, 2, seed=digits)
    p["W"][0][:, 7] = 0.0                                    # an all-zero unit: delta = 1, digits 0
    p["W"][0][11, 3] *= 40.0                                  # one dominant weight sets that unit's scale
    net = build_net(x, y, p)
    image, bn4 = build_image(net, digits)
    delta, q = decode_image(image, net.d, digits)
    w = np.zeros((q.shape[0], width), np.float32)
    w[:K] = p["W"][0].astype(np.float32) * bn4[:K].cpu().numpy()[:, None]
    lim = {2: 32639, 3: 8355711}[digits]
    mx = np.abs(w).max(0)
    assert delta[7] == 1.0 and not q[:, 7].any()
    live = mx > 0
    assert np.all(np.log2(delta[:width][live]) % 1 == 0)
    ratio = mx[live] / delta[:width][live]
    assert np.all(ratio <= lim) and np.all(ratio > lim / 2 - 1), (ratio.min(), ratio.max())
    want = np.rint(w.astype(np.float64) / delta[:width].astype(np.float64)).astype(np.int64)
    assert np.array_equal(q[:, :width], want)
    assert not q[:, width:].any() and not q[K:].any()
    cvec = image[:8 * G8_HP * 4].cpu().numpy().view(np.float32).reshape(8, G8_HP).sum(0)
    shift = bn4[net.d.Kp: net.d.Kp + K].cpu().numpy().astype(np.float64)
    assert maxerr(cvec[:width], shift @ p["W"][0]) < 2e-5


@pytest.mark.parametrize("unit_tiles", [1, 2])
@pytest.mark.parametrize("K,n", [(5830, 450), (64, 1), (97, 130), (4096, 129), (3000, 300), (8192, 1000),
                                 (100, 128), (32, 5), (20000, 257)])
def test_three_digits_is_fp32_exact(K, n, unit_tiles):
    """K not a multiple of 64 / 128 (zero tail of the image, a zero tile pads an odd block count), K < one block, row
    counts around the 128-row tile edge, more row tiles than SNP groups allow at 256 workgroups."""
    width = 256
    x, y, p, rng = make_problem(max(n, 8), K, width, 2, seed=K + n)
    net = build_net(x, y, p)
    assert net.lib.loc_l1_gemm_i8_supported(net.d.Hp, 3)
    r = rng.permutation(x.shape[0])[:n].astype(np.int32)
    a1 = run_gemm_i8(net, torch.from_numpy(r).cuda(), n, 3, unit_tiles=unit_tiles)
    ref, _ = _a1_reference(p, x[r])
    assert maxerr(a1[:n, :width], ref) < 2e-5, maxerr(a1[:n, :width], ref)
    assert np.isfinite(a1).all()


def test_genotypes_up_to_127_and_padded_width():
    """Any int8-representable genotype is exact (x_max = 127 here); width 250 pads to 256 with zero units."""
    K, n, width = 1000, 200, 250
    x, y, p, rng = make_problem(n, K, width, 2, seed=3)
    x = rng.integers(0, 128, x.shape).astype(np.uint8)
    net = build_net(x, y, p)
    assert net.genotype_max() == int(x.max()) == 127
    r = np.arange(n, dtype=np.int32)
    a1 = run_gemm_i8(net, torch.from_numpy(r).cuda(), n, 3, x_max=127)
    ref, z = _a1_reference(p, x)
    assert maxerr(a1[:n, :width], ref) < 2e-5 * max(1.0, np.abs(z).max())
    assert not a1[:n, width:].any()


def test_two_digits_error_is_the_quantisation_step():
    """16-bit fixed point: a1 deviates from the fp64 forward by a random walk of per-weight errors <= delta_h / 2 over
    the row's non-zero genotypes - bounded here by 6 sigma of that walk, and far below the 1e-2 of plain bf16."""
    K, width, n = 5830, 256, 200
    x, y, p, rng = make_problem(n, K, width, 2, seed=2)
    net = build_net(x, y, p)
    image, _ = build_image(net, 2)
    delta, _ = decode_image(image, net.d, 2)
    r = np.arange(n, dtype=np.int32)
    a1 = run_gemm_i8(net, torch.from_numpy(r).cuda(), n, 2, image=image)
    ref, z = _a1_reference(p, x[r])
    err = np.abs(a1[:n, :width] - ref)
    walk = np.sqrt((x[r].astype(np.float64) ** 2).sum(1))[:, None] * delta[None, :width] / np.sqrt(12.0)
    assert np.all(err < 6.0 * walk + 2e-5), float((err / (6.0 * walk + 2e-5)).max())
    assert err.max() < 2e-3 and err.max() > 1e-6, err.max()


def test_deterministic_and_independent_of_the_group_split():
    from tests.test_gpu_gemm import run_gemm
    K, width, n = 5830, 256, 300
    x, y, p, rng = make_problem(n, K, width, 2, seed=5)
    net = build_net(x, y, p)
    r = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
    a = run_gemm_i8(net, r, n, 3)
    b = run_gemm_i8(net, r, n, 3)
    assert np.array_equal(a, b)
    c = run_gemm_i8(net, r, n, 3, target_blocks=24, scratch_tiles=24)      # a different SNP-group split
    assert maxerr(a[:n], c[:n]) < 5e-6
    d = run_gemm(net, r, n, 3)                                             # exactly-split bf16 x 3
    assert maxerr(a[:n], d[:n]) < 1e-5


@pytest.mark.parametrize("unit_tiles", [1, 2])
@pytest.mark.parametrize("digits,target_blocks", [(2, 0), (3, 0), (2, 48), (3, 24), (2, 8), (3, 8)])
def test_every_loop_shape(digits, target_blocks, unit_tiles):
    """The unrolled body holds 3 pairs of SNP blocks (eight waves, 12 fragments in flight) or 2 / 4 pairs (four waves,
    32 in flight: two planes / one plane) and a remainder follows; the number of SNP groups decides how
    many blocks a workgroup walks.  K = 20,000 is 313 blocks (odd: one zero tile pads the last pair); 1000 rows at the
    default 256 workgroups = 32 groups of 8-10 blocks; 48 -> 6 groups of 52-54, 24 -> 3 of 104-106, 8 -> 1 group of all
    314.  Every row and unit is compared with the fp64 forward."""
    K, width, n = 20000, 256, 1000
    x, y, p, rng = make_problem(n, K, width, 2, seed=11 + digits)
    net = build_net(x, y, p)
    r = rng.permutation(n).astype(np.int32)
    a1 = run_gemm_i8(net, torch.from_numpy(r).cuda(), n, digits, target_blocks=target_blocks, unit_tiles=unit_tiles)
    ref, z = _a1_reference(p, x[r])
    err = maxerr(a1[:n, :width], ref)
    assert err < {3: 3e-5, 2: 4e-3}[digits], err


def test_rejects_what_it_cannot_do():
    from locator_amd import _lib
    x, y, p, rng = make_problem(40, 256, 128, 2, seed=9)
    net = build_net(x, y, p)
    assert not net.lib.loc_l1_gemm_i8_supported(128, 3) and not net.lib.loc_l1_gemm_i8_supported(256, 1)
    assert net.lib.loc_l1_image_i8_bytes(C.byref(net.d), 3) == 0
    x, y, p, rng = make_problem(300, 256, 256, 2, seed=9)
    net = build_net(x, y, p)
    rows = torch.arange(300, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.LocatorHipError, match="scratch too small"):
        run_gemm_i8(net, rows, 300, 3, scratch_tiles=1)
    for bad in (0, 128, 255):
        with pytest.raises(_lib.LocatorHipError, match="0..127"):
            run_gemm_i8(net, rows, 300, 3, x_max=bad)
    # one group over 140,000 SNPs with genotypes up to 127: 127 * 128 * 140,032 > 2^31
    x, y, p, rng = make_problem(130, 140_000, 256, 2, seed=1)
    net = build_net(x, y, p)
    with pytest.raises(_lib.LocatorHipError, match="overflow int32"):
        run_gemm_i8(net, torch.arange(130, dtype=torch.int32, device="cuda"), 130, 2, x_max=127, target_blocks=2)


def test_predict_takes_the_int8_path_only_when_the_genotypes_allow_it():
    """loc_predict: >= 512 rows of genotypes <= 127 -> int8 image (exact digits by default); a genotype of 200 ->
    bf16 pieces, same predictions."""
    from oracle import locator_oracle as O
    K, width, n = 3000, 256, 1300
    x, y, p, rng = make_problem(n, K, width, 4, seed=21)
    ref = O.predict(p, x)
    outs = {}
    for name, xx, kw in (("i8", x, {}), ("i8fast", x, {"predict_digits": 2}), ("bf16", x, {"predict_digits": -1})):
        net = build_net(xx, y, p, **kw)
        rows = torch.arange(n, dtype=torch.int32, device="cuda")
        yhat = torch.zeros((n, 2), device="cuda")
        net.predict_rows(rows, n, yhat)
        torch.cuda.synchronize()
        outs[name] = yhat.cpu().numpy()
        want = {"i8": net.lib.loc_l1_image_i8_bytes(C.byref(net.d), 3), "i8fast": net.lib.loc_l1_image_i8_bytes(C.byref(net.d), 2),
                "bf16": net.lib.loc_l1_image_bytes(C.byref(net.d), 3)}[name]
        # (the exact mode runs under the dynamic-range guard, which may send the weights to the bf16 pieces: room for both)
        assert net.l1_image is not None and net.l1_image.numel() >= want
        if name != "bf16":
            assert net.cnet().x_max == 2
    assert maxerr(outs["i8"], ref) < 2e-5 and maxerr(outs["bf16"], ref) < 2e-5
    assert maxerr(outs["i8fast"], ref) < 1e-3 * np.abs(ref).max()
    x2 = x.copy()
    x2[5, 17] = 200
    net = build_net(x2, y, p)
    yhat = torch.zeros((n, 2), device="cuda")
    net.predict_rows(torch.arange(n, dtype=torch.int32, device="cuda"), n, yhat)
    torch.cuda.synchronize()
    assert net.genotype_max() == 200 and net.l1_image.numel() >= net.lib.loc_l1_image_bytes(C.byref(net.d), 3)
    assert maxerr(yhat.cpu().numpy(), O.predict(p, x2)) < 2e-5


def test_second_predict_with_unchanged_weights_reuses_the_image_and_any_change_rebuilds_it():
    """predict_locs predicts twice with the same weights (locator.py:414, :441): the second many-row call must skip the
    conversion (loc_net.l1_image_ready) and give identical numbers; a training step in between must rebuild it."""
    from oracle import locator_oracle as O
    K, width, n = 2000, 256, 700
    x, y, p, rng = make_problem(n, K, width, 4, seed=33)
    net = build_net(x, y, p)
    rows = torch.arange(n, dtype=torch.int32, device="cuda")
    y1, y2, y3 = (torch.zeros((n, 2), device="cuda") for _ in range(3))
    net.predict_rows(rows, n, y1)
    assert net._image_mode == 13 and net._net.l1_image_ready == 0
    t0 = int(net.lib.loc_l1_image_i8_tiles_offset(C.byref(net.d)))
    sl = slice(t0 + 8192, t0 + 12288)                 # inside the first digit tiles
    assert net.l1_image[sl].any()
    net.l1_image[sl].zero_()                          # if the second call rebuilt the image these bytes would come back
    tail = net.l1_image[sl].clone()
    net.predict_rows(rows, n, y2)
    torch.cuda.synchronize()
    assert net._net.l1_image_ready == 13 and torch.equal(net.l1_image[sl], tail)
    net.import_params(O.cast_params(p, np.float32))   # same values, but "changed": rebuild restores the zeroed tail
    net.predict_rows(rows, n, y3)
    torch.cuda.synchronize()
    assert net._net.l1_image_ready == 0 and not torch.equal(net.l1_image[sl], tail)
    assert torch.equal(y1, y3) and maxerr(y1.cpu().numpy(), O.predict(p, x)) < 2e-5
    loss = torch.zeros(1, device="cuda")
    mask = torch.ones(32 * width, dtype=torch.uint8, device="cuda")
    net.train_step(rows, 32, 1, mask, loss)
    assert net._image_mode == 0


def test_predict_over_more_rows_than_one_chunk():
    """loc_predict walks LOC_PREDICT_CHUNK = 16,384 rows per large-M launch with ONE weight image: 40,000 rows drawn from
    a 600-row matrix (two full chunks and a remainder of 7,232) against oracle.predict on the distinct rows."""
    from oracle import locator_oracle as O
    K, width, n_mat, n = 1500, 256, 600, 40_000
    x, y, p, rng = make_problem(n_mat, K, width, 4, seed=41)
    ref = O.predict(p, x)
    for digits in (3, 2):
        net = build_net(x, y, p, predict_digits=digits)
        r = rng.integers(0, n_mat, n).astype(np.int32)
        yhat = torch.zeros((n, 2), device="cuda")
        net.predict_rows(torch.from_numpy(r).cuda(), n, yhat)
        torch.cuda.synchronize()
        got = yhat.cpu().numpy()
        bar = 2e-5 if digits == 3 else 1e-3 * np.abs(ref).max()
        assert maxerr(got, ref[r]) < bar, (digits, maxerr(got, ref[r]))
        assert net._image_mode == 10 + digits


_BITCMP = """
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[2])
from locator_amd import _lib
if sys.argv[1] != "-":
    _lib.use_library(sys.argv[1])
from tests.gpu_util import build_net, make_problem
from tests.test_gpu_gemm import run_gemm
from tests.test_gpu_gemm_i8 import run_gemm_i8
x, y, p, rng = make_problem(1000, 20000, 256, 2, seed=7)
net = build_net(x, y, p)
r = torch.from_numpy(rng.permutation(1000).astype(np.int32)).cuda()
h = hashlib.sha256()
for d in (3, 2):
    for ut in (1, 2):
        h.update(run_gemm_i8(net, r, 1000, d, unit_tiles=ut)[:1000].tobytes())
for pcs in (3, 2, 1):
    h.update(run_gemm(net, r, 1000, pcs)[:1000].tobytes())
print("DIGEST", h.hexdigest())
"""


def test_hand_counted_vmcnt_build_equals_the_drained_build_bit_for_bit(repo_root, tmp_path):
    """the GEMM kernels issue their global loads from inline asm and count `s_waitcnt vmcnt(N)` by hand.
    `make debug_drain` (part of build()) compiles the same kernels with every count replaced by vmcnt(0); both libraries
    must give identical bytes on the same inputs - every int8 mode in both wave forms and every bf16 mode, K = 20,000
    (main body and remainders of the unrolled loops).  Separate processes: one library per process."""
    import os
    import subprocess
    import sys
    drain = os.path.join(repo_root, "locator_amd", "liblocator_hip_drain.so")
    if not os.path.exists(drain):
        pytest.skip("locator_amd/liblocator_hip_drain.so not built (make -C locator_amd/csrc debug_drain)")
    script = tmp_path / "bitcmp.py"
    script.write_text(_BITCMP)
    out = []
    for lib in ("-", drain):
        r = subprocess.run([sys.executable, str(script), lib, repo_root], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert out[0] == out[1], out


# ------------------------------------------------------------------ 2-bit packed genotypes
def _pack(net):
    from locator_amd import _lib
    X2 = torch.zeros((net.X.shape[0], net.d.Kp // 4), dtype=torch.uint8, device="cuda")
    _lib.check(net.lib.loc_pack_genotypes_2bit(net.X.data_ptr(), net.X.stride(0), net.X.shape[0], net.d.Kp, X2.data_ptr(),
                                               X2.stride(0), None), "loc_pack_genotypes_2bit")
    torch.cuda.synchronize()
    return X2


def test_pack_genotypes_2bit_matches_numpy():
    x, y, p, rng = make_problem(37, 1003, 256, 2, seed=5)
    x[rng.random(x.shape) < 0.05] = 3                       # 3 is a legal 2-bit value too
    net = build_net(x, y, p)
    X2 = _pack(net).cpu().numpy()
    xp = np.zeros((37, net.d.Kp), np.uint8)
    xp[:, :1003] = x
    q = xp.reshape(37, -1, 4)
    want = q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)
    assert np.array_equal(X2, want)


@pytest.mark.parametrize("digits", [2, 3])
@pytest.mark.parametrize("K,n,target_blocks", [(8192, 1000, 0), (5830, 450, 0), (40000, 4096, 0), (3000, 300, 8),
                                               (97, 130, 0), (20000, 700, 48)])
def test_packed_genotypes_give_the_same_bits_as_the_unpacked_call(K, n, target_blocks, digits):
    """loc_l1_forward_gemm_i8_packed on the 2-bit matrix against loc_l1_forward_gemm_i8 on the byte matrix: integer
    products and sums of the same numbers, so every activation must agree bit for bit - for every loop shape (one pair
    per group up to hundreds, K not a multiple of 128, rows not a multiple of 128, row indices that repeat)."""
    from locator_amd import _lib
    x, y, p, rng = make_problem(max(64, min(n, 1500)), K, 256, 2, seed=K % 71 + digits)
    net = build_net(x, y, p)
    d, lay, lib = net.d, net.lay, net.lib
    rows = torch.from_numpy(rng.integers(0, x.shape[0], n).astype(np.int32)).cuda()
    image, _ = build_image(net, digits)
    ref = run_gemm_i8(net, rows, n, digits, target_blocks=target_blocks, image=image)
    X2 = _pack(net)
    mp = (n + 127) // 128 * 128
    partial = torch.empty(256 * 128 * d.Hp, device="cuda")
    a1 = torch.full((mp, d.Hp), float("nan"), device="cuda")
    _lib.check(lib.loc_l1_forward_gemm_i8_packed(X2.data_ptr(), X2.stride(0), rows.data_ptr(), n, C.byref(d),
                                                 image.data_ptr(), digits, net.params.data_ptr() + 4 * lay.b1,
                                                 partial.data_ptr(), partial.numel(), a1.data_ptr(), target_blocks, None,
                                                 None), "loc_l1_forward_gemm_i8_packed")
    torch.cuda.synchronize()
    assert np.array_equal(ref[:n], a1.cpu().numpy()[:n])


def test_packed_call_rejects_a_misaligned_matrix():
    from locator_amd import _lib
    x, y, p, rng = make_problem(64, 2000, 256, 2, seed=9)
    net = build_net(x, y, p)
    d, lay, lib = net.d, net.lay, net.lib
    image, _ = build_image(net, 2)
    X2 = _pack(net)
    rows = torch.arange(640, dtype=torch.int32, device="cuda") % 64
    partial = torch.empty(256 * 128 * d.Hp, device="cuda")
    a1 = torch.zeros((640, d.Hp), device="cuda")
    rc = lib.loc_l1_forward_gemm_i8_packed(X2.data_ptr() + 1, X2.stride(0), rows.data_ptr(), 640, C.byref(d), image.data_ptr(), 2,
                                           net.params.data_ptr() + 4 * lay.b1, partial.data_ptr(), partial.numel(),
                                           a1.data_ptr(), 0, None, None)
    assert rc != 0 and "packed" in lib.loc_last_error().decode()
    rc = lib.loc_pack_genotypes_2bit(net.X.data_ptr(), net.X.stride(0), 64, d.Kp, X2.data_ptr(), d.Kp // 4 - 4, None)
    assert rc != 0


def test_predict_reads_the_packed_matrix_when_it_is_there_and_gives_the_same_bits():
    """LocatorNet.pack_genotypes() + loc_predict: 4096 rows go through loc_l1_forward_gemm_i8_packed (loc_net.X2 set, chunk
    >= LOC_GEMM_I8_PACKED_MIN_ROWS), 1000 rows through the unpacked call; predictions are bit-identical either way, and a
    matrix with a genotype above 3 is left unpacked."""
    x, y, p, rng = make_problem(700, 6000, 256, 4, seed=12)
    net = build_net(x, y, p)
    rows = torch.from_numpy(rng.integers(0, 700, 4096).astype(np.int32)).cuda()
    out = {}
    for packed in (False, True):
        if packed:
            assert net.pack_genotypes() and net.X.loc_x2.shape == (700, net.d.Kp // 4)
        for n in (4096, 1000):
            yhat = torch.zeros((n, 2), device="cuda")
            net.predict_rows(rows, n, yhat)
            torch.cuda.synchronize()
            out[(packed, n)] = yhat.cpu().numpy().copy()
        assert bool(net._net.X2) == packed
    assert np.array_equal(out[(False, 4096)], out[(True, 4096)]) and np.array_equal(out[(False, 1000)], out[(True, 1000)])
    ref = O.predict(p, x[rows.cpu().numpy()])
    assert maxerr(out[(True, 4096)], ref) < 5e-5
    x2 = x.copy()
    x2[3, 17] = 4
    net2 = build_net(x2, y, p)
    assert net2.pack_genotypes() is False and getattr(net2.X, "loc_x2", None) is None
    # the default since round 6: no predict packs the matrix on its own (the pass costs eight predicts' worth of the gain);
    # with auto_pack (--predict_packed) the first predict of >= 3072 rows does, a smaller one does not
    net3 = build_net(x, y, p)
    yhat = torch.zeros((4096, 2), device="cuda")
    net3.predict_rows(rows, 4096, yhat)
    torch.cuda.synchronize()
    assert getattr(net3.X, "loc_x2", None) is None and np.array_equal(yhat.cpu().numpy(), out[(False, 4096)])
    net3.auto_pack = True
    net3.predict_rows(rows, 1000, yhat)
    assert getattr(net3.X, "loc_x2", None) is None
    net3.predict_rows(rows, 4096, yhat)
    torch.cuda.synchronize()
    assert getattr(net3.X, "loc_x2", None) is not None and np.array_equal(yhat.cpu().numpy(), out[(True, 4096)])


@pytest.mark.parametrize("n", [600, 1300, 3300])
def test_group_reduction_fused_into_the_stack_launch_gives_the_same_bits(n):
    """loc_tuning.gemm_reduce = 1: the int8 GEMM leaves its SNP-group partial sums and the hidden-stack launch adds them up in
    its input stage (loc_l1_forward_gemm_i8_partial + loc_stack_forward_eval_partial) instead of the dedicated reduction
    launch of the default: same association of the sums, so identical predictions and distances - for byte genotypes and
    (3300 rows) for the 2-bit packed ones and the matrix-pipe form of the stack, two and three digit planes.  (Measured slower
    than the default, kept as a switch: include/locator_hip.h.)"""
    K, width = 2500, 256
    x, y, p, rng = make_problem(n, K, width, 4, seed=n)
    outs = []
    rows = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
    for digits in (3, 2):
        pair = []
        for tuning in ({}, {"gemm_reduce": 1}):
            net = build_net(x, y, p, predict_digits=digits, tuning=tuning)
            yhat, dist = torch.zeros((n, 2), device="cuda"), torch.zeros(n, device="cuda")
            net.predict_rows(rows, n, yhat, dist)
            torch.cuda.synchronize()
            pair.append((yhat.cpu().numpy(), dist.cpu().numpy(), rows))
        assert np.array_equal(pair[0][0], pair[1][0]) and np.array_equal(pair[0][1], pair[1][1]), digits
        outs.append(pair[0][0])
    from oracle import locator_oracle as O
    ref = O.predict(p, x[rows.cpu().numpy()])
    assert maxerr(outs[0], ref) < 2e-5
