"""CPU check of the arithmetic fact the large-M layer-1 forward rests on (locator_amd/csrc/l1_rows.hip):
an fp32 value splits EXACTLY into three bf16 pieces by truncation, and a uint8 genotype times a bf16
piece is exact in fp32 — so three bf16 MFMAs with fp32 accumulation reproduce the fp32 contraction up
to summation order.  Pure NumPy restatement of the device code's bit manipulation."""
import numpy as np


def _trunc16(x):
    """Top 16 bits of an fp32 (= a bf16 value, kept in an fp32 container)."""
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def _split3(w):
    hi = _trunc16(w)
    r1 = (w - hi).astype(np.float32)
    mid = _trunc16(r1)
    lo = (r1 - mid).astype(np.float32)
    return hi, mid, lo


def test_three_truncated_pieces_reassemble_every_fp32_weight_exactly():
    rng = np.random.default_rng(0)
    w = np.concatenate([
        rng.normal(0, 0.01, 200000), rng.uniform(-4, 4, 200000), rng.normal(0, 1e-6, 50000),
        np.array([0.0, -0.0, 1.0, -1.0, 3.4e38, -3.4e38, 1.1754944e-38, 2.0 ** -100, 1 + 2.0 ** -23]),
    ]).astype(np.float32)
    hi, mid, lo = _split3(w)
    # the last piece needs no rounding: its low 16 bits are already zero, i.e. it IS a bf16 value
    assert not (lo.view(np.uint32) & np.uint32(0xFFFF)).any()
    # subtraction steps were exact and the pieces add back to w exactly (checked in float64)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64),
                          w.astype(np.float64))
    # pieces shrink by at least 2^-8 each (8 significand bits per piece)
    nz = w != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(w[nz]) * 2.0 ** -7)
    assert np.all(np.abs(lo[nz]) <= np.abs(w[nz]) * 2.0 ** -15)


def test_genotype_times_piece_is_exact_in_fp32():
    rng = np.random.default_rng(1)
    w = rng.normal(0, 0.05, 100000).astype(np.float32)
    x = rng.integers(0, 256, w.size).astype(np.float32)          # any uint8 is exact in bf16 (8 bits)
    for piece in _split3(w):
        p32 = (x * piece).astype(np.float32)
        assert np.array_equal(p32.astype(np.float64), x.astype(np.float64) * piece.astype(np.float64))


def test_split_contraction_equals_fp32_contraction_up_to_summation_order():
    """sum_k x (s_k w) through the three pieces == the same sum with unsplit fp32 weights, both
    accumulated in float64 (so only the split, not the order, could differ): identical."""
    rng = np.random.default_rng(2)
    K, H, M = 4096, 8, 16
    w = rng.normal(0, 0.02, (K, H)).astype(np.float32)
    s = rng.uniform(0.5, 2.0, K).astype(np.float32)
    ws = (w * s[:, None]).astype(np.float32)                      # the device scales in fp32 first
    x = rng.integers(0, 3, (M, K)).astype(np.float64)
    hi, mid, lo = _split3(ws.ravel())
    via_pieces = sum(x @ p.reshape(K, H).astype(np.float64) for p in (hi, mid, lo))
    assert np.array_equal(via_pieces, x @ ws.astype(np.float64))


def test_one_piece_rounding_is_bf16_round_to_nearest_even():
    """P = 1 path: u + 0x7FFF + ((u >> 16) & 1), then the top 16 bits."""
    rng = np.random.default_rng(3)
    w = rng.normal(0, 1, 100000).astype(np.float32)
    u = w.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    err = np.abs(r.astype(np.float64) - w.astype(np.float64))
    ulp = 2.0 ** (np.floor(np.log2(np.abs(w.astype(np.float64)))) - 7)      # bf16 spacing at |w|
    assert np.all(err <= ulp / 2 * (1 + 1e-12))
    # ties go to even: 1 + 2^-8 sits exactly between two bf16 values
    tie = np.array([1 + 2.0 ** -8, 1 + 3 * 2.0 ** -8], np.float32)
    ut = tie.view(np.uint32).astype(np.uint64)
    rt = ((ut + 0x7FFF + ((ut >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    assert rt[0] == 1.0 and rt[1] == np.float32(1 + 2.0 ** -6)
