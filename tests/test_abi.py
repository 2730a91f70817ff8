"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/locator_hip.h declares, ctypes signatures cover them all, and the host-only layout helpers
agree with the documented swizzle.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from locator_amd import _lib


def _header_symbols(repo_root):
    src = open(os.path.join(repo_root, "include", "locator_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(loc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(repo_root):
    lib = _lib.load()
    names = _header_symbols(repo_root)
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/locator_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.loc_version() == 1


def test_struct_sizes_match_header():
    assert C.sizeof(_lib.Dims) == 24
    assert C.sizeof(_lib.Layout) == 14 * 8


def _struct_fields(src, name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return [re.search(r"([A-Za-z_0-9]+)\s*$", decl.strip()).group(1) for decl in body.split(";") if decl.strip()]


def test_ctypes_structs_follow_the_header_field_for_field(repo_root):
    """loc_net / loc_dims / loc_layout in include/locator_hip.h, the ctypes Structures in locator_amd/_lib.py
    and the stub shown in INTEGRATION.md must list the same fields in the same order."""
    src = open(os.path.join(repo_root, "include", "locator_hip.h")).read()
    for cname, ctype in (("loc_dims", _lib.Dims), ("loc_layout", _lib.Layout), ("loc_net", _lib.Net), ("loc_tuning", _lib.Tuning)):
        assert _struct_fields(src, cname) == [f[0] for f in ctype._fields_], cname
    doc = open(os.path.join(repo_root, "INTEGRATION.md")).read()
    stub = re.search(r"class Net\(C\.Structure\):\s*_fields_ = \[(.*?)\]\n", doc, flags=re.S).group(1)
    assert re.findall(r'\("([a-z_0-9A-Z]+)"', stub) == [f[0] for f in _lib.Net._fields_]
    tstub = re.search(r"class Tuning\(C\.Structure\): _fields_ = \[\(n, C\.c_int\) for n in \((.*?)\)\]", doc).group(1)
    assert re.findall(r'"([a-z_0-9A-Z]+)"', tstub) == [f[0] for f in _lib.Tuning._fields_]
    # pointer / int64 / int / float members only: the natural-alignment size is what both compilers produce
    assert C.sizeof(_lib.Net) % 8 == 0


def test_dims_and_layout():
    d = _lib.make_dims(5830, 256, 10)
    assert (d.K, d.Kp, d.H, d.Hp, d.L, d.n_pre) == (5830, 5856, 256, 256, 10, 5)
    lay = _lib.param_layout(d)
    assert lay.w1 == 0 and lay.gamma == 5856 * 256 and lay.beta == lay.gamma + 5856
    assert lay.wh == lay.b1 + 256 and lay.bh == lay.wh + 9 * 256 * 256
    assert lay.n_trainable % 4 == 0 and lay.mov_var == lay.mov_mean + 5856
    assert lay.n_total == lay.mov_var + 5856
    d1 = _lib.make_dims(10, 256, 1)         # --nlayers 1: no Dense layer before the Dropout layer
    assert (d1.L, d1.n_pre) == (1, 0) and _lib.param_layout(d1).wh == _lib.param_layout(d1).bh
    d2 = _lib.make_dims(33, 100, 3)
    assert (d2.Kp, d2.Hp, d2.n_pre) == (64, 128, 1)
    with pytest.raises(_lib.LocatorHipError):
        _lib.make_dims(10, 256, 0)          # nlayers < 1
    with pytest.raises(_lib.LocatorHipError):
        _lib.make_dims(10, 1025, 4)            # LOC_MAX_WIDTH is 1024


def test_w1s_index_is_a_bijection_and_matches_mfma_layout():
    lib = _lib.load()
    Hp, Kp = 64, 96
    idx = np.array([[lib.loc_w1s_index(h, k, Hp) for k in range(Kp)] for h in range(Hp)])
    assert sorted(idx.ravel().tolist()) == list(range(Hp * Kp))
    # lane l = hi*32 + (k&31), float4 q, component c  <->  unit 8q + 4hi + c of the tile (accumulator rows)
    for (h, k) in [(0, 0), (5, 3), (37, 40), (63, 95)]:
        kt, kl, ht, hl = k >> 5, k & 31, h >> 5, h & 31
        q, hi, c = hl >> 3, (hl >> 2) & 1, hl & 3
        assert idx[h, k] == ((kt * (Hp // 32) + ht) * 4 + q) * 256 + (hi * 32 + kl) * 4 + c
    # a (k-tile, unit-tile) block is one contiguous 1024-float run
    blk = idx[32:64, 32:64]
    assert blk.min() == (1 * 2 + 1) * 1024 and blk.max() == blk.min() + 1023


def test_codecs_library_loads_and_exports_its_entry_points(repo_root):
    """libloc_codecs.so (csrc/codecs.c, host-side Blosc-1 / LZ4 chunk decoder of the zarr reader)."""
    lib = C.CDLL(os.path.join(repo_root, "locator_amd", "libloc_codecs.so"))
    for name in ("loc_lz4_decompress", "loc_blosc1_decompress", "loc_blosc1_info"):
        assert hasattr(lib, name), name


def test_train_chain_supported_is_decided_by_shape_alone():
    """loc_train_chain_supported is host logic over the loc_net fields (no kernel, no device memory): width padding to
    64, 128, 256 or 512, at least one hidden layer, batch <= 32, Dropout not on the BatchNorm output, 16-byte row pitch, and the chained
    kernel's 32-bit byte offsets (Kp * 1024 < 2^32: just under 4.2 million SNPs)."""
    lib = _lib.load()

    def net(K=100000, width=256, nlayers=10, drop=0.25, slot_rows=32, pitch=None, wht=1, grid=512):
        n = _lib.Net()
        n.d = _lib.make_dims(K, width, nlayers)
        n.drop_p = drop
        n.wht = wht                                # only tested for NULL here
        n.slot_rows, n.l1_bwd_grid = slot_rows, grid
        n.x_pitch = n.d.Kp if pitch is None else pitch
        return n

    ok = lambda **kw: bool(lib.loc_train_chain_supported(C.byref(net(**kw))))
    assert ok() and ok(width=225) and ok(nlayers=2) and ok(drop=0.0) and ok(K=31)
    assert ok(width=64) and ok(width=33) and ok(width=128) and ok(width=97)        # round 4: 2 and 4 unit tiles
    assert ok(width=512) and ok(width=481)                                         # ... and 16 (two per wave)
    assert not ok(width=224) and not ok(width=257) and not ok(width=32) and not ok(width=513) and not ok(width=1024)
    assert not ok(nlayers=1) and ok(nlayers=1, drop=0.0) is False
    assert not ok(slot_rows=128)                                     # --batch_size > 64
    assert ok(slot_rows=64) and not ok(slot_rows=64, width=128) and not ok(slot_rows=64, width=512)   # 33..64: two row blocks, width 256 only
    assert not ok(wht=None)                                          # no fused hidden stack
    assert not ok(pitch=100008)                                      # rows not 16-byte aligned
    assert ok(K=4194240) and not ok(K=4194304)                       # Kp * 1024 < 2^32
    assert ok(nlayers=3) and ok(nlayers=2, drop=0.25)                # Dropout after layer 1 is fine (n_pre = 1), only n_pre = 0 is not
