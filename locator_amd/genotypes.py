"""Genotype ingest and SNP filtering for the locator path, without scikit-allel / zarr / h5py.

Mirrors /root/reference/locator/locator.py:187-281 (load_genotypes, replace_md, filter_snps).
The reference delegates to scikit-allel (`read_vcf`, `GenotypeArray.count_alleles / is_biallelic /
to_allele_counts / is_missing`) and zarr; neither is available on the target, so the small subset
of their behaviour that the path relies on is implemented here on plain NumPy:

  genotypes : int8 array (variants, samples, ploidy=2), -1 = missing allele  (allel.GenotypeArray)
  count_alleles()[:, a]      number of called alleles equal to a, per variant
  is_biallelic()             exactly two distinct alleles observed at the variant
  to_allele_counts()[:,:,1]  per-sample count of allele 1 (missing alleles count as 0)
  is_missing()               any allele of the call is < 0
"""
from __future__ import annotations

import gzip
import json
import os
import struct
import zlib

import numpy as np


# ------------------------------------------------------------------ VCF
def _parse_gt_fast(fields, n):
    """All calls are single-digit diploid 'a|b' / 'a/b' with GT the only (or first) FORMAT key."""
    s = "\t".join(fields)
    if len(s) != 4 * n - 1:
        return None
    arr = np.frombuffer((s + "\t").encode("ascii"), dtype=np.uint8).reshape(n, 4)
    sep = arr[:, 1]
    if not np.all((sep == 124) | (sep == 47)) or not np.all(arr[:, 3] == 9):
        return None
    out = np.empty((n, 2), np.int8)
    for j, col in enumerate((0, 2)):
        c = arr[:, col]
        digit = (c >= 48) & (c <= 57)
        if not np.all(digit | (c == 46)):
            return None
        out[:, j] = np.where(digit, c.astype(np.int16) - 48, -1).astype(np.int8)
    return out


def _parse_gt_slow(fields, n):
    out = np.full((n, 2), -1, np.int8)
    for i, f in enumerate(fields):
        gt = f.split(":", 1)[0].replace("|", "/").split("/")
        for j, a in enumerate(gt[:2]):
            if a != "." and a != "":
                out[i, j] = int(a)
        if len(gt) == 1:            # haploid call: second allele stays missing (-1), as allel pads
            out[i, 1] = -1
    return out


def read_vcf(path):
    """Subset of allel.read_vcf used by locator.py:195-199: returns dict with 'calldata/GT'
    (variants, samples, 2) int8, 'samples' (object array of str) and 'variants/POS' (int32)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    samples, gts, pos = None, [], []
    with opener(path, "rt") as fh:
        for line in fh:
            if line.startswith("##"):
                continue
            line = line.rstrip("\n")
            if line.startswith("#CHROM"):
                samples = np.array(line.split("\t")[9:], dtype=object)
                continue
            if not line:
                continue
            f = line.split("\t")
            n = len(f) - 9
            fmt = f[8]
            g = None
            if fmt == "GT":
                g = _parse_gt_fast(f[9:], n)
            if g is None:
                if fmt.split(":")[0] != "GT":
                    raise ValueError("VCF FORMAT must start with GT")
                g = _parse_gt_slow(f[9:], n)
            gts.append(g)
            pos.append(int(f[1]))
    if samples is None:
        raise ValueError(f"{path}: no #CHROM header line")
    gt = np.stack(gts, axis=0) if gts else np.zeros((0, len(samples), 2), np.int8)
    return {"calldata/GT": gt, "samples": samples, "variants/POS": np.asarray(pos, dtype=np.int32)}


# ------------------------------------------------------------------ zarr v2 directory store
_CODECS = None


def _codecs():
    """libloc_codecs.so (locator_amd/csrc/codecs.c): the Blosc-1 / LZ4 chunk decoder, built with the HIP library."""
    global _CODECS
    if _CODECS is None:
        import ctypes as C
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libloc_codecs.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it with `make -C locator_amd/csrc` (blosc-compressed zarr "
                               "chunks cannot be read without it)")
        lib = C.CDLL(path)
        lib.loc_blosc1_decompress.restype = C.c_int64
        lib.loc_blosc1_decompress.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        lib.loc_blosc1_info.restype = C.c_int
        lib.loc_blosc1_info.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]
        lib.loc_lz4_decompress.restype = C.c_int64
        lib.loc_lz4_decompress.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
        lib.loc_zstd_available.restype = C.c_int
        lib.loc_zstd_decompress.restype = C.c_int64
        lib.loc_zstd_decompress.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
        lib.loc_zstd_compress.restype = C.c_int64
        lib.loc_zstd_compress.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]
        lib.loc_snp_flags.restype = C.c_int64
        lib.loc_snp_flags.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
        lib.loc_snp_allele_counts.restype = None
        lib.loc_snp_allele_counts.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
        lib.loc_rows_transposed.restype = None
        lib.loc_rows_transposed.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                            C.c_int64]
        _CODECS = lib
    return _CODECS


NO_ZSTD = ("this store's chunks are zstd-compressed and libzstd.so.1 was not found on this machine; install the zstd "
           "runtime library (libzstd1) or re-encode the store with Blosc(cname='lz4') / zlib")


def zstd_available():
    return bool(_codecs().loc_zstd_available())


def zstd_decompress(raw, nbytes):
    """One zstd frame of known decoded size (the numcodecs `zstd` compressor; also the payload of Blosc codec 4)."""
    out = np.empty(max(int(nbytes), 1), np.uint8)
    r = _codecs().loc_zstd_decompress(bytes(raw), len(raw), out.ctypes.data, int(nbytes))
    if r == -5:
        raise RuntimeError(NO_ZSTD)
    if r != nbytes:
        raise ValueError("malformed zstd chunk")
    return out[:nbytes].tobytes()


def zstd_compress(raw, level=3):
    """Writer side, for test / synthetic stores only."""
    raw = bytes(raw)
    out = np.empty(len(raw) + len(raw) // 128 + 256, np.uint8)
    r = _codecs().loc_zstd_compress(raw, len(raw), out.ctypes.data, out.size, int(level))
    if r == -5:
        raise RuntimeError(NO_ZSTD)
    if r < 0:
        raise ValueError("zstd compression failed")
    return out[:r].tobytes()


def blosc_decompress(raw):
    """One Blosc-1 chunk (what numcodecs.Blosc / `allel.vcf_to_zarr` write by default: LZ4, byte-shuffle) -> bytes.
    Codecs: lz4 / lz4hc, zlib, zstd (through the system's libzstd), stored; filters: byte-shuffle.  blosclz /
    bit-shuffle raise."""
    import ctypes as C
    lib = _codecs()
    info = (C.c_int64 * 5)()
    if lib.loc_blosc1_info(raw, len(raw), info) != 0:
        raise ValueError("blosc chunk shorter than its header")
    nbytes, blocksize = int(info[0]), int(info[1])
    out = np.empty(nbytes, np.uint8)
    tmp = np.empty(max(blocksize, 1), np.uint8)
    for mode in (0, 1, 2):            # as the header says; then the two split conventions of writers that do not say
        r = lib.loc_blosc1_decompress(raw, len(raw), out.ctypes.data, nbytes, tmp.ctypes.data, mode)
        if r == nbytes:
            return out.tobytes()
        if r == -5:
            raise RuntimeError(NO_ZSTD)
        if r in (-2, -3):
            what = "codec (supported: lz4, zlib, zstd, stored)" if r == -2 else "filter bit-shuffle"
            raise ValueError(f"blosc chunk uses an unsupported {what}; re-encode the store with "
                             "Blosc(cname='lz4') or zlib")
    raise ValueError("malformed blosc chunk")


def _lz4_compress_block(data):
    """Greedy LZ4 block encoder (hash of 4-byte windows, one candidate per hash).  Only used to WRITE small
    synthetic / test stores in the format real tools produce; the reader side is csrc/codecs.c."""
    n = len(data)
    out = bytearray()
    table = {}
    i = anchor = 0
    limit = n - 12                              # format rule: the last match starts >= 12 bytes before the end
    while i < limit:
        key = data[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 65535:
            m = 4
            end = n - 5                         # and the last 5 bytes are literals
            while i + m < end and data[cand + m] == data[i + m]:
                m += 1
            lit = i - anchor
            token = (min(lit, 15) << 4) | min(m - 4, 15)
            out.append(token)
            if lit >= 15:
                r = lit - 15
                while r >= 255:
                    out.append(255)
                    r -= 255
                out.append(r)
            out += data[anchor:i]
            out += (i - cand).to_bytes(2, "little")
            if m - 4 >= 15:
                r = m - 4 - 15
                while r >= 255:
                    out.append(255)
                    r -= 255
                out.append(r)
            i += m
            anchor = i
        else:
            i += 1
    lit = n - anchor
    out.append(min(lit, 15) << 4)
    if lit >= 15:
        r = lit - 15
        while r >= 255:
            out.append(255)
            r -= 255
        out.append(r)
    out += data[anchor:]
    return bytes(out)


def blosc_compress(raw, typesize=1, shuffle=True, blocksize=1 << 16, cname="lz4"):
    """Blosc-1 chunk with LZ4 (or zstd) streams and byte-shuffle, laid out as c-blosc writes it (16-byte header, block
    offsets, per-stream {int32 size, data}, split into `typesize` streams when the block is large enough)."""
    raw = bytes(raw)
    nbytes = len(raw)
    typesize = max(1, int(typesize))
    blocksize = max(typesize, min(blocksize, nbytes) // typesize * typesize) if nbytes else typesize
    do_shuffle = shuffle and typesize > 1
    flags = ({"lz4": 1, "zstd": 4}[cname] << 5) | (1 if do_shuffle else 0)
    nblocks = (nbytes + blocksize - 1) // blocksize if nbytes else 0
    body, starts = bytearray(), []
    pos = 16 + 4 * nblocks
    for b in range(nblocks):
        blk = raw[b * blocksize:(b + 1) * blocksize]
        leftover = len(blk) != blocksize
        if do_shuffle:
            ne = len(blk) // typesize
            arr = np.frombuffer(blk[:ne * typesize], np.uint8).reshape(ne, typesize).T.tobytes() + blk[ne * typesize:]
        else:
            arr = blk
        split = typesize > 1 and typesize <= 16 and blocksize // typesize >= 128 and not leftover
        ns = typesize if split else 1
        ss = len(arr) // ns
        starts.append(pos)
        for k in range(ns):
            stream = arr[k * ss:(k + 1) * ss]
            comp = _lz4_compress_block(stream) if cname == "lz4" else zstd_compress(stream)
            if len(comp) >= len(stream):
                comp = stream                                    # stored
            body += len(comp).to_bytes(4, "little") + comp
            pos += 4 + len(comp)
    head = bytes([2, 1, flags, typesize]) + nbytes.to_bytes(4, "little") + blocksize.to_bytes(4, "little") + \
        (16 + 4 * nblocks + len(body)).to_bytes(4, "little")
    return head + b"".join(x.to_bytes(4, "little") for x in starts) + bytes(body)


class ZarrArray:
    """Read-only zarr-v2 array in a directory store.  Compressors: none, zlib, gzip and blosc (the default of
    `allel.vcf_to_zarr`, scripts/vcf_to_zarr.py:12: Blosc-1 container with LZ4 / zlib payloads and byte-shuffle,
    decoded by csrc/codecs.c).  Filters: vlen-utf8 (object string arrays)."""

    def __init__(self, path):
        self.path = path
        with open(os.path.join(path, ".zarray")) as fh:
            meta = json.load(fh)
        if meta.get("zarr_format") != 2:
            raise ValueError(f"{path}: only zarr_format 2 is supported")
        self.shape = tuple(meta["shape"])
        self.chunks = tuple(meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self.order = meta.get("order", "C")
        self.fill_value = meta.get("fill_value")
        self.compressor = meta.get("compressor")
        self.filters = meta.get("filters") or []
        self.sep = meta.get("dimension_separator", ".")
        self.ndim = len(self.shape)
        self.vlen_utf8 = any(f.get("id") == "vlen-utf8" for f in self.filters)
        if self.compressor is not None and self.compressor.get("id") not in ("zlib", "gzip", "blosc", "zstd"):
            raise ValueError(f"{path}: compressor {self.compressor.get('id')!r} is not supported here "
                             "(supported: null, zlib, gzip, blosc, zstd)")

    def __len__(self):
        return self.shape[0]

    def _decode(self, raw, cshape):
        if self.compressor is not None and self.compressor.get("id") == "blosc":
            raw = blosc_decompress(raw)
        elif self.compressor is not None and self.compressor.get("id") == "zstd":
            if self.vlen_utf8:
                raise ValueError(f"{self.path}: zstd-compressed string arrays are not supported")
            raw = zstd_decompress(raw, int(np.prod(cshape)) * self.dtype.itemsize)
        elif self.compressor is not None:
            raw = zlib.decompress(raw, 15 + 32)        # auto-detect zlib / gzip framing
        if self.vlen_utf8:
            n = struct.unpack_from("<I", raw, 0)[0]
            out = np.empty(n, dtype=object)
            off = 4
            for i in range(n):
                ln = struct.unpack_from("<I", raw, off)[0]
                off += 4
                out[i] = raw[off:off + ln].decode("utf-8")
                off += ln
            return out.reshape(cshape, order=self.order)
        return np.frombuffer(raw, dtype=self.dtype).reshape(cshape, order=self.order)

    def _chunk(self, idx):
        fn = os.path.join(self.path, self.sep.join(str(i) for i in idx))
        if not os.path.exists(fn):
            fill = 0 if self.fill_value is None else self.fill_value
            return np.full(self.chunks, fill, dtype=object if self.vlen_utf8 else self.dtype)
        with open(fn, "rb") as fh:
            return self._decode(fh.read(), self.chunks)

    def __getitem__(self, key):
        """Supports a[:], a[a:b] and a[a:b, :, :] (slicing on the first axis only)."""
        if isinstance(key, tuple):
            if any(k != slice(None) for k in key[1:]):
                raise IndexError("ZarrArray: only the first axis can be sliced")
            key = key[0]
        if not isinstance(key, slice):
            raise IndexError("ZarrArray: use a slice")
        start, stop, step = key.indices(self.shape[0])
        if step != 1:
            raise IndexError("ZarrArray: step must be 1")
        n0 = max(stop - start, 0)
        out = np.empty((n0,) + self.shape[1:], dtype=object if self.vlen_utf8 else self.dtype)
        if n0 == 0:
            return out
        c0 = self.chunks[0]
        grid = [range((s + c - 1) // c) for s, c in zip(self.shape[1:], self.chunks[1:])]
        for ci in range(start // c0, (stop - 1) // c0 + 1):
            lo, hi = max(start, ci * c0), min(stop, (ci + 1) * c0)

            def rec(prefix, dims):
                if not dims:
                    ch = self._chunk((ci,) + tuple(prefix))
                    sl = [slice(lo - ci * c0, hi - ci * c0)]
                    dst = [slice(lo - start, hi - start)]
                    for ax, j in enumerate(prefix):
                        c = self.chunks[ax + 1]
                        a, b = j * c, min((j + 1) * c, self.shape[ax + 1])
                        sl.append(slice(0, b - a))
                        dst.append(slice(a, b))
                    out[tuple(dst)] = ch[tuple(sl)]
                    return
                for j in dims[0]:
                    rec(prefix + [j], dims[1:])

            rec([], grid)
        return out

    def read_into(self, out, start, stop, threads=4):
        """out[...] = self[start:stop] for a caller-owned array (e.g. a view of pinned memory that is uploaded as it is:
        the --windows loop, locator.py:539).  Uncompressed chunks that span the trailing axes are read straight from the
        file into `out` (one copy, page cache -> destination); compressed chunks are decoded on `threads` threads (the
        C codecs release the GIL) and copied in."""
        start, stop, _ = slice(start, stop).indices(self.shape[0])
        n0 = max(stop - start, 0)
        if tuple(out.shape) != (n0,) + tuple(self.shape[1:]) or out.dtype != self.dtype or not out.flags.c_contiguous:
            raise ValueError("read_into: `out` must be a C-contiguous array of the slice's shape and dtype")
        if n0 == 0:
            return out
        c0 = self.chunks[0]
        whole = tuple(self.chunks[1:]) == tuple(self.shape[1:]) and self.order == "C" and not self.vlen_utf8
        if not whole:
            out[...] = self[start:stop]
            return out
        row = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        flat = out.reshape(n0, -1).view(np.uint8).reshape(-1)
        tail = (0,) * (self.ndim - 1)

        def one(ci):
            lo, hi = max(start, ci * c0), min(stop, (ci + 1) * c0)
            dst = memoryview(flat)[(lo - start) * row:(hi - start) * row]
            fn = os.path.join(self.path, self.sep.join(str(i) for i in (ci,) + tail))
            if not os.path.exists(fn):
                np.frombuffer(dst, dtype=np.uint8)[...] = 0 if self.fill_value is None else self.fill_value
            elif self.compressor is None:
                with open(fn, "rb", buffering=0) as fh:
                    fh.seek((lo - ci * c0) * row)
                    got, want = 0, len(dst)
                    while got < want:
                        k = fh.readinto(dst[got:])
                        if not k:
                            raise IOError(f"{fn}: chunk shorter than its shape")
                        got += k
            else:
                with open(fn, "rb") as fh:
                    ch = self._decode(fh.read(), self.chunks)
                np.frombuffer(dst, dtype=self.dtype).reshape((hi - lo,) + tuple(self.shape[1:]))[...] = ch[lo - ci * c0:hi - ci * c0]

        cis = list(range(start // c0, (stop - 1) // c0 + 1))
        if len(cis) > 1 and threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(min(threads, len(cis))) as ex:
                list(ex.map(one, cis))
        else:
            for ci in cis:
                one(ci)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self[:]
        return a.astype(dtype) if dtype is not None else a


class ZarrGroup:
    """`zarr.open_group(path, mode='r')` stand-in: group['calldata/GT'] -> ZarrArray."""

    def __init__(self, path):
        if not os.path.isdir(path):
            raise FileNotFoundError(path)
        self.path = path

    def __getitem__(self, key):
        p = os.path.join(self.path, *key.split("/"))
        if os.path.exists(os.path.join(p, ".zarray")):
            return ZarrArray(p)
        if os.path.isdir(p):
            return ZarrGroup(p)
        raise KeyError(key)


def open_group(path, mode="r"):
    return ZarrGroup(path)


def write_zarr_array(path, arr, chunks, compressor=None):
    """Minimal zarr-v2 writer (uncompressed, zlib, "blosc" = Blosc-1 / LZ4 / byte-shuffle as `allel.vcf_to_zarr` writes
    by default, "blosc-zstd" = the same container with zstd streams, "zstd" = numcodecs' plain zstd frames) used to
    build synthetic stores (config 4) and test fixtures."""
    arr = np.asarray(arr)
    os.makedirs(path, exist_ok=True)
    chunks = tuple(int(min(c, s)) if s else 1 for c, s in zip(chunks, arr.shape))
    meta = {"zarr_format": 2, "shape": list(arr.shape), "chunks": list(chunks), "dtype": arr.dtype.str,
            "compressor": {None: None, "zlib": {"id": "zlib", "level": 1},
                           "blosc": {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0},
                           "blosc-zstd": {"id": "blosc", "cname": "zstd", "clevel": 5, "shuffle": 1, "blocksize": 0},
                           "zstd": {"id": "zstd", "level": 3}}[compressor], "fill_value": 0,
            "order": "C", "filters": None}
    with open(os.path.join(path, ".zarray"), "w") as fh:
        json.dump(meta, fh)
    ranges = [range((s + c - 1) // c) for s, c in zip(arr.shape, chunks)]
    import itertools
    for idx in itertools.product(*ranges):
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, arr.shape))
        block = np.zeros(chunks, arr.dtype)
        block[tuple(slice(0, s.stop - s.start) for s in sl)] = arr[sl]
        raw = block.tobytes()
        if compressor == "zlib":
            raw = zlib.compress(raw, 1)
        elif compressor == "blosc":
            raw = blosc_compress(raw, arr.dtype.itemsize)
        elif compressor == "blosc-zstd":
            raw = blosc_compress(raw, arr.dtype.itemsize, cname="zstd")
        elif compressor == "zstd":
            raw = zstd_compress(raw)
        with open(os.path.join(path, ".".join(map(str, idx))), "wb") as fh:
            fh.write(raw)


def write_callset_zarr(path, gt, pos, samples, chunk_variants=65536, compressor=None):
    """Synthetic `allel.vcf_to_zarr`-shaped store: calldata/GT, variants/POS, samples."""
    os.makedirs(path, exist_ok=True)
    for g in ("", "calldata", "variants"):
        os.makedirs(os.path.join(path, g), exist_ok=True)
        with open(os.path.join(path, g, ".zgroup"), "w") as fh:
            json.dump({"zarr_format": 2}, fh)
    gt = np.asarray(gt, dtype=np.int8)
    write_zarr_array(os.path.join(path, "calldata", "GT"), gt, (chunk_variants, gt.shape[1], 2), compressor)
    write_zarr_array(os.path.join(path, "variants", "POS"), np.asarray(pos, np.int32), (chunk_variants,), compressor)
    s = np.asarray(samples)
    write_zarr_array(os.path.join(path, "samples"), s.astype("U"), (len(s),), None)


# ------------------------------------------------------------------ tab-delimited count matrix
def read_matrix(path):
    """locator.py:200-227: first column 'sampleID', then one column per site holding 0/1/2 counts.
    The reference turns count c into haplotypes (1 if c>=1, 1 if c==2) and re-counts; other values
    contribute nothing.  Returns the equivalent (variants, samples, 2) int8 genotype array."""
    import pandas as pd
    gmat = pd.read_csv(path, sep="\t")
    samples = np.array(gmat["sampleID"])
    g = np.array(gmat.drop(labels="sampleID", axis=1), dtype="int8")      # samples x sites
    h1 = ((g == 1) | (g == 2)).astype(np.int8)
    h2 = (g == 2).astype(np.int8)
    gt = np.stack([h1.T, h2.T], axis=2)
    return gt, samples


# ------------------------------------------------------------------ allel.GenotypeArray subset
def count_alleles(gt, max_allele=None):
    """(variants, max_allele+1) int32 counts of called alleles; max_allele defaults to the data's max."""
    if max_allele is None:
        max_allele = int(gt.max()) if gt.size else 0
    max_allele = max(max_allele, 0)
    out = np.zeros((gt.shape[0], max_allele + 1), np.int32)
    flat = gt.reshape(gt.shape[0], -1)
    for a in range(max_allele + 1):
        out[:, a] = np.count_nonzero(flat == a, axis=1)
    return out


def is_biallelic(ac):
    return (ac > 0).sum(axis=1) == 2


def to_allele_counts_1(gt):
    """to_allele_counts()[:, :, 1]: per-sample number of allele-1 copies; int8 (variants, samples).
    Written as byte adds over the ploidy axis: `(gt == 1).sum(axis=2)` goes through int64 and is ~100x slower
    on a (150k, 765, 2) window, which makes the serial window prologue the Amdahl term of --windows."""
    b = (gt == 1).view(np.uint8)
    out = b[:, :, 0].copy()
    for p in range(1, gt.shape[2]):
        out += b[:, :, p]
    return out.view(np.int8)


def is_missing(gt):
    return (gt < 0).any(axis=2)


def replace_md(gt, rng=np.random):
    """locator.py:251-262: impute missing calls with Binomial(2, allele-1 frequency of the site), drawing
    from the global legacy NumPy stream once per missing call in variant-major order."""
    dc = count_alleles(gt, max(int(gt.max()), 1))[:, 1]
    ac = to_allele_counts_1(gt)
    missingness = is_missing(gt)
    ninds = (~missingness).sum(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        af = dc / (2 * ninds)
    for i, j in np.argwhere(missingness):
        ac[i, j] = rng.binomial(2, af[i])
    return ac


HOST_THREADS = 8      # threads of the C passes below (they release the GIL)


def _chunks(n, parts):
    step = max(1, -(-n // max(1, parts)))
    return [(a, min(a + step, n)) for a in range(0, n, step)]


def _filter_snps_native(gt, min_mac):
    """The biallelic + allele-1-count filters and to_allele_counts()[:, :, 1] of filter_snps as two C passes over the calls
    (csrc/codecs.c: loc_snp_flags, loc_snp_allele_counts), a few threads each: (ac int8 [K][N], keep flags)."""
    from concurrent.futures import ThreadPoolExecutor
    lib = _codecs()
    gt = np.ascontiguousarray(gt, dtype=np.int8)
    V, N, P = gt.shape
    keep = np.zeros(V, np.uint8)
    parts = _chunks(V, HOST_THREADS * 4)
    with ThreadPoolExecutor(HOST_THREADS) as ex:
        list(ex.map(lambda ab: lib.loc_snp_flags(gt.ctypes.data, ab[0], ab[1], N * P, int(min_mac), keep.ctypes.data), parts))
        pos = np.zeros(V, np.int64)
        np.cumsum(keep[:-1], out=pos[1:]) if V > 1 else None
        K = int(keep.sum())
        ac = np.empty((K, N), np.int8)
        list(ex.map(lambda ab: lib.loc_snp_allele_counts(gt.ctypes.data, ab[0], ab[1], N, P, keep.ctypes.data,
                                                         pos.ctypes.data, ac.ctypes.data), parts))
    return ac, keep


def rows_transposed(ac, rows):
    """`ac[:, rows].T` (split_train_test, locator.py:303-306) as a C blocked transpose on a few threads: the sample-major
    int8 matrix [len(rows)][K] the training path uploads."""
    from concurrent.futures import ThreadPoolExecutor
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    K, N = ac.shape
    out = np.empty((len(rows), K), ac.dtype)
    if len(rows) == 0 or K == 0:
        return out
    if ac.dtype.itemsize != 1 or not ac.flags.c_contiguous:
        return np.ascontiguousarray(ac[:, rows].T)
    lib = _codecs()
    step = (max(64, -(-K // (HOST_THREADS * 4))) + 63) // 64 * 64          # 64-SNP blocks: a write run is one cache line
    parts = [(a, min(a + step, K)) for a in range(0, K, step)]
    with ThreadPoolExecutor(HOST_THREADS) as ex:
        list(ex.map(lambda ab: lib.loc_rows_transposed(ac.ctypes.data, ab[0], ab[1], N, rows.ctypes.data, len(rows),
                                                       out.ctypes.data, K), parts))
    return out


def filter_snps(gt, min_mac=2, max_snps=None, impute_missing=False, rng=np.random, verbose=True, native=True):
    """locator.py:265-281: biallelic sites -> allele-1 count >= min_mac (skipped when min_mac == 1) ->
    allele-1 count matrix (sites x samples) -> optional random subset of max_SNPs sites.
    Quirks kept: the count filtered on is allele 1's, not the minor allele's; monomorphic
    sites never pass (Q8).  native: the two filters and the allele counts as C passes (same integers; the NumPy spelling
    below stays as their restatement and as the --impute_missing path, whose draws follow the filtered array)."""
    if verbose:
        print("filtering SNPs")
    if native and not impute_missing and np.ndim(gt) == 3 and gt.shape[0] > 0 and int(np.min(gt.shape)) > 0:
        ac, _ = _filter_snps_native(gt, min_mac)
        if max_snps is not None:
            ac = ac[rng.choice(range(ac.shape[0]), max_snps, replace=False), :]
        if verbose:
            print("running on " + str(len(ac)) + " genotypes after filtering\n\n\n")
        return ac
    tmp = count_alleles(gt)
    gt = gt[is_biallelic(tmp)]
    if not min_mac == 1:
        derived = count_alleles(gt, max(int(gt.max()) if gt.size else 1, 1))[:, 1]
        gt = gt[derived >= min_mac]
    ac = replace_md(gt, rng) if impute_missing else to_allele_counts_1(gt)
    if max_snps is not None:
        ac = ac[rng.choice(range(ac.shape[0]), max_snps, replace=False), :]
    if verbose:
        print("running on " + str(len(ac)) + " genotypes after filtering\n\n\n")
    return ac
