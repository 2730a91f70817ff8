/* Host-side chunk codecs for the zarr reader (locator_amd/genotypes.py): the Blosc-1 container with LZ4 (and zlib)
 * payloads and byte-shuffle.  The reference reads zarr stores through zarr + numcodecs (locator.py:188-193,
 * scripts/vcf_to_zarr.py:12: `allel.vcf_to_zarr`, whose default compressor is Blosc(cname="lz4", clevel=5,
 * shuffle=1)); neither library is available here, so the published formats are restated:
 *   Blosc-1 chunk: 16-byte header {version, versionlz, flags, typesize, nbytes u32, blocksize u32, cbytes u32}
 *     flags: 0x01 byte-shuffle, 0x02 memcpy'ed, 0x04 bit-shuffle, 0x10 do-not-split, bits 5-7 codec
 *     (0 blosclz, 1 lz4/lz4hc, 2 snappy, 3 zlib, 4 zstd - read through the system's libzstd when present); then, unless memcpy'ed, nblocks int32 block offsets; every
 *     block is 1 stream or (split) `typesize` streams, each {int32 csize, data}; csize == stream size means stored.
 *   LZ4 block: sequences of {token, [literal length bytes], literals, offset u16, [match length bytes]}.
 * Pinned against the reference's own Blosc/LZ4 chunks (locator_py/map.zarr, tests/golden/blosc_*): tests/test_host.py.
 * Plain C, no device code; built into locator_amd/libloc_codecs.so by the same Makefile. */
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <zlib.h>

/* zstd payloads (Blosc codec 4, and the numcodecs "zstd" compressor): common in published Ag1000G-style stores.  The
 * system's libzstd.so.1 is bound at first use with dlopen - the build has no zstd headers and the codecs library must
 * still load where it is absent (-5 = not available, with a message on the Python side). */
typedef size_t (*zstd_decompress_t)(void*, size_t, const void*, size_t);
typedef size_t (*zstd_compress_t)(void*, size_t, const void*, size_t, int);
typedef unsigned (*zstd_iserror_t)(size_t);
static zstd_decompress_t z_dec;
static zstd_compress_t z_cmp;
static zstd_iserror_t z_err;
static int z_state; /* 0 untried, 1 bound, -1 absent */
static int zstd_bind(void) {
    if (z_state == 0) {
        void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
        if (h) {
            z_dec = (zstd_decompress_t)dlsym(h, "ZSTD_decompress");
            z_cmp = (zstd_compress_t)dlsym(h, "ZSTD_compress");
            z_err = (zstd_iserror_t)dlsym(h, "ZSTD_isError");
        }
        z_state = (z_dec && z_cmp && z_err) ? 1 : -1;
    }
    return z_state == 1;
}
int loc_zstd_available(void) { return zstd_bind(); }
/* one zstd frame -> dst; bytes written, -1 malformed / too small, -5 no libzstd */
int64_t loc_zstd_decompress(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap) {
    if (!zstd_bind()) return -5;
    const size_t n = z_dec(dst, (size_t)dst_cap, src, (size_t)src_len);
    return z_err(n) ? -1 : (int64_t)n;
}
/* writer side, for test / synthetic stores only */
int64_t loc_zstd_compress(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap, int level) {
    if (!zstd_bind()) return -5;
    const size_t n = z_cmp(dst, (size_t)dst_cap, src, (size_t)src_len, level);
    return z_err(n) ? -1 : (int64_t)n;
}

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* returns the number of bytes written to dst, or -1 on malformed input / overflow */
int64_t loc_lz4_decompress(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap) {
    const uint8_t *ip = src, *iend = src + src_len;
    uint8_t *op = dst, *oend = dst + dst_cap;
    while (ip < iend) {
        const unsigned token = *ip++;
        int64_t ll = token >> 4;
        if (ll == 15) {
            unsigned b;
            do { if (ip >= iend) return -1; b = *ip++; ll += b; } while (b == 255);
        }
        if (ll > iend - ip || ll > oend - op) return -1;
        memcpy(op, ip, (size_t)ll);
        ip += ll; op += ll;
        if (ip >= iend) break;                    /* the last sequence is literals only */
        if (iend - ip < 2) return -1;
        const int64_t off = ip[0] | (ip[1] << 8);
        ip += 2;
        if (off == 0 || off > op - dst) return -1;
        int64_t ml = (token & 15) + 4;
        if ((token & 15) == 15) {
            unsigned b;
            do { if (ip >= iend) return -1; b = *ip++; ml += b; } while (b == 255);
        }
        if (ml > oend - op) return -1;
        const uint8_t* m = op - off;
        if (off >= ml) { memcpy(op, m, (size_t)ml); op += ml; }
        else { for (int64_t i = 0; i < ml; ++i) op[i] = m[i]; op += ml; }      /* overlapping run */
    }
    return op - dst;
}

static int64_t inflate_to(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap) {
    uLongf n = (uLongf)dst_cap;
    if (uncompress(dst, &n, src, (uLong)src_len) != Z_OK) return -1;
    return (int64_t)n;
}

/* Decode one Blosc-1 chunk.  split_mode: 0 = follow the header, 1 = force split streams, 2 = force unsplit (older
 * writers decide the split without recording it; the caller retries).  tmp: at least `blocksize` bytes (see
 * loc_blosc1_info).  Returns the decompressed byte count (== header nbytes) or a negative error:
 *   -1 malformed, -2 unsupported codec (blosclz, snappy), -3 unsupported filter (bit-shuffle), -4 output buffer too
 *   small, -5 zstd payload but no libzstd on this machine. */
int64_t loc_blosc1_decompress(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap, uint8_t* tmp,
                              int split_mode) {
    if (src_len < 16) return -1;
    const unsigned flags = src[2];
    const int64_t typesize = src[3] ? src[3] : 1;
    const int64_t nbytes = rd32(src + 4), blocksize = rd32(src + 8), cbytes = rd32(src + 12);
    if (cbytes > src_len || nbytes > dst_cap) return nbytes > dst_cap ? -4 : -1;
    if (nbytes == 0) return 0;
    if (flags & 0x02) {                                                   /* stored */
        if (16 + nbytes > src_len) return -1;
        memcpy(dst, src + 16, (size_t)nbytes);
        return nbytes;
    }
    if (flags & 0x04) return -3;
    const unsigned codec = flags >> 5;
    if (codec != 1 && codec != 3 && codec != 4) return -2;
    if (codec == 4 && !zstd_bind()) return -5;
    if (blocksize <= 0) return -1;
    const int shuffle = (flags & 0x01) && typesize > 1;
    const int64_t nblocks = (nbytes + blocksize - 1) / blocksize;
    if (16 + 4 * nblocks > src_len) return -1;
    for (int64_t b = 0; b < nblocks; ++b) {
        const int64_t bsize = (b == nblocks - 1 && nbytes % blocksize) ? nbytes % blocksize : blocksize;
        const int leftover = bsize != blocksize;
        int split = typesize > 1 && typesize <= 16 && (blocksize / typesize) >= 128 && !leftover && !(flags & 0x10);
        if (split_mode == 1) split = typesize > 1 && !leftover && bsize % typesize == 0;
        if (split_mode == 2) split = 0;
        const int64_t nstreams = split ? typesize : 1, ssize = bsize / nstreams;
        int64_t pos = (int64_t)(int32_t)rd32(src + 16 + 4 * b);
        uint8_t* out = shuffle ? tmp : dst + b * blocksize;
        for (int64_t s = 0; s < nstreams; ++s) {
            if (pos < 0 || pos + 4 > src_len) return -1;
            const int64_t cs = (int64_t)(int32_t)rd32(src + pos);
            pos += 4;
            if (cs < 0 || pos + cs > src_len) return -1;
            if (cs == ssize) memcpy(out + s * ssize, src + pos, (size_t)ssize);
            else {
                const int64_t got = codec == 1   ? loc_lz4_decompress(src + pos, cs, out + s * ssize, ssize)
                                    : codec == 3 ? inflate_to(src + pos, cs, out + s * ssize, ssize)
                                                 : loc_zstd_decompress(src + pos, cs, out + s * ssize, ssize);
                if (got != ssize) return -1;
            }
            pos += cs;
        }
        if (shuffle) {                 /* byte j of element i sits at tmp[j*nelem + i]; trailing bytes are copied */
            uint8_t* o = dst + b * blocksize;
            const int64_t nelem = bsize / typesize;
            for (int64_t j = 0; j < typesize; ++j) {
                const uint8_t* in = tmp + j * nelem;
                for (int64_t i = 0; i < nelem; ++i) o[i * typesize + j] = in[i];
            }
            memcpy(o + nelem * typesize, tmp + nelem * typesize, (size_t)(bsize - nelem * typesize));
        }
    }
    return nbytes;
}

/* header fields without decoding: out[0..4] = {nbytes, blocksize, cbytes, typesize, flags}; 0 ok, -1 too short */
int loc_blosc1_info(const uint8_t* src, int64_t src_len, int64_t* out) {
    if (src_len < 16) return -1;
    out[0] = rd32(src + 4); out[1] = rd32(src + 8); out[2] = rd32(src + 12); out[3] = src[3]; out[4] = src[2];
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * filter_snps + split transposes on the host (reference: locator.py:265-273, :295-308; scikit-allel's count_alleles /
 * is_biallelic / to_allele_counts there).  The NumPy spelling of these made the parent's prologue of a --bootstrap run
 * its serial part (15 s of a 669 s job on one GPU, but a tenth of the wall on eight): boolean temporaries over
 * 560,000 x 1000 x 2 calls and strided ploidy reads.  One pass per function over the calls; Python drives them over
 * variant chunks from a few threads (ctypes releases the GIL).  Integer work: bit-identical to the NumPy path.
 * --------------------------------------------------------------------------------------------------------------- */
/* keep[v] = 1 iff exactly two distinct alleles 0..127 occur among the n_calls = N * ploidy calls of variant v (negative =
 * missing, ignored) and (min_mac == 1 or allele 1 occurs at least min_mac times).  Returns the number kept in [v0, v1). */
int64_t loc_snp_flags(const int8_t* gt, int64_t v0, int64_t v1, int64_t n_calls, int min_mac, uint8_t* keep) {
    int64_t kept = 0;
    for (int64_t v = v0; v < v1; ++v) {
        const int8_t* row = gt + v * n_calls;
        uint64_t m0 = 0, m1 = 0;
        int64_t c1 = 0;
        for (int64_t i = 0; i < n_calls; ++i) {
            const int a = row[i];
            if (a >= 0) {
                if (a < 64) m0 |= (uint64_t)1 << a; else m1 |= (uint64_t)1 << (a - 64);
                c1 += a == 1;
            }
        }
        const int distinct = __builtin_popcountll(m0) + __builtin_popcountll(m1);
        const int k = distinct == 2 && (min_mac == 1 || c1 >= min_mac);
        keep[v] = (uint8_t)k;
        kept += k;
    }
    return kept;
}

/* ac[pos[v]][s] = number of allele-1 copies of sample s at kept variant v (to_allele_counts()[:, :, 1]), v in [v0, v1);
 * pos = exclusive prefix sum of keep; ac rows are n_samples bytes. */
void loc_snp_allele_counts(const int8_t* gt, int64_t v0, int64_t v1, int64_t n_samples, int ploidy, const uint8_t* keep,
                           const int64_t* pos, int8_t* ac) {
    for (int64_t v = v0; v < v1; ++v) {
        if (!keep[v]) continue;
        const int8_t* row = gt + v * n_samples * ploidy;
        int8_t* out = ac + pos[v] * n_samples;
        if (ploidy == 2) {
            for (int64_t s = 0; s < n_samples; ++s) out[s] = (int8_t)((row[2 * s] == 1) + (row[2 * s + 1] == 1));
        } else {
            for (int64_t s = 0; s < n_samples; ++s) {
                int c = 0;
                for (int p = 0; p < ploidy; ++p) c += row[s * ploidy + p] == 1;
                out[s] = (int8_t)c;
            }
        }
    }
}

/* out[r][k] = ac[k][rows[r]] for k in [k0, k1): the sample-major matrices of split_train_test (`ac[:, rows].T`), 64 SNPs at a
 * time so that the strided reads of a block stay in L1 and every write run is one cache line.  out pitch = out_pitch bytes. */
void loc_rows_transposed(const int8_t* ac, int64_t k0, int64_t k1, int64_t n_samples, const int64_t* rows, int64_t n_rows,
                         int8_t* out, int64_t out_pitch) {
    for (int64_t kb = k0; kb < k1; kb += 64) {
        const int64_t ke = kb + 64 < k1 ? kb + 64 : k1;
        for (int64_t r = 0; r < n_rows; ++r) {
            const int8_t* src = ac + rows[r];
            int8_t* dst = out + r * out_pitch;
            for (int64_t k = kb; k < ke; ++k) dst[k] = src[k * n_samples];
        }
    }
}
