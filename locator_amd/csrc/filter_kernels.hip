// filter_snps + the split's transposes on the device, for the --windows replicate loop
// (reference: /root/reference/locator/locator.py:265-273 filter_snps without --impute_missing / --max_SNPs, :295-308
// split_train_test, called per window at :539-545).  The reference does this with scikit-allel on the host per window:
// count_alleles -> is_biallelic -> allele-1 count >= min_mac -> to_allele_counts()[:, :, 1] -> ac[:, rows].T.  Integer work
// over k_w x N x 2 int8 calls (230 MB for a 150,000-variant window of 765 samples): a streaming pass for the per-variant
// flags, a prefix sum, and one compaction + transpose pass that writes the sample-major uint8 rows the training kernels
// read, rows in train | validation | prediction order.  Results are bit-identical to
// genotypes.filter_snps + the NumPy transposes (tests/test_gpu_filter.py).
#include "common.h"

// One wave per variant: which alleles 0..127 occur among the 2 N calls (negative = missing, ignored - allel's
// count_alleles), and how often allele 1 does.  keep = exactly two distinct alleles (is_biallelic of the counts over
// 0..max allele) and, unless min_mac == 1 (the reference skips the second filter then, locator.py:270), allele-1 count >= min_mac.
__global__ __launch_bounds__(256) void snp_flags_kernel(const int8_t* __restrict__ gt, int64_t n_variants, int row_bytes,
                                                        int min_mac, uint8_t* __restrict__ keep) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_variants) return;
    const int8_t* row = gt + v * row_bytes;
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0, c1 = 0;
    for (int i = lane; i < row_bytes; i += 64) {
        const int a = row[i];
        if (a >= 0) {
            const uint32_t bit = 1u << (a & 31);
            const int word = a >> 5;
            m0 |= word == 0 ? bit : 0u; m1 |= word == 1 ? bit : 0u; m2 |= word == 2 ? bit : 0u; m3 |= word == 3 ? bit : 0u;
            c1 += a == 1 ? 1u : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m0 |= (uint32_t)__shfl_xor((int)m0, o); m1 |= (uint32_t)__shfl_xor((int)m1, o);
        m2 |= (uint32_t)__shfl_xor((int)m2, o); m3 |= (uint32_t)__shfl_xor((int)m3, o);
        c1 += (uint32_t)__shfl_xor((int)c1, o);
    }
    if (lane == 0) {
        const int distinct = __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3);
        keep[v] = (distinct == 2 && (min_mac == 1 || (int)c1 >= min_mac)) ? 1 : 0;
    }
}

// Exclusive prefix sum of the keep flags by ONE workgroup (a window has a few hundred thousand variants): every thread
// sums a contiguous chunk, the 1024 chunk sums are scanned through LDS, every thread writes its chunk's positions.
__global__ __launch_bounds__(1024) void snp_scan_kernel(const uint8_t* __restrict__ keep, int64_t n, int32_t* __restrict__ pos,
                                                        int32_t* __restrict__ n_kept) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int64_t chunk = (n + 1023) / 1024, a = t * chunk, b = a + chunk < n ? a + chunk : n;
    int32_t s = 0;
    for (int64_t i = a; i < b; ++i) s += keep[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int32_t add = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int32_t run = part[t] - s;                  // exclusive prefix of this chunk
    for (int64_t i = a; i < b; ++i) {
        pos[i] = run;
        run += keep[i];
    }
    if (t == 1023) *n_kept = part[1023];
}

// A workgroup takes FT consecutive variants: the allele-1 count of every (kept variant, sample) goes into an LDS tile
// indexed [kept position inside the tile][sample], then every wave writes output rows: lane = kept variant, so a row's
// bytes of this tile are one contiguous run X[r][pos0 .. pos0 + cnt).  Samples are taken in chunks of FS so that any N
// fits the LDS.
#define FT 64
#define FS 960
#define FPAD 4      /* row pitch FS + 4 bytes = 241 words (odd): lanes (variants) reading one sample fall into distinct banks; 61.7 KB */
__global__ __launch_bounds__(256) void snp_rows_kernel(const int8_t* __restrict__ gt, int64_t n_variants, int n_samples, int ploidy,
                                                       const uint8_t* __restrict__ keep, const int32_t* __restrict__ pos,
                                                       const int32_t* __restrict__ sample_order, int n_out,
                                                       uint8_t* __restrict__ X, int64_t x_pitch) {
    __shared__ uint8_t tile[FT][FS + FPAD];
    __shared__ int32_t slot[FT];                // kept position inside the tile, or -1
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int64_t v0 = (int64_t)blockIdx.x * FT;
    const int nv = n_variants - v0 < FT ? (int)(n_variants - v0) : FT;
    const int32_t pos0 = pos[v0];
    if (t < FT) slot[t] = (t < nv && keep[v0 + t]) ? pos[v0 + t] - pos0 : -1;
    __syncthreads();
    int cnt = 0;
    {
        const int last = nv - 1;
        cnt = pos[v0 + last] - pos0 + keep[v0 + last];
    }
    if (cnt == 0) return;
    const int64_t row_bytes = (int64_t)n_samples * ploidy;
    for (int s0 = 0; s0 < n_samples; s0 += FS) {
        const int ns = n_samples - s0 < FS ? n_samples - s0 : FS;
        __syncthreads();
        for (int j = 0; j < nv; ++j) {
            const int sl = slot[j];
            if (sl < 0) continue;
            const int8_t* row = gt + (v0 + j) * row_bytes + (int64_t)s0 * ploidy;
            for (int s = t; s < ns; s += 256) {
                int c = 0;
                for (int p = 0; p < ploidy; ++p) c += row[s * ploidy + p] == 1 ? 1 : 0;
                tile[sl][s] = (uint8_t)c;
            }
        }
        __syncthreads();
        // rows whose sample lies in this chunk: wave w takes rows w, w + 4, ...
        for (int r = w; r < n_out; r += 4) {
            const int s = sample_order[r] - s0;
            if (s < 0 || s >= ns) continue;
            if (lane < cnt) X[(int64_t)r * x_pitch + pos0 + lane] = tile[lane][s];
        }
    }
}

extern "C" int loc_filter_snps_flags(const int8_t* gt, int64_t n_variants, int n_samples, int ploidy, int min_mac,
                                     uint8_t* keep, int32_t* pos, int32_t* n_kept, void* stream) {
    if (n_variants < 1 || n_samples < 1 || ploidy < 1 || (int64_t)n_samples * ploidy > (1 << 30) || n_variants > ((int64_t)1 << 31) - 1024) {
        loc_set_error("loc_filter_snps_flags: n_variants=%lld n_samples=%d ploidy=%d", (long long)n_variants, n_samples, ploidy);
        return -1;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(snp_flags_kernel, dim3((unsigned)((n_variants + 3) / 4)), dim3(256), 0, st, gt, n_variants,
                       n_samples * ploidy, min_mac, keep);
    LOC_CHECK_LAUNCH();
    hipLaunchKernelGGL(snp_scan_kernel, dim3(1), dim3(1024), 0, st, keep, n_variants, pos, n_kept);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_filter_snps_rows(const int8_t* gt, int64_t n_variants, int n_samples, int ploidy, const uint8_t* keep,
                                    const int32_t* pos, const int32_t* sample_order, int n_out, uint8_t* X, int64_t x_pitch,
                                    void* stream) {
    if (n_variants < 1 || n_samples < 1 || ploidy < 1 || n_out < 0) {
        loc_set_error("loc_filter_snps_rows: n_variants=%lld n_samples=%d ploidy=%d n_out=%d", (long long)n_variants, n_samples,
                      ploidy, n_out);
        return -1;
    }
    if (n_out == 0) return 0;
    hipLaunchKernelGGL(snp_rows_kernel, dim3((unsigned)((n_variants + FT - 1) / FT)), dim3(256), 0, (hipStream_t)stream, gt,
                       n_variants, n_samples, ploidy, keep, pos, sample_order, n_out, X, x_pitch);
    LOC_CHECK_LAUNCH();
    return 0;
}
