// Large-M first-layer genotype GEMM on the INT8 matrix pipe (reference: model.predict on predgen / testgen and the
// --jacknife replicate predictions, /root/reference/locator/locator.py:414, :441, :683-747).
//
//     z1[m][h] = sum_k x[m][k] (s_k W1[k][h])  +  sum_k t_k W1[k][h]          (BatchNorm in inference form)
//
// Why int8.  tools/probes/mfma_clock_probe.hip (profiles/r03_mfma_clock_probe.jsonl) measured what the matrix pipe
// sustains on genotype-like operands once the chip has settled at its power budget: bf16 32x32x16 1,969 TFLOP/s at
// 1.97 GHz (0.79 of the 2.5 PF dense peak), f16 1,839 at 1.83 GHz (0.74), i8 32x32x32 3,557 TOP/s at 1.78 GHz (1.42).
// A genotype (0, 1, 2; anything up to 127) IS an int8, so the A operand needs no widening at all - the u8 rows go
// global -> register -> LDS -> MFMA untouched - and the weight w' = s_k W1[k][h] is carried as base-256 signed digits
// of the fixed-point number q = rint(w' / delta_h), delta_h a power of two chosen per unit from max_k |w'| (a
// streaming max pass over W1S).  Products and sums are exact integers (i32 accumulation):
//     3 digits   24-bit fixed point against the unit's largest weight: |error| <= delta_h / 2 = max_k|w'| 2^-24 per
//                weight, i.e. no more than the rounding of an fp32 accumulation ("exact" mode; 1.5 bf16-MFMA
//                equivalents per product where the exactly-split bf16 form needs 3 and an fp16 split 2)
//     2 digits   16-bit fixed point: max_k|w'| 2^-16 per weight ("fast" mode; 1 bf16-MFMA equivalent)
// Genotypes above 127 and K ranges long enough to overflow an i32 (x_max * 128 * SNPs per group >= 2^31) are refused
// here; loc_predict then takes the bf16 kernels of l1_gemm.hip / l1_rows.hip, which are exact for any uint8.
//
//   l1_scan_kernel       per-unit max_k |s_k W1[k][h]| (order-independent: max over non-negative floats as uints), and the
//                        per-workgroup shares of sum_k |s_k W1[k][h]| (guard) and of the shift term sum_k t_k W1[k][h]
//   l1_image_i8_kernel   W1S (fp32, swizzled) x BN scale -> HBM image of digit planes, one 16 KB tile per (64-SNP block,
//                        digit plane), laid out [16-SNP chunk c][unit n][16 SNPs]: a wave's MFMA B operand for one
//                        32-SNP step is two contiguous 512-byte runs.  Plane 0 is the most significant digit.  One extra
//                        workgroup (g8_guard_body) turns the scan's shares into the guard and the shift vector meanwhile.
//   l1_gemm_i8_kernel    workgroup = 8 waves on a 128-row x 256-unit tile; the SNP range is split over groups, group g
//                        owning a CONTIGUOUS run of pairs of 64-SNP blocks.  A wave owns 32 units and ALL 128 rows: its
//                        digit fragments go HBM/L2 -> VGPRs (12 fragments = 1.5 pairs in flight, saddr-form requests),
//                        only the raw genotype rows are shared: LDS-DMA into a ring of pairs, [row][128 B] with the
//                        eight 16-byte pieces of a row XOR-placed so that the lane-linear DMA and the 16-lane read
//                        groups are bank-conflict free.  All eight waves run one software-pipelined stream and meet
//                        once per pair (g8_sweep below); there is NO vector-ALU work in the loop.  Two digits
//                        accumulate side by side (2 x 64 accumulator registers); the exact mode stores their fp32
//                        fold and walks the K range a second time for the least significant plane (genotypes then
//                        L2-hot, 8 MFMAs per block), adding it in place.
//                        History (git log): the first version was l1_gemm_kernel's half-block
//                        ping-pong with the widening removed - 215 us at 4096 rows x 2 digits, the same cycle count as
//                        bf16 x 1; this schedule with scalar-base requests 203, DMA lookahead 2 pairs 194.
//   reduction            l1_gemm_reduce_kernel of l1_gemm.hip (fixed-order sum of the group partials + shift + b1, ELU).
#include "common.h"
#include <type_traits>

typedef int32_t i32x16 __attribute__((ext_vector_type(16)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

#define G8_BM 128
#define G8_BK 64
#define G8_NT 512
#define G8_HP 256
#define G8_TILE (G8_HP * G8_BK)   /* 16384: one (SNP block, digit plane) tile, int8 */
#define G8_AIMG (G8_BM * G8_BK)   /* 8192: raw genotype image of one SNP block     */
#define G8_LDS 131072             /* 4 pairs of genotype rows (64 KB) in the loop; the epilogue stages 128 KB of partials */
#define G8_WAVE_UNIT_TILES 1      /* default unit tiles per wave: 1 = 8 waves x 256 registers, 2 = 4 waves x 512 registers
                                     (loc_tuning.gemm_i8_unit_tiles overrides) */


// ---------------------------------------------------------------------------------------------------------
// per-unit scale and digit image
// ---------------------------------------------------------------------------------------------------------
// Workgroup b walks the 64-SNP blocks b, b + grid, ...; one atomicMax per unit and workgroup at the end (a few hundred
// workgroups, not one per block: the atomics all hit the same 256 words).  colmax[n] = max |s_k W1[k][n]| as the bit pattern
// of a non-negative float: the maximum is order-independent, so the result is deterministic.
// The same pass also leaves two shares per workgroup, added up in workgroup order by g8_guard_body (the image
// kernel's extra workgroup, or l1_quant_guard_kernel):
//   sumabs_part[workgroup][n]  its share of sum_k |w'|: the unit's TYPICAL weight, against which its largest weight decides
//                              how many digit planes the weights need (dynamic range guard below).  The mean magnitude, not
//                              the rms: one weight 3000 x the rest moves the rms of 100,000 weights by 10 x (max / rms can
//                              never exceed sqrt(K), and the first version of this guard, built on the rms, let exactly that
//                              case through) but the mean by 3 %.
//   csum_part[workgroup][n]    its share of the shift term sum_k t_k W1[k][n] (round 5: it used to be a by-product of the
//                              image kernel plus a launch of l1_image_cvec_kernel; here it costs one FMA per weight)
// Thread = unit n (the addressing of l1_image_kernel): four-byte loads, the BatchNorm scale / shift of a SNP are wave-uniform
// (scalar loads).  A 16-byte-per-lane form (a quad's lanes taking the four units of one SNP each, statistics exchanged at
// the end) was built in round 5 and measured SLOWER: 30.4 us against 19.4 for the same 102 MB - its scale / shift loads are
// lane-dependent vector loads, 16 more per 32-SNP tile.
__global__ __launch_bounds__(G8_HP) void l1_scan_kernel(const float* __restrict__ w1s, const float* __restrict__ ss4,
                                                        int Kp, int nkt64, uint32_t* __restrict__ colmax,
                                                        float* __restrict__ sumabs_part, float* __restrict__ csum_part) {
    constexpr int nht = G8_HP / 32;
    const int n = threadIdx.x;
    const int ht = n >> 5, hl = n & 31, q = hl >> 3, hi = (hl >> 2) & 1, c4 = hl & 3;
    float mx = 0.f, ss = 0.f, cs = 0.f;
    for (int kt64 = blockIdx.x; kt64 < nkt64; kt64 += gridDim.x) {
#pragma unroll
        for (int h32 = 0; h32 < 2; ++h32) {
            const int kt32 = 2 * kt64 + h32;
            if (kt32 * 32 < Kp) {
                const float* src = w1s + ((int64_t)(kt32 * nht + ht) * 4 + q) * 256 + hi * 128 + c4;
                float m4[4] = {0.f, 0.f, 0.f, 0.f}, s4[4] = {0.f, 0.f, 0.f, 0.f}, c4s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kl = 0; kl < 32; ++kl) {
                    const float wr = src[kl * 4];
                    const float w = wr * ss4[kt32 * 32 + kl];
                    m4[kl & 3] = fmaxf(m4[kl & 3], fabsf(w));
                    s4[kl & 3] += fabsf(w);
                    c4s[kl & 3] = fmaf(ss4[Kp + kt32 * 32 + kl], wr, c4s[kl & 3]);
                }
                mx = fmaxf(mx, fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])));
                ss += (s4[0] + s4[1]) + (s4[2] + s4[3]);
                cs += (c4s[0] + c4s[1]) + (c4s[2] + c4s[3]);
            }
        }
    }
    if (mx > 0.f) atomicMax(colmax + n, fbits(mx));
    sumabs_part[(int64_t)blockIdx.x * G8_HP + n] = ss;
    csum_part[(int64_t)blockIdx.x * G8_HP + n] = cs;
}

// Dynamic-range guard of the fixed-point image.  A unit's weights share one power-of-two step delta_h chosen from its
// LARGEST |w'|, so what a digit count leaves for a typical weight is set by R_h = max_k |w'| / (1.2533 mean_k |w'|) - the
// largest weight over the rms a Gaussian bulk of that mean magnitude would have: a typical weight keeps
// log2(2^(8 D - 1) / R_h) bits and its quantisation noise is R_h 2^-(8 D - 1) / sqrt(12) of its size.  After training the
// ratio is a few tens (rare SNPs carry BatchNorm scales up to 30 x the common ones, Adam grows informative rows;
// tools/quant_study.py, tests/test_gpu_trained_predict.py), which two digits carry with a measured deviation of 5e-5 on the
// predictions; a weight 1000 x its unit's typical size takes two digits to 2.5e-3 and three to 1e-5.
// guard[0] = median R over the real units (lower middle value), guard[1] = largest R, guard[2] = digit planes that hold
// the tolerances of include/locator_hip.h (LOC_GUARD_*): 2, 3, or -1 = not even three (bf16 x 3 pieces: exact for any
// weights), guard[3] = 3 or -1: the same decision when the caller insists on the exact mode.
// Stage 1 of the guard: share group q (of 16) = workgroup shares b = q, q + 16, ..., q + 240, added by the tree
// ((e, e + 8), (e, e + 4), (e, e + 2), (e, e + 1)) -> one row of 256 values.  Sixteen 16-byte loads per thread at once: the
// shares were written by workgroups on every XCD, so each load is a trip to memory, and a compute unit retires a
// 4-byte-per-lane load instruction no faster than a 16-byte one (1,024 of them took 10 of the guard kernel's 15 us; a single
// wave per unit walking 512 strided shares one after the other: 127 us; a 256-thread workgroup taking the sixteen groups in
// four rounds: 35 us - each round pays the full memory latency).
// 64 threads per group (thread = four units): returns the group's sums of the four units 4 n4 .. 4 n4 + 3.
__device__ __forceinline__ f32x4 g8_share_group16(const float* __restrict__ part, int q, int n4, int nparts) {
    f32x4 a[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int b = q + 16 * e;
        a[e] = b < nparts ? *reinterpret_cast<const f32x4*>(part + (int64_t)b * G8_HP + 4 * n4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int w2 = 8; w2 > 0; w2 >>= 1)
#pragma unroll
        for (int e = 0; e < w2; ++e) a[e] = a[e] + a[e + w2];
    return a[0];
}
// The same sums by 256 threads for ONE group: quarter `sub` (= wave) of the workgroup loads e = sub, sub + 4, sub + 8, sub + 12
// and forms (a[sub] + a[sub + 8]) + (a[sub + 4] + a[sub + 12]) - what the tree above holds in a[sub] after its first two
// levels - and the four quarters meet in LDS as (x0 + x2) + (x1 + x3): the same bits.
__device__ __forceinline__ f32x4 g8_share_group4(const float* __restrict__ part, int q, int n4, int sub, int nparts) {
    f32x4 a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = q + 16 * (sub + 4 * i);
        a[i] = b < nparts ? *reinterpret_cast<const f32x4*>(part + (int64_t)b * G8_HP + 4 * n4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return (a[0] + a[2]) + (a[1] + a[3]);
}

// Stage 2: the sixteen group rows qs[16][256] (LDS or global) -> guard[0..3]; qc non-NULL: cvec8[0][n] = the shift term,
// cvec8[1..7][n] = 0 (l1_gemm_reduce_kernel adds eight slices).  NT >= 256 threads, the statistics are the first 256's work.
template <int NT>
__device__ __forceinline__ void g8_guard_finish(const uint32_t* __restrict__ colmax, const float* qs, const float* qc, int K,
                                                int H, float* __restrict__ guard, float* __restrict__ cvec8) {
    __shared__ __attribute__((aligned(16))) float R[G8_HP];
    const int n = threadIdx.x & (G8_HP - 1);
    const bool first = threadIdx.x < G8_HP;
    if (qc && first) {
        float cs = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) cs += qc[q * G8_HP + n];
        cvec8[n] = cs;
#pragma unroll
        for (int sl = 1; sl < 8; ++sl) cvec8[sl * G8_HP + n] = 0.f;
    }
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) ss += qs[q * G8_HP + n];
    const float mx = bitsf(colmax[n]);
    const float typ = 1.2533141f * ss / (float)K;           // sqrt(pi / 2) x mean magnitude = the rms of a Gaussian bulk
    const float r = (n < H && typ > 0.f) ? mx / typ : 0.f;
    if (first) R[n] = r;
    __syncthreads();
    // median by rank counting (256 values): the value with exactly floor((H - 1) / 2) smaller-or-earlier entries.  The
    // 256 x 256 comparisons are most of this part's arithmetic: with 1024 threads every thread takes a quarter of the
    // candidates of its unit (thread (p, n): entries 64 p .. 64 p + 63), the partial counts meet in LDS
    constexpr int NP = NT / G8_HP;
    __shared__ int rk[NP][G8_HP];
    {
        const int p = threadIdx.x >> 8;
        const f32x4* R4 = reinterpret_cast<const f32x4*>(R) + (64 / NP) * p;
        int cnt = 0;
#pragma unroll 4
        for (int j4 = 0; j4 < 64 / NP; ++j4) {
            const f32x4 v = R4[j4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = (256 / NP) * p + 4 * j4 + e;
                cnt += (j < H && (v[e] < r || (v[e] == r && j < n))) ? 1 : 0;
            }
        }
        rk[p][n] = cnt;
    }
    __syncthreads();
    int rank = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) rank += rk[p][n];
    // the largest R: inside each of the first four waves by lane exchanges, then four values through LDS
    __shared__ float red[4];
    float wmax = r;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
    if (first && (threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wmax;
    __shared__ float s_med;
    if (first && n < H && rank == (H - 1) / 2) s_med = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float rmed = s_med, rmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        guard[0] = rmed;
        guard[1] = rmax;
        const float exact = rmax <= LOC_GUARD_EXACT_MAX ? 3.f : -1.f;
        guard[2] = (rmed <= LOC_GUARD_FAST_MEDIAN && rmax <= LOC_GUARD_FAST_MAX) ? 2.f : exact;
        guard[3] = exact;
    }
}

// the guard as a launch of its own (loc_l1_quant_scan: a caller that reads it back before the image is built): 1024 threads,
// thread (q, n4) = share group q, units 4 n4 .. 4 n4 + 3
__global__ __launch_bounds__(1024) void l1_quant_guard_kernel(const uint32_t* __restrict__ colmax,
                                                              const float* __restrict__ sumabs_part, int nparts, int K, int H,
                                                              float* __restrict__ guard) {
    __shared__ __attribute__((aligned(16))) float qs[16 * G8_HP];
    const int n4 = threadIdx.x & 63, q = threadIdx.x >> 6;
    *reinterpret_cast<f32x4*>(&qs[q * G8_HP + 4 * n4]) = g8_share_group16(sumabs_part, q, n4, nparts);
    __syncthreads();
    g8_guard_finish<1024>(colmax, qs, nullptr, K, H, guard, nullptr);
}

// delta_h = 2^e with max_k|w'| / delta_h inside the signed-digit range: 127 (256^DT - 1) / 255.
template <int DT>
__device__ __forceinline__ float digit_delta(float mx) {
    if (!(mx > 0.f)) return 1.f;
    int x;
    const float f = frexpf(mx, &x);                       // mx = f 2^x, f in [0.5, 1)
    int e = x - (8 * DT - 1);                             // mx / 2^e = f 2^(8 DT - 1)
    constexpr float lim = DT == 2 ? 32639.f : 8355711.f;
    if (ldexpf(f, 8 * DT - 1) > lim - 1.f) ++e;
    return ldexpf(1.f, e);
}

// tiles[(kt64*DT + p)][c][n][e] = digit plane p (0 = most significant) of q = rint(s_k W1[k][n] / delta_n),
// k = kt64*64 + c*16 + e;  delta[n] written by the first tile workgroup.  Workgroup G8_TAIL_WGS + kt64 converts 64-SNP block
// kt64 (thread = unit).  The first G8_TAIL_WGS = 16 workgroups do the once-per-image leftovers meanwhile (round 5: they used
// to be two launches, l1_quant_guard_kernel and l1_image_cvec_kernel, 9 + 5 us on the stream): workgroup q adds share group q
// of the scan (sum |w'| and the shift term) into tail_rows[q] / tail_rows[16 + q], publishes it (agent-scope release, then a
// ticket on *tail_ticket), and the one that draws the last ticket turns the sixteen rows into guard[0..3] and cvec8 - the
// same association of every sum as l1_quant_guard_kernel.  Measured on the way: ONE leftover workgroup with 512 four-byte
// loads per thread took longer than all the tiles together (71 us per image instead of 60); the same workgroup with the
// sixteen groups in four rounds of 16-byte loads 35 us (image kernel 38.7 us); tiles by 1024-thread workgroups (so that one
// of them could be the 1024-thread guard): 29.8 us against 25.8.
#define G8_TAIL_WGS 16
template <int DT>
__global__ __launch_bounds__(G8_HP) void l1_image_i8_kernel(const float* __restrict__ w1s, const float* __restrict__ ss4,
                                                            int Kp, int nkt64, int K, int H, const uint32_t* __restrict__ colmax,
                                                            float* __restrict__ delta, unsigned char* __restrict__ tiles,
                                                            const float* __restrict__ sumabs_part,
                                                            const float* __restrict__ csum_part, int nparts,
                                                            float* __restrict__ guard, float* __restrict__ cvec8,
                                                            float* __restrict__ tail_rows, unsigned* __restrict__ tail_ticket) {
    if (blockIdx.x < G8_TAIL_WGS) {
        __shared__ __attribute__((aligned(16))) float xs[2][4][G8_HP];
        __shared__ unsigned last;
        const int q = blockIdx.x, n4 = threadIdx.x & 63, sub = threadIdx.x >> 6;
        *reinterpret_cast<f32x4*>(&xs[0][sub][4 * n4]) = g8_share_group4(sumabs_part, q, n4, sub, nparts);
        *reinterpret_cast<f32x4*>(&xs[1][sub][4 * n4]) = g8_share_group4(csum_part, q, n4, sub, nparts);
        __syncthreads();
        {
            const int n = threadIdx.x;
            tail_rows[q * G8_HP + n] = (xs[0][0][n] + xs[0][2][n]) + (xs[0][1][n] + xs[0][3][n]);
            tail_rows[(16 + q) * G8_HP + n] = (xs[1][0][n] + xs[1][2][n]) + (xs[1][1][n] + xs[1][3][n]);
        }
        // publish, then take a ticket (cdna_hip_programming.md, the in-launch split-K ending: drain, barrier, ONE agent-scope
        // release by one lane, the wait restated after it, relaxed ticket; the last arriver acquires once)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned k = __hip_atomic_fetch_add(tail_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (k % G8_TAIL_WGS == G8_TAIL_WGS - 1) ? 1u : 0u;      // (the count is never reset: sixteen tickets per image)
            if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (last) g8_guard_finish<G8_HP>(colmax, tail_rows, tail_rows + 16 * G8_HP, K, H, guard, cvec8);
        return;
    }
    constexpr int nht = G8_HP / 32;
    const int kt64 = blockIdx.x - G8_TAIL_WGS, n = threadIdx.x;
    const int ht = n >> 5, hl = n & 31, q = hl >> 3, hi = (hl >> 2) & 1, c4 = hl & 3;
    const float* scale = ss4;
    const float dl = digit_delta<DT>(bitsf(colmax[n]));
    const float inv = 1.0f / dl;                          // a power of two: exact
    if (kt64 == 0) delta[n] = dl;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int kt32 = 2 * kt64 + (c >> 1);
        uint32_t pk[DT][4];
#pragma unroll
        for (int p = 0; p < DT; ++p) pk[p][0] = pk[p][1] = pk[p][2] = pk[p][3] = 0u;
        if (kt32 * 32 < Kp) {
            const float* src = w1s + ((int64_t)(kt32 * nht + ht) * 4 + q) * 256 + hi * 128 + c4;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kl = (c & 1) * 16 + e, k = kt32 * 32 + kl;
                const float w = src[kl * 4];
                int qv = (int)rintf((w * scale[k]) * inv);
                // signed base-256 digits, least significant first; the top digit takes what is left
                int dg[DT];
#pragma unroll
                for (int p = DT - 1; p > 0; --p) {
                    const int lo = ((qv + 128) & 255) - 128;
                    dg[p] = lo;
                    qv = (qv - lo) >> 8;
                }
                dg[0] = qv;
#pragma unroll
                for (int p = 0; p < DT; ++p) pk[p][e >> 2] |= (uint32_t)(dg[p] & 255) << (8 * (e & 3));
            }
        }
#pragma unroll
        for (int p = 0; p < DT; ++p) {
            u32x4 v;
            v[0] = pk[p][0]; v[1] = pk[p][1]; v[2] = pk[p][2]; v[3] = pk[p][3];
            *reinterpret_cast<u32x4*>(tiles + ((int64_t)kt64 * DT + p) * G8_TILE + c * (G8_HP * 16) + n * 16) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_addr32_i8(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// the four A fragments (row tiles 0..3) of one 32-SNP step by four 16-byte LDS reads; volatile so that they stay in
// the load phase, ahead of the barrier that hands the matrix pipe to this wave
__device__ __forceinline__ void rd4_i8(i32x4& a0, i32x4& a1, i32x4& a2, i32x4& a3, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\t"
                 "ds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %4 offset:12288"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr) : "memory");
}
// Global loads as asm with hand-counted s_waitcnt vmcnt (see l1_gemm.hip: the compiler's own counts collapse at the
// loop header).  COUNT TABLE, per wave, requests in program order.  One iteration = one PAIR of 64-SNP blocks =
// 4 steps of 32 SNPs; a wave owns UT unit tiles (of 32 units); FP = 4 D UT digit fragments per pair, consumed in the
// order (step, plane, unit tile); NB fragments in flight; DPW = 2 UT genotype DMAs per wave and pair:
//   head of an iteration   DPW genotype DMAs (this wave's 16 UT rows of the pair G8_LA iterations ahead)
//   after each 4 MFMAs     1 fragment request, NB fragments ahead of the one just consumed
//   => fragment j of a pair (j = 0..FP-1) was requested NB fragments ago; younger than it are NB - 1 fragment requests
//      and the DMAs of every iteration head crossed (g8_heads() below):          vmcnt(NB - 1 + DPW heads)
//   => before the rendezvous of an iteration (after steps 0 and 1) the wave's DMAs of the NEXT pair must have landed:
//      they were issued LA - 1 heads ago; younger are (LA - 1) later heads' DMAs, (LA - 1) whole iterations of FP
//      fragment requests and the FP / 2 of steps 0, 1:                 vmcnt((LA - 1)(DPW + FP) + FP / 2)
//   (a LOC_GEMM_DEBUG_DRAIN build replaces every count by vmcnt(0) for parity debugging: make debug_drain)
template <int N>
__device__ __forceinline__ void wait_vm_i8() {
#ifdef LOC_GEMM_DEBUG_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wait_lgkm0_i8() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void barrier_i8() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// iteration heads between the request of fragment j of a pair and its consumption NB fragments later: a fragment
// F = c FP + j is requested right after F - NB is consumed and the head of iteration c' sits just before fragment
// c' FP, so the heads counted are those with F - NB < c' FP <= F:  ceil((NB - j) / FP)
constexpr int g8_heads(int j, int FP, int NB) { return NB > j ? (NB - j + FP - 1) / FP : 0; }

// Timing ablations (never in the product build: `make ablate A=<bits>` writes ../liblocator_hip_ablate<bits>.so, results
// are wrong by construction): 1 no fragment requests in the loop, 2 no genotype DMAs, 4 no rendezvous, 8 no A-fragment
// reads, 16 genotype DMA always from the group's first pair (L2-hot), 64 no global stores of the partial sums (the LDS staging
// stays), 128 no epilogue at all.
#ifndef LOC_GEMM_ABLATE
#define LOC_GEMM_ABLATE 0
#endif
#ifndef G8_LA
#define G8_LA 2       /* pairs between a genotype DMA and the iteration that reads it (measured: 2 beats 1, 3, 4)  */
#endif
#define G8_RP 4       /* pairs in the genotype ring (4 x 16 KB) >= G8_LA + 2                                       */
#ifndef G8_DMA_MOD
#define G8_DMA_MOD "" /* cache-policy modifier of the genotype DMA (" nt", " sc1": measured, within 2 %)            */
#endif

// PACKED genotypes (template flag PK): rows of 2-bit genotypes (values 0..3, four SNPs per byte: loc_pack_genotypes_2bit),
// a quarter of the HBM lines and of the bytes per request.  The DMA moves 4 bytes per lane (8 rows x 32 bytes per request,
// the same number of requests per wave and pair, so every hand count above holds) into a ring of G8_RPK pairs x 4 KB behind
// the epilogue's staging area; it runs ONE pair further ahead, and after the rendezvous of pair pc every thread expands
// two (UT = 2: four) packed words of pair pc + 2 into the int8 ring, which the rendezvous of pair pc + 1 publishes.
#define G8_PKOFF G8_LDS   /* byte offset of the packed ring in the kernel's LDS                                       */
#define G8_RPK 4          /* pairs in the packed ring >= G8_LA + 2                                                      */
#define G8_LDS_PK (G8_LDS + G8_RPK * 4096)

struct g8_ctx {
    const uint8_t* xrow[4];       // this lane's genotype rows (DMA role): rows 16 UT w + 8 i + (lane >> 3), i < 2 UT
    uint32_t xpiece[4];           // 16 x the piece of the pair's 128-byte line this lane fetches for each of them
    const unsigned char* tiles;
    unsigned char* As;
    int p0, cntp, Kp, w, jl, hi, lane;   // this group's pairs of 64-SNP blocks: [p0, p0 + cntp)
};

// 16 bytes per lane global -> LDS without passing a register (LDS-DMA): the wave's 64 lanes fill the 1 KB at `lds_dst`
// (wave-uniform) in lane order, each from its own address.  M0 is the destination base and is compiler-reserved: saved,
// set and restored inside the one statement (guide: cdna_hip_programming.md, "LDS-DMA recipe").
__device__ __forceinline__ void dma16_i8(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" G8_DMA_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void dma4_i8(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" G8_DMA_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// four 2-bit genotypes (one byte, least significant pair first) -> four int8 in one word
__device__ __forceinline__ uint32_t g8_expand4(uint32_t b) {
    uint32_t x = (b | (b << 12)) & 0x000F000Fu;            // genotypes 1,0 in bits 3:0, genotypes 3,2 in bits 19:16
    return (x | (x << 6)) & 0x03030303u;                    // ... each in the low two bits of its own byte
}

// One walk over this workgroup's SNP blocks with D digit planes [plane0, plane0 + D) of the DT planes in the image, by a
// wave that owns UT unit tiles (UT = 1: eight waves per workgroup, two per SIMD, 256 registers each; UT = 2: four waves,
// one per SIMD with the whole 512-register file - twice the accumulators and room for NB = 32 fragments in flight).
// All waves run the same software-pipelined stream and meet once per pair.  Genotypes: at the head of an iteration
// the wave starts the DMA of its rows of the pair G8_LA iterations ahead (8 rows x 128 bytes per request; ring of
// G8_RP pairs x 16 KB, row m at m*128 with its eight 16-byte pieces XOR-placed by (m >> 1) & 7, which keeps both the
// lane-linear DMA stores and the 16-lane read groups on distinct banks) - no register, no LDS store instruction.  Each
// 32-SNP step then waits for the A fragments it prefetched one step earlier, prefetches the next step's four fragments
// (16 registers, double buffered) and issues D x UT x 4 MFMAs, one fragment request after every 4 (saddr form: the pair
// base is a scalar, the lane part and the in-pair constant sit in FP VGPRs - no address arithmetic in the loop).  The
// rendezvous sits in the middle of the pair - before it the wave waits for its own share of the NEXT pair's DMA - so
// the first step of the next pair is prefetched half a pair ahead of its use and nobody waits for an LDS round trip
// after it; with two waves per SIMD they are NOT phase-locked: whichever has operands feeds the matrix pipe.
template <int D, int DT, int UT, int NB, bool PK>
__device__ __forceinline__ void g8_sweep(const g8_ctx& c, int plane0, i32x16 (&acc)[D][UT][4]) {
    constexpr int FP = 4 * D * UT;                          // fragments per pair
    constexpr int DPW = 2 * UT;                             // genotype DMAs per wave and pair
    // pairs per unrolled body, so that the ring slot of every fragment is static: the smallest UP with UP FP % NB == 0
    constexpr int UP = (FP % NB == 0) ? 1 : ((2 * FP) % NB == 0) ? 2 : ((3 * FP) % NB == 0) ? 3 : 4;
    static_assert((UP * FP) % NB == 0 && G8_RP >= G8_LA + 2 && G8_RPK >= G8_LA + 2, "ring / unroll shapes");
    const uint32_t lds0 = lds_addr32_i8(c.As);
    auto dma_x = [&](int pc, int i) {
        int cc = pc < c.cntp ? pc : c.cntp - 1;
        if (LOC_GEMM_ABLATE & 16) cc = 0;
        if (LOC_GEMM_ABLATE & 32) cc &= ~3;                 // four consecutive pairs fetch the same lines: 1/4 of them from HBM
        if (PK) {
            uint32_t koff = (uint32_t)(c.p0 + cc) * 32 + c.xpiece[i];           // 32 packed bytes per row and pair
            if (koff > (uint32_t)(c.Kp / 4 - 4)) koff = c.Kp / 4 - 4;          // only in the zero-weight padding of the last pair
            dma4_i8(c.xrow[i] + koff, lds0 + G8_PKOFF + (pc % G8_RPK) * 4096 + (16 * UT * c.w + 8 * i) * 32);
            return;
        }
        uint32_t koff = (uint32_t)(c.p0 + cc) * (2 * G8_BK) + c.xpiece[i];
        if (koff > (uint32_t)(c.Kp - 16)) koff = c.Kp - 16; // only in the zero-weight padding of the last pair
        dma16_i8(c.xrow[i] + koff, lds0 + (pc % G8_RP) * (2 * G8_AIMG) + (16 * UT * c.w + 8 * i) * 128);
    };
    // PK: packed pair pc (landed, published by a barrier) -> int8 ring slot of pair pc: packed word d = 8 m + j is row m,
    // SNPs 16 j .. 16 j + 15 of the pair = the row's 16-byte piece j, which sits at the XOR-swizzled slot the DMA of the
    // unpacked form would have given it
    constexpr int NTH = G8_NT / UT, NPK = 1024 / NTH;        // threads, packed words per thread and pair
    auto unpack_read = [&](int pc, uint32_t (&pk)[NPK]) {
        const unsigned char* src = c.As + G8_PKOFF + (pc % G8_RPK) * 4096;
#pragma unroll
        for (int e = 0; e < NPK; ++e) pk[e] = *reinterpret_cast<const uint32_t*>(src + 4 * (c.w * 64 + c.lane + NTH * e));
    };
    auto unpack_write1 = [&](int pc, uint32_t pkv, int e) {
        unsigned char* dst = c.As + (pc % G8_RP) * (2 * G8_AIMG);
        const int d = c.w * 64 + c.lane + NTH * e, m = d >> 3, j = d & 7;
        u32x4 o;
        o[0] = g8_expand4(pkv & 255u); o[1] = g8_expand4((pkv >> 8) & 255u);
        o[2] = g8_expand4((pkv >> 16) & 255u); o[3] = g8_expand4(pkv >> 24);
        *reinterpret_cast<u32x4*>(dst + m * 128 + ((j ^ ((m >> 1) & 7)) << 4)) = o;
    };
    auto unpack = [&](int pc) {
        uint32_t pk[NPK];
        unpack_read(pc, pk);
#pragma unroll
        for (int e = 0; e < NPK; ++e) unpack_write1(pc, pk[e], e);
    };
    // Fragment j = (step * D + p) * UT + ut of a pair sits at  pair base + voff[j],
    //   voff[j] = lane part (unit tile UT w + ut) + e DT TILE + kk 8192 + p TILE      (step = 2 e + kk)
    uint32_t voff[FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        const int ut = j % UT, sp = j / UT, st = sp / D, p = sp - st * D;
        voff[j] = (uint32_t)(c.hi * 4096 + ((c.w * UT + ut) * 32 + c.jl) * 16 + (st >> 1) * (DT * G8_TILE) +
                             (st & 1) * 8192 + (plane0 + p) * G8_TILE);
    }
    auto pair_base = [&](int pc) -> const unsigned char* { // pairs past the end re-read the last one (never used)
        const int cc = pc < c.cntp ? pc : c.cntp - 1;
        return c.tiles + (int64_t)(c.p0 + cc) * (2 * DT * G8_TILE);
    };
    auto load_b = [&](i32x4& R, const unsigned char* sbase, uint32_t vo) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(R) : "v"(vo), "s"(sbase) : "memory");
    };
    uint32_t aoff[4];                                       // A fragment of step st = 2 e + kk: piece 4 e + 2 kk + hi of row jl
#pragma unroll
    for (int st = 0; st < 4; ++st)
        aoff[st] = lds0 + c.jl * 128 + (((2 * st + c.hi) ^ ((c.jl >> 1) & 7)) << 4);

    // prologue: pairs 0..LA-1 and fragments 0..NB-1 requested, step 0 of pair 0 fetched
    i32x4 B[NB];
    i32x4 A[2][4];
#pragma unroll
    for (int pc = 0; pc < G8_LA + (PK ? 1 : 0); ++pc)
#pragma unroll
        for (int i = 0; i < DPW; ++i) dma_x(pc, i);
#pragma unroll
    for (int f = 0; f < NB; ++f) load_b(B[f], pair_base(f / FP), voff[f % FP]);
    wait_vm_i8<0>();
    barrier_i8();
    if (PK) {                                               // pairs 0 and 1 expanded before the loop, pair pc + 2 inside it
        unpack(0);
        unpack(1);
        barrier_i8();
    }
    rd4_i8(A[0][0], A[0][1], A[0][2], A[0][3], aoff[0]);

    constexpr int N_DMA = (G8_LA - 1) * (DPW + FP) + FP / 2;
    auto pair = [&](int pc0, auto uc) {
        constexpr int u = decltype(uc)::value;             // position inside the unrolled body: ring slots are static
        const int pc = pc0 + u;
        if (!(LOC_GEMM_ABLATE & 2)) {
#pragma unroll
            for (int i = 0; i < DPW; ++i) dma_x(pc + G8_LA + (PK ? 1 : 0), i);
        }
        const uint32_t so = (pc % G8_RP) * (2 * G8_AIMG), so1 = ((pc + 1) % G8_RP) * (2 * G8_AIMG);
        // the pairs the fragments requested in this iteration belong to: NB fragments ahead
        const unsigned char* const sb_lo = pair_base(pc + NB / FP);
        const unsigned char* const sb_hi = pair_base(pc + NB / FP + 1);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st == 2 && !(LOC_GEMM_ABLATE & 4)) {
                wait_vm_i8<N_DMA>();                        // my rows of pair pc + 1 (PK: packed pair pc + 2) are in the ring
                barrier_i8();                               // ... and so are everyone's; nobody still reads pair pc - 1
                //                                            (PK: and pair pc + 1, expanded during the last half pair, is complete)
            } else {
                wait_lgkm0_i8();                            // A[st & 1] has landed
            }
            if (LOC_GEMM_ABLATE & 8) {}
            else if (st < 3) rd4_i8(A[(st + 1) & 1][0], A[(st + 1) & 1][1], A[(st + 1) & 1][2], A[(st + 1) & 1][3], aoff[st + 1] + so);
            else rd4_i8(A[0][0], A[0][1], A[0][2], A[0][3], aoff[0] + so1);
#pragma unroll
            for (int p = 0; p < D; ++p)
#pragma unroll
                for (int ut = 0; ut < UT; ++ut) {
                    const int j = (st * D + p) * UT + ut;   // fragment of the pair; static after unrolling
                    const int slot = (u * FP + j) % NB;
                    // static after unrolling: the if-chain stands in for a template argument that depends on loop variables
                    {
                        const int h = g8_heads(j, FP, NB);
                        if (h == 0) wait_vm_i8<NB - 1>();
                        else if (h == 1) wait_vm_i8<NB - 1 + DPW>();
                        else if (h == 2) wait_vm_i8<NB - 1 + 2 * DPW>();
                        else if (h == 3) wait_vm_i8<NB - 1 + 3 * DPW>();
                        else if (h == 4) wait_vm_i8<NB - 1 + 4 * DPW>();
                        else wait_vm_i8<0>();
                    }
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm)
                        acc[p][ut][tm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[st & 1][tm], B[slot], acc[p][ut][tm], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // PK: the expansion rides behind the first MFMA group after the rendezvous (vector ALU and LDS are idle
                    // while the matrix pipe works through the four MFMAs just issued)
                    if (PK && st >= 2) {                    // one packed word behind each MFMA group of steps 2 and 3
                        constexpr int GPS = D * UT;         // MFMA groups per step
                        const int gi = (st - 2) * GPS + p * UT + ut;
                        if (gi < NPK) {                     // read + expand + store one word: no register is held across groups
                            const unsigned char* src = c.As + G8_PKOFF + ((pc + 2) % G8_RPK) * 4096;
                            unpack_write1(pc + 2, *reinterpret_cast<const uint32_t*>(src + 4 * (c.w * 64 + c.lane + NTH * gi)), gi);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (!(LOC_GEMM_ABLATE & 1)) load_b(B[slot], (j + NB % FP) < FP ? sb_lo : sb_hi, voff[(j + NB) % FP]);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };
    int pc = 0;
    for (; pc + UP <= c.cntp; pc += UP) {
        pair(pc, std::integral_constant<int, 0>{});
        if (UP > 1) pair(pc, std::integral_constant<int, 1>{});
        if (UP > 2) pair(pc, std::integral_constant<int, 2>{});
        if (UP > 3) pair(pc, std::integral_constant<int, 3>{});
    }
    if (UP > 1 && pc < c.cntp) {                            // up to UP - 1 pairs left; the ring is back at slot 0
        pair(pc, std::integral_constant<int, 0>{});
        if (UP > 2 && pc + 1 < c.cntp) pair(pc, std::integral_constant<int, 1>{});
        if (UP > 3 && pc + 2 < c.cntp) pair(pc, std::integral_constant<int, 2>{});
    }
    // requests past the end (clamped, never used) are still landing: drain them while their registers are allocated
    wait_vm_i8<0>();
    wait_lgkm0_i8();
#pragma unroll
    for (int f = 0; f < NB; ++f) asm volatile("" ::"v"(B[f]));
    asm volatile("" ::"v"(A[0][0]), "v"(A[0][1]), "v"(A[0][2]), "v"(A[0][3]));
    __syncthreads();                                        // the epilogue's staging images overlap the ring
}

// UT = 1: 512 threads, a wave = 128 rows x 32 units, 12 fragments in flight; UT = 2: 256 threads, one wave per SIMD,
// a wave = 128 rows x 64 units, 32 fragments in flight.
template <int DT, int UT, bool PK = false>
__global__ __launch_bounds__(G8_NT / UT) void l1_gemm_i8_kernel(const uint8_t* __restrict__ X, int64_t pitch,
                                                                 const int32_t* __restrict__ rows, int n, int Kp,
                                                                 const unsigned char* __restrict__ tiles,
                                                                 const float* __restrict__ delta,
                                                                 float* __restrict__ partial, int G, int n_mt, int npairs) {
#ifndef G8_NB_UT1
#define G8_NB_UT1 12       // digit fragments in flight per wave, 8-wave form (4 / 6 / 8 / 12: same time; 16 spills)
#endif
    constexpr int NB = UT == 1 ? G8_NB_UT1 : 32;
    constexpr int WU = 32 * UT;                             // units per wave
    extern __shared__ __attribute__((aligned(1024))) unsigned char g8_smem[];
    const int t = threadIdx.x, lane = t & 63;
    g8_ctx c;
    c.w = __builtin_amdgcn_readfirstlane(t >> 6);
    c.lane = lane;
    c.jl = lane & 31;
    c.hi = lane >> 5;
    // (row tile, SNP group) of this workgroup.  Workgroup b runs on XCD b % 8 (observed dispatch; a speed hint only), and the
    // workgroups that share a SNP group - the same weight tiles, different row tiles - should share an XCD's L2.
    //   G a multiple of 8: XCD x takes the groups x, x + 8, ... whole (rounds 3-4).
    //   any other G (round 5): the n_mt x G pairs, ordered group-major, are cut into eight contiguous runs of equal length
    //   (+- 1), one per XCD; the k-th workgroup of an XCD takes the k-th pair of its run, surplus workgroups (the grid is 8 x
    //   the longest run) leave at once.  Rounds 3-4 rounded G DOWN to a multiple of 8 instead and left up to 37 % of the
    //   compute units idle at in-between row counts: 2500 rows = 20 row tiles x 8 groups = 160 workgroups, now 20 x 12 = 240:
    //   155 -> 124 us; 3000 rows 163 -> 146; 1500 rows 79 -> 75 (same box, profiles/r05_gemm_group_mapping.log).  For
    //   multiples of 8 the run form measured 2 % slower at 1000 rows (56.6 against 55.4 us) and equal elsewhere, so they
    //   keep the strided form.
    int mt, g;
    if ((G & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        g = xcd + 8 * (idx / n_mt);
    } else {
        const int P = n_mt * G, x = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int s0 = (int)((int64_t)x * P / 8), s1 = (int)((int64_t)(x + 1) * P / 8);
        if (k >= s1 - s0) return;
        const int pr = s0 + k;
        g = pr / n_mt;
        mt = pr - g * n_mt;
    }
    // group g owns a CONTIGUOUS run of pairs (balanced to within one): a workgroup then walks each of its 128 genotype
    // rows sequentially, which HBM serves at 6.0 TB/s where the interleaved assignment of l1_gemm.hip (every G-th pair)
    // tops out at 4.2 (tools/probes/geno_stream_probe.hip, profiles/r03_geno_stream_probe.jsonl)
    const int qp = npairs / G, rp = npairs - qp * G;
    c.p0 = g * qp + (g < rp ? g : rp);
    c.cntp = qp + (g < rp ? 1 : 0);
    c.Kp = Kp;
    c.tiles = tiles;
    c.As = g8_smem;
#pragma unroll
    for (int i = 0; i < 2 * UT; ++i) {
        const int m = 16 * UT * c.w + 8 * i + (lane >> 3);  // row of the 128-row tile this lane moves
        int r = mt * G8_BM + m;
        if (r > n - 1) r = n - 1;
        c.xrow[i] = X + (int64_t)rows[r] * pitch;           // PK: X / pitch are the packed matrix's
        c.xpiece[i] = PK ? (uint32_t)((lane & 7) << 2) : (uint32_t)(((lane & 7) ^ ((m >> 1) & 7)) << 4);
    }
    const int Mp = n_mt * G8_BM;

    i32x16 acc[2][UT][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) acc[p][ut][tm] = i32x16{0};
    g8_sweep<2, DT, UT, NB, PK>(c, 0, acc);

    // D[i = row][j = unit] x delta_j: wave tile 128 rows x WU units through a wave-private LDS image, then 16-byte stores.
    // Three planes: the two leading ones go out first (65536 * plane 0 + 256 * plane 1, in units of delta), the K range
    // is walked again for the least significant plane, and the same thread adds it to what it stored - delta is a power
    // of two, so the sum equals the one a three-plane accumulator set would have produced, without the registers.
    float dl[UT];
#pragma unroll
    for (int ut = 0; ut < UT; ++ut) dl[ut] = delta[(c.w * UT + ut) * 32 + c.jl];
    float* const ep = reinterpret_cast<float*>(g8_smem) + c.w * (G8_BM * WU);
    float* const pout = partial + ((int64_t)g * Mp + mt * G8_BM) * G8_HP + c.w * WU;
    auto emit = [&](const i32x16 (&hi_p)[UT][4], const i32x16 (&lo_p)[UT][4], float s_hi, float s_lo, bool add) {
#if LOC_GEMM_ABLATE & 128
        {   // keep the accumulators alive without the epilogue: one word per lane that is (almost) never stored
            int keep = 0;
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) keep ^= hi_p[ut][tm][r] ^ lo_p[ut][tm][r];
            if (keep == 0x7fffffff && s_hi == -1.f) pout[0] = (float)keep;
            return;
        }
#endif
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[(tm * 32 + rowmap(r, c.hi)) * WU + ut * 32 + c.jl] =
                        (s_hi * (float)hi_p[ut][tm][r] + s_lo * (float)lo_p[ut][tm][r]) * dl[ut];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int LPR = WU / 4, RPI = 64 / LPR;        // lanes per row, rows per wave instruction
#pragma unroll
        for (int i = 0; i < G8_BM / RPI; ++i) {
            const int row = i * RPI + lane / LPR, c4 = (lane % LPR) * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * WU + c4);
            f32x4* dst = reinterpret_cast<f32x4*>(pout + (int64_t)row * G8_HP + c4);
            if (add) v = v + *dst;
#if LOC_GEMM_ABLATE & 64
            if (v[0] != 12345.678f) continue;           // (timing build: the store almost never happens)
#endif
            *dst = v;
        }
    };
    if (DT == 2) {
        emit(acc[0], acc[1], 256.f, 1.f, false);
    } else {
        emit(acc[0], acc[1], 65536.f, 256.f, false);
        __syncthreads();                                    // the staging images overlap the genotype ring
        i32x16 lo[1][UT][4];
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) lo[0][ut][tm] = i32x16{0};
        g8_sweep<1, DT, UT, NB, PK>(c, 2, lo);
        emit(lo[0], lo[0], 1.f, 0.f, true);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static int g8_nkt64(const loc_dims* d) { return ((d->Kp + G8_BK - 1) / G8_BK + 1) & ~1; }

extern "C" int loc_l1_gemm_i8_supported(int Hp, int digits) { return Hp == G8_HP && (digits == 2 || digits == 3); }

// image = [cvec8: 8*Hp floats][delta: Hp floats][colmax: Hp uints][guard: Hp floats: 4 used + the tail ticket at word 64]
// [shares: 2*grid*Hp floats][tail rows: 32*Hp floats][tiles], 1 KB aligned sections; shares = the scan's per-workgroup sums
// (sum |w'| first, then the shift term), grid = g8_scan_grid; the tail rows / ticket belong to the image kernel's leftover
// workgroups (the ticket is zeroed by the scan's memset together with colmax and the guard, and counts modulo 16 after that:
// sixteen tickets per image).
static int64_t g8_delta_off() { return 8 * G8_HP * 4; }
static int64_t g8_colmax_off() { return g8_delta_off() + G8_HP * 4; }
static int64_t g8_guard_off() { return g8_colmax_off() + G8_HP * 4; }
static int64_t g8_cpart_off() { return g8_guard_off() + G8_HP * 4; }
static int g8_scan_grid(const loc_dims* d) { const int nkt = g8_nkt64(d); return nkt < 256 ? nkt : 256; }   // one workgroup per compute unit
// [shares: 2 * grid rows][tail rows: 32], rows of Hp floats; the tail ticket is word 64 of the guard section
static int64_t g8_tail_rows_off(const loc_dims* d) { return g8_cpart_off() + (int64_t)2 * g8_scan_grid(d) * G8_HP * 4; }
static int64_t g8_tail_ticket_off() { return g8_guard_off() + 64 * 4; }
static int64_t g8_tiles_off(const loc_dims* d) { return (g8_tail_rows_off(d) + (int64_t)32 * G8_HP * 4 + 1023) / 1024 * 1024; }
extern "C" int64_t loc_l1_image_i8_bytes(const loc_dims* d, int digits) {
    if (!loc_l1_gemm_i8_supported(d->Hp, digits)) return 0;
    return g8_tiles_off(d) + (int64_t)g8_nkt64(d) * digits * G8_TILE;
}

// colmax + per-unit mean magnitude + shift-term shares (l1_scan_kernel); with_guard: also the guard as a launch of its own
static int g8_scan(const loc_dims* d, const float* scale_shift, const float* w1s, void* image, bool with_guard, void* stream) {
    if (d->Hp != G8_HP) { loc_set_error("loc_l1_quant_scan: needs padded width 256 (got %d)", d->Hp); return -1; }
    unsigned char* base = static_cast<unsigned char*>(image);
    uint32_t* colmax = reinterpret_cast<uint32_t*>(base + g8_colmax_off());
    float* guard = reinterpret_cast<float*>(base + g8_guard_off());
    float* shares = reinterpret_cast<float*>(base + g8_cpart_off());
    const int nkt = g8_nkt64(d), grid = g8_scan_grid(d);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(colmax, 0, 2 * G8_HP * 4, st);      // colmax and the guard section behind it (with the tail ticket)
    if (e != hipSuccess) { loc_set_error("loc_l1_quant_scan: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(l1_scan_kernel, dim3(grid), dim3(G8_HP), 0, st, w1s, scale_shift, d->Kp, nkt, colmax, shares,
                       shares + (int64_t)grid * G8_HP);
    LOC_CHECK_LAUNCH();
    if (with_guard) {             // a caller that reads the guard back before the image is built (loc_predict_scan)
        hipLaunchKernelGGL(l1_quant_guard_kernel, dim3(1), dim3(1024), 0, st, colmax, shares, grid, d->K, d->H, guard);
        LOC_CHECK_LAUNCH();
    }
    return 0;
}
extern "C" int loc_l1_quant_scan(const loc_dims* d, const float* scale_shift, const float* w1s, void* image, void* stream) {
    return g8_scan(d, scale_shift, w1s, image, true, stream);
}
extern "C" int64_t loc_l1_image_i8_guard_offset(void) { return g8_guard_off(); }
extern "C" int64_t loc_l1_image_i8_tiles_offset(const loc_dims* d) { return g8_tiles_off(d); }

static int g8_image_build(const loc_dims* d, const float* scale_shift, const float* w1s, int digits, void* image,
                          bool scanned, void* stream) {
    if (!loc_l1_gemm_i8_supported(d->Hp, digits)) {
        loc_set_error("loc_l1_image_i8_build: width %d / %d digits unsupported (needs padded width 256, 2 or 3 digits)",
                      d->Hp, digits);
        return -1;
    }
    unsigned char* base = static_cast<unsigned char*>(image);
    float* cvec = reinterpret_cast<float*>(base);
    float* delta = reinterpret_cast<float*>(base + g8_delta_off());
    uint32_t* colmax = reinterpret_cast<uint32_t*>(base + g8_colmax_off());
    float* guard = reinterpret_cast<float*>(base + g8_guard_off());
    const float* shares = reinterpret_cast<const float*>(base + g8_cpart_off());
    float* tail_rows = reinterpret_cast<float*>(base + g8_tail_rows_off(d));
    unsigned* tail_ticket = reinterpret_cast<unsigned*>(base + g8_tail_ticket_off());
    unsigned char* tiles = base + g8_tiles_off(d);
    const int nkt = g8_nkt64(d), grid = g8_scan_grid(d);
    hipStream_t st = (hipStream_t)stream;
    if (!scanned) {               // two launches per image (+ the 1 KB memset): scan, then tiles with the tail workgroup
        const int rc = g8_scan(d, scale_shift, w1s, image, false, stream);
        if (rc) return rc;
    }
    if (digits == 2)
        hipLaunchKernelGGL(l1_image_i8_kernel<2>, dim3(nkt + G8_TAIL_WGS), dim3(G8_HP), 0, st, w1s, scale_shift, d->Kp, nkt, d->K, d->H, colmax,
                           delta, tiles, shares, shares + (int64_t)grid * G8_HP, grid, guard, cvec, tail_rows, tail_ticket);
    else
        hipLaunchKernelGGL(l1_image_i8_kernel<3>, dim3(nkt + G8_TAIL_WGS), dim3(G8_HP), 0, st, w1s, scale_shift, d->Kp, nkt, d->K, d->H, colmax,
                           delta, tiles, shares, shares + (int64_t)grid * G8_HP, grid, guard, cvec, tail_rows, tail_ticket);
    LOC_CHECK_LAUNCH();
    return 0;
}
extern "C" int loc_l1_image_i8_build(const loc_dims* d, const float* scale_shift, const float* w1s, int digits,
                                     void* image, void* stream) {
    return g8_image_build(d, scale_shift, w1s, digits, image, false, stream);
}
extern "C" int loc_l1_image_i8_build_scanned(const loc_dims* d, const float* scale_shift, const float* w1s, int digits,
                                             void* image, void* stream) {
    return g8_image_build(d, scale_shift, w1s, digits, image, true, stream);
}

// a1 == nullptr: the SNP-group partial sums stay in `partial` ([groups][ceil(n/128)*128][256] floats; *groups_out groups)
// for a consumer that adds them up itself (loc_stack_forward_eval_partial)
static int g8_forward(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n, const loc_dims* d, const void* image,
                      int digits, int x_max, const float* b1, float* partial, int64_t partial_floats, float* a1,
                      int target_blocks, const loc_tuning* tune, bool packed, void* stream, int* groups_out = nullptr) {
    if (n < 1) { loc_set_error("loc_l1_forward_gemm_i8: n=%d", n); return -1; }
    if (!loc_l1_gemm_i8_supported(d->Hp, digits)) {
        loc_set_error("loc_l1_forward_gemm_i8: width %d / %d digits unsupported", d->Hp, digits);
        return -1;
    }
    if (x_max < 1 || x_max > 127) {
        loc_set_error("loc_l1_forward_gemm_i8: genotypes must be known to lie in 0..127 (x_max = %d); use loc_l1_forward_gemm",
                      x_max);
        return -1;
    }
    if (packed) {
        if (x_max > 3 || d->Kp % 16 || x_pitch % 4 || x_pitch < d->Kp / 4 || ((uintptr_t)X & 3)) {
            loc_set_error("loc_l1_forward_gemm_i8_packed: needs genotypes <= 3 (x_max = %d), a 4-byte aligned packed matrix and "
                          "a 4-byte row pitch >= Kp / 4", x_max);
            return -1;
        }
    } else if (d->Kp < 16 || d->Kp % 16 || x_pitch % 16 || x_pitch < d->Kp || ((uintptr_t)X & 15)) {
        loc_set_error("loc_l1_forward_gemm_i8: needs a 16-byte aligned X, Kp %% 16 == 0 and a 16-byte row pitch >= Kp");
        return -1;
    }
    const int nkt = g8_nkt64(d);
    const int n_mt = (n + G8_BM - 1) / G8_BM, Mp = n_mt * G8_BM;
    if (target_blocks < 1) target_blocks = 256;
    int G = target_blocks / n_mt;
    const int64_t cap = partial_floats / ((int64_t)Mp * G8_HP);
    if (G > cap) G = (int)cap;
    if (G > nkt / 2) G = nkt / 2;
    if (G < 1) { loc_set_error("loc_l1_forward_gemm_i8: scratch too small for %d rows", n); return -1; }
    // an i32 accumulator holds sum_k x d with |d| <= 128 over one group's SNPs
    const int64_t snps_per_group = (int64_t)((nkt / 2 + G - 1) / G) * 2 * G8_BK;
    if ((int64_t)x_max * 128 * snps_per_group >= ((int64_t)1 << 31)) {
        loc_set_error("loc_l1_forward_gemm_i8: %lld SNPs per group with genotypes up to %d could overflow int32",
                      (long long)snps_per_group, x_max);
        return -1;
    }
    const unsigned char* base = static_cast<const unsigned char*>(image);
    const float* cvec = reinterpret_cast<const float*>(base);
    const float* delta = reinterpret_cast<const float*>(base + g8_delta_off());
    const unsigned char* tiles = base + g8_tiles_off(d);
    hipStream_t st = (hipStream_t)stream;
    const int g8_grid = 8 * ((n_mt * G + 7) / 8);       // eight runs of pairs, the longest decides (see the kernel)
#define G8_LAUNCH(DTV, UTV)                                                                                     \
    {                                                                                                           \
        if (packed) {                                                                                           \
            LOC_ENSURE_LDS((l1_gemm_i8_kernel<DTV, 1, true>), G8_LDS_PK);                                       \
            hipLaunchKernelGGL((l1_gemm_i8_kernel<DTV, 1, true>), dim3(g8_grid), dim3(G8_NT), G8_LDS_PK, st, X,     \
                               x_pitch, rows, n, d->Kp, tiles, delta, partial, G, n_mt, nkt / 2);               \
        } else {                                                                                                \
            LOC_ENSURE_LDS((l1_gemm_i8_kernel<DTV, UTV, false>), G8_LDS);                                       \
            hipLaunchKernelGGL((l1_gemm_i8_kernel<DTV, UTV, false>), dim3(g8_grid), dim3(G8_NT / UTV), G8_LDS, st, X, \
                               x_pitch, rows, n, d->Kp, tiles, delta, partial, G, n_mt, nkt / 2);               \
        }                                                                                                       \
    }
    const int ut = tune && (tune->gemm_i8_unit_tiles == 1 || tune->gemm_i8_unit_tiles == 2) ? tune->gemm_i8_unit_tiles
                                                                                             : G8_WAVE_UNIT_TILES;
    if (ut == 2 && !packed) {             // (packed genotypes: the 8-wave form only; the 4-wave form has no registers left for the expansion)
        if (digits == 2) G8_LAUNCH(2, 2) else G8_LAUNCH(3, 2)
    } else {
        if (digits == 2) G8_LAUNCH(2, 1) else G8_LAUNCH(3, 1)
    }
#undef G8_LAUNCH
    LOC_CHECK_LAUNCH();
    if (groups_out) *groups_out = G;
    if (!a1) return 0;
    return gm_launch_reduce(partial, G, (int64_t)Mp * G8_HP, cvec, b1, a1, stream);
}

extern "C" int loc_l1_forward_gemm_i8_partial(const uint8_t* X, int64_t x_pitch, int packed, const int32_t* rows, int n,
                                              const loc_dims* d, const void* image, int digits, int x_max, float* partial,
                                              int64_t partial_floats, int target_blocks, const loc_tuning* tune,
                                              int* h_groups, const float** cvec8, void* stream) {
    if (!h_groups || !cvec8) { loc_set_error("loc_l1_forward_gemm_i8_partial: h_groups / cvec8 must not be NULL"); return -1; }
    *cvec8 = reinterpret_cast<const float*>(image);
    return g8_forward(X, x_pitch, rows, n, d, image, digits, packed ? 3 : x_max, nullptr, partial, partial_floats, nullptr,
                      target_blocks, tune, packed != 0, stream, h_groups);
}

extern "C" int loc_l1_forward_gemm_i8(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n, const loc_dims* d,
                                      const void* image, int digits, int x_max, const float* b1, float* partial,
                                      int64_t partial_floats, float* a1, int target_blocks, const loc_tuning* tune,
                                      void* stream) {
    return g8_forward(X, x_pitch, rows, n, d, image, digits, x_max, b1, partial, partial_floats, a1, target_blocks, tune,
                      false, stream);
}

extern "C" int loc_l1_forward_gemm_i8_packed(const uint8_t* X2, int64_t x2_pitch, const int32_t* rows, int n,
                                             const loc_dims* d, const void* image, int digits, const float* b1,
                                             float* partial, int64_t partial_floats, float* a1, int target_blocks,
                                             const loc_tuning* tune, void* stream) {
    return g8_forward(X2, x2_pitch, rows, n, d, image, digits, 3, b1, partial, partial_floats, a1, target_blocks, tune, true,
                      stream);
}
