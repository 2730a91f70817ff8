// Large-M first-layer genotype GEMM on the bf16 matrix pipe, weights converted ONCE per sweep
// (reference: model.predict on predgen / testgen and the --jacknife replicate predictions,
// /root/reference/locator/locator.py:414, :441, :683-747).
//
//     z1[m][h] = sum_k x[m][k] (s_k W1[k][h])  +  sum_k t_k W1[k][h]          (BatchNorm in inference form)
//
// l1_rows.hip converts the fp32 weight tile (scale, split into bf16 pieces, pack) inside the K loop of EVERY
// 128-row tile, which made it vector-ALU bound (PMC, round 1: matrix pipe 43 % busy, VALU 28 %, waits 31 %); it
// stays the kernel for few rows (the per-epoch validation sweep), where converting once would not pay.  Here the
// conversion is its own streaming pass:
//
//   l1_image_kernel      W1S (fp32, swizzled) x BN scale  ->  HBM image of bf16 tiles, one 32 KB tile per
//                        (64-SNP block, piece), laid out [8-SNP chunk c][unit n][8 SNPs]: the MFMA B operand of a
//                        32-unit group for one k-step is two contiguous 512-byte runs.  3 pieces = the exact
//                        8+8+8-bit truncation split (fp32-exact products), 1 piece = round-to-nearest bf16.  Also the
//                        per-unit shift term c[h] = sum_k t_k W1[k][h] (per-block partial sums, fixed-order reduction).
//                        Streams at the HBM rate (26 us for 1 piece, 46 us for 3 at K = 100,000).
//   l1_gemm_kernel       workgroup = 8 waves on a 128-row x 256-unit tile, split over SNP groups (a group owns every
//                        G-th PAIR of adjacent 64-SNP blocks).  A wave owns 32 units and ALL 128 rows (4 row tiles x 1
//                        unit tile), so its weight fragments are private: HBM/L2 -> VGPRs in the B-operand layout
//                        (16-byte loads, three tiles ahead, three static register sets), never through the LDS.  Only
//                        the genotype block is shared: u8 rows -> registers -> bf16 image in a 4-slot LDS ring (1 cvt
//                        + 1/2 perm per genotype, once per workgroup and block), read as MFMA A operands by 16-byte
//                        loads; reads and writes are bank-conflict free.  A genotype request covers the pair's whole
//                        128-byte line of a row (8 lanes a row).  The two halves of the workgroup (waves 0-3 / 4-7:
//                        the two waves of each SIMD) run half a block apart between two barriers per block: one
//                        issues the block's P x 16 MFMAs from registers while its SIMD partner reads fragments, widens
//                        and requests.  Global requests are asm with hand-counted vmcnt.  MFMA 32x32x16 bf16, fp32
//                        accumulation.
//   l1_gemm_reduce_kernel  fixed-order sum of the SNP-group partials + shift term + b1, ELU.
//
// Blocks that share a SNP group (same weight tiles, different row tiles) sit on one XCD (block b runs on XCD b % 8): a
// weight tile leaves HBM once.  Placement is a speed hint only.
//
// What was built and measured on the way (1000 rows x 100,000 SNPs x 256 units, kernel time 1 piece / 3 pieces;
// in-kernel s_memtime stamps, timing ablations and PMC in profiles/r02_gemm_*; the schedules are in the git history):
//   (1) LDS-DMA lockstep: weight tiles and raw genotypes by global_load_lds into a 3-slot ring, counted vmcnt,
//       raw s_barrier per tile, 160 KB LDS: 70 / 147 us.  Ablations add up instead of overlapping (genotype widening
//       +11 us, weight requests +13, fragment reads +9.5 on a 36 us matrix-only run): one global_load_lds costs the
//       issuing wave ~150 cycles, and all eight waves do the same thing at the same time.
//   (2) the same with two groups of four waves half a step apart: 73 / 163 us (a matrix segment stretches from 750 to
//       1,100-1,700 cycles whenever the partner moves data through the LDS by DMA).
//   (3) register-staged tiles (global -> VGPR -> ds_write_b128), one barrier per tile: 63 / 151 us.
//   (4) weight fragments straight to VGPRs, only genotypes through the LDS, all 8 waves in lockstep, one barrier per
//       block: 60 / 145 us, matrix pipe 48 % / 57 % busy.  Stamps: per block ~1,000 cycles of MFMA issue for both
//       waves of a SIMD, then ~700-1,000 cycles in which neither issues one (widen + requests + barrier skew).
//   (5) this kernel.  Steps, each measured at 4096 rows, 1 piece: (4) 221 us -> half-a-block phase shift between the
//       two waves of each SIMD, A fragments held in registers across the phase: 209 -> weight requests interleaved
//       one per 4 MFMAs (a request stalls its wave ~100 cycles; the 4 queued MFMAs cover it): 205 -> 128-byte
//       genotype lines (8 rows per request instead of 16 half lines; ablation: the genotype REQUEST, not the widening
//       or the LDS traffic, was the most expensive item, 55 us of 209): -> hand-counted vmcnt (the compiler's counts
//       were 1-4 where 9-14 are right after its loop-header merge): 193 us.  1000 rows: 57 / 139 us.
//       What is left (ablations at 4096 rows, 1 piece): MFMAs + barriers alone 127 us (0.66 of the bf16 peak: the
//       clock under sustained MFMA load), weight requests +25, genotype stream +30 when it comes from HBM / MALL
//       (+5 when the same lines are L2-hot; vmcnt returns in order, so a slow genotype line holds back the weight
//       fragments queued behind it), deeper genotype prefetch (6 blocks) and s_setprio either way: no gain.
#include "common.h"
#include <type_traits>

#define GM_BM 128
#define GM_BK 64
#define GM_NT 512
#define GM_HP 256
#define GM_BTILE (GM_HP * GM_BK * 2) /* 32768: one (SNP block, piece) weight tile, bf16 */
#define GM_AIMG (GM_BM * GM_BK * 2)  /* 16384: bf16 genotype image of one SNP block    */
#define GM_LDS 131072                /* 4 genotype images in the loop; the epilogue stages 8 x 16 KB of partials */

__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t rne16g(uint32_t u) { return u + 0x7FFFu + ((u >> 16) & 1u); }

// ---------------------------------------------------------------------------------------------------------
// weight image
// ---------------------------------------------------------------------------------------------------------
// One workgroup per 64-SNP block, thread = unit n.  tiles[(kt64*P + p)][c][n][e] = piece p of s_k W1[k][n],
// k = kt64*64 + c*8 + e;  cpart[kt64][n] = sum over the block's SNPs of t_k W1[k][n].
template <int P>
__global__ __launch_bounds__(GM_HP) void l1_image_kernel(const float* __restrict__ w1s, const float* __restrict__ ss4,
                                                         int Kp, unsigned char* __restrict__ tiles,
                                                         float* __restrict__ cpart) {
    constexpr int nht = GM_HP / 32;
    const int kt64 = blockIdx.x, n = threadIdx.x;
    const int ht = n >> 5, hl = n & 31, q = hl >> 3, hi = (hl >> 2) & 1, c4 = hl & 3;
    const float* scale = ss4;
    const float* shift = ss4 + Kp;
    float csum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int kt32 = 2 * kt64 + (c >> 2);
        uint32_t pc[P][8];
        if (kt32 * 32 < Kp) {
            const float* src = w1s + ((int64_t)(kt32 * nht + ht) * 4 + q) * 256 + hi * 128 + c4;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kl = (c & 3) * 8 + e, k = kt32 * 32 + kl;
                const float w = src[kl * 4];
                csum = fmaf(shift[k], w, csum);
                float r = w * scale[k];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    uint32_t u = fbits(r);
                    if (p == P - 1 && P < 3) u = rne16g(u);
                    pc[p][e] = u >> 16;
                    if (p < P - 1) r -= bitsf(u & 0xFFFF0000u);
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int e = 0; e < 8; ++e) pc[p][e] = 0;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            u32x4 v;
            v[0] = pc[p][0] | (pc[p][1] << 16);
            v[1] = pc[p][2] | (pc[p][3] << 16);
            v[2] = pc[p][4] | (pc[p][5] << 16);
            v[3] = pc[p][6] | (pc[p][7] << 16);
            *reinterpret_cast<u32x4*>(tiles + ((int64_t)kt64 * P + p) * GM_BTILE + c * (GM_HP * 16) + n * 16) = v;
        }
    }
    cpart[(int64_t)kt64 * GM_HP + n] = csum;
}

// cvec8[s][h] = sum over the SNP blocks kt = s*16 + q + 128 i of cpart[kt][h] (16 strided partial sums q, then 16 adds):
// 8 slices x Hp/64 blocks so the 6 KB-per-unit column is not summed by one wave; the 8 slices are added, in order, by
// l1_gemm_reduce_kernel.
__global__ __launch_bounds__(1024) void l1_image_cvec_kernel(const float* __restrict__ cpart, int nkt64,
                                                             float* __restrict__ cvec8) {
    __shared__ float red[16][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int h = blockIdx.x * 64 + o, sl = blockIdx.y;
    float s0 = 0.f, s1 = 0.f;
    int kt = sl * 16 + q;
    for (; kt + 128 < nkt64; kt += 256) {
        s0 += cpart[(int64_t)kt * GM_HP + h];
        s1 += cpart[(int64_t)(kt + 128) * GM_HP + h];
    }
    if (kt < nkt64) s0 += cpart[(int64_t)kt * GM_HP + h];
    red[q][o] = s0 + s1;
    __syncthreads();
    if (q == 0) {
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) z += red[j][o];
        cvec8[sl * GM_HP + h] = z;
    }
}

// ---------------------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_addr32(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// four A fragments (row tiles 0..3 of one k-step) by four 16-byte LDS reads; volatile so that they stay in the load
// phase, ahead of the barrier that hands the matrix pipe to this wave
__device__ __forceinline__ void rd4(bf16x8& a0, bf16x8& a1, bf16x8& a2, bf16x8& a3, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\t"
                 "ds_read_b128 %2, %4 offset:1024\n\tds_read_b128 %3, %4 offset:1536"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr) : "memory");
}
// Global loads as asm with hand-counted s_waitcnt vmcnt: the compiler's own bookkeeping merges the prologue's and the
// loop's in-flight state at the loop header and then waits for almost everything in the first blocks of every
// unrolled body (seen as vmcnt(2) where 9 loads may stay in flight).  The counts are in the loop below.
// COUNT TABLE (per wave, requests in program order; P = pieces, 4 fragments per (block, piece) tile):
//   load phase of a block      1 genotype request (load_x)
//   matrix phase of a block    P tiles x 4 fragment requests (load_b_part), each issued right after the 4 MFMAs that
//                              consumed the fragment it replaces, three tiles (12 requests) ahead
//   wait before widening       the bytes were requested two blocks ago: younger are 2 x 4P fragment requests and one
//                              genotype request                                               -> vmcnt(8P + 1)
//   wait before a fragment     requested three tiles ago: 11 younger fragment requests plus one genotype request per
//                              block boundary crossed on the way (P = 1: three, P = 2: two for piece 0 and one for
//                              piece 1, P = 3: one)                                           -> vmcnt(14 | 13 | 12)
// Editing the request order in block() means re-deriving these; a -DLOC_GEMM_DEBUG_DRAIN build turns every count into
// vmcnt(0) for parity debugging.
template <typename T>
__device__ __forceinline__ void gload16(T& r, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
#ifdef LOC_GEMM_DEBUG_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
}
// LDS traffic of this wave done, then the workgroup barrier (no vmcnt wait: weight / genotype requests stay in flight)
// and nothing (MFMAs included) scheduled across it
__device__ __forceinline__ void phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int P>
__global__ __launch_bounds__(GM_NT) void l1_gemm_kernel(const uint8_t* __restrict__ X, int64_t pitch,
                                                         const int32_t* __restrict__ rows, int n, int Kp,
                                                         const unsigned char* __restrict__ tiles,
                                                         float* __restrict__ partial, int G, int n_mt, int npairs) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char gm_smem[];
    unsigned char* const As = gm_smem;                         // 4 x 16 KB in the loop (the epilogue reuses 8 x 16 KB)

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int jl = lane & 31, hi = lane >> 5;
    int g, mt;
    if ((G & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        g = xcd + 8 * (idx / n_mt);
    } else {
        g = blockIdx.x % G;
        mt = blockIdx.x / G;
    }
    const int Mp = n_mt * GM_BM;
    // this group's SNP blocks: pairs c = 0..cntp-1 of adjacent blocks, block a = 2c + e  <->  kt64 = 2 (g + c G) + e
    const int cntp = (npairs - g + G - 1) / G;
    const int cnt = 2 * cntp, nB = cnt * P;

    // genotypes: a thread moves 16 bytes (piece q of a pair's 128-byte line) of rows xr and xr + 64
    const int xr = t >> 3, q = t & 7;
    const uint8_t* xsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = mt * GM_BM + xr + 64 * i;
        if (r > n - 1) r = n - 1;
        xsrc[i] = X + (int64_t)rows[r] * pitch;
    }
    auto load_x = [&](u32x4& R, int c, int i) {
        const int cc = c < cntp ? c : cntp - 1;
        int koff = (g + cc * G) * (2 * GM_BK) + 16 * q;
        if (koff > Kp - 16) koff = Kp - 16;                    // only in the zero-weight padding of the last pair
        gload16(R, xsrc[i] + koff);
    };
    // image of block a: chunk c8 (8 SNPs) of row m at  c8*2048 + ((m ^ (c8 & 6) ^ (a & 1)) << 4)  - the 8 lanes of
    // one 16-byte store group (one row, both blocks of the pair) fall on 8 different 16-byte bank groups, and so do
    // the 16 lanes of one read group
    const uint32_t woff = (q >> 2) * GM_AIMG + (2 * (q & 3)) * 2048 + ((xr ^ (2 * (q & 3)) ^ (q >> 2)) << 4);
    auto widen = [&](const u32x4& R, int c, int i) {
        u32x4 o0, o1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t b = R[d];
            const uint32_t lo = pack_top16(fbits((float)(b & 255u)), fbits((float)((b >> 8) & 255u)));
            const uint32_t hh = pack_top16(fbits((float)((b >> 16) & 255u)), fbits((float)(b >> 24)));
            if (d < 2) { o0[2 * d] = lo; o0[2 * d + 1] = hh; }
            else { o1[2 * (d - 2)] = lo; o1[2 * (d - 2) + 1] = hh; }
        }
        unsigned char* ad = As + woff + ((2 * c) & 3) * GM_AIMG + i * 1024;
        *reinterpret_cast<u32x4*>(ad) = o0;
        *reinterpret_cast<u32x4*>(ad + 2048) = o1;
    };
    // this wave's weight fragments of tile j (block j / P, piece j % P): chunk 2 kk + hi, unit w*32 + jl
    struct bregs { bf16x8 b[4]; };
    const int b_lane = hi * 4096 + (w * 32 + jl) * 16;
    auto load_b_part = [&](bregs& R, int j, int k0, int k1) {
        const int jj = j < nB ? j : nB - 1;
        const int a = jj / P, p = jj - a * P;
        const int kt = 2 * (g + (a >> 1) * G) + (a & 1);
        const unsigned char* src = tiles + ((int64_t)kt * P + p) * GM_BTILE + b_lane;
#pragma unroll
        for (int kk = k0; kk < k1; ++kk) gload16(R.b[kk], src + kk * 8192);
    };

    f32x16 acc[4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) acc[tm] = f32x16{0};
    uint32_t aoff[2][4];                                        // A-fragment addresses for even / odd blocks
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            aoff[e][kk] = lds_addr32(As) + (2 * kk + hi) * 2048 + ((jl ^ (2 * kk) ^ e) << 4);

    // prologue: pair 0 widened at once, pair 1 and weight tiles 0..2 requested
    u32x4 XR[2];
    bregs B0, B1, B2;
    {
        u32x4 x0, x1;
        load_x(x0, 0, 0);
        load_x(x1, 0, 1);
        load_x(XR[0], 1, 0);
        load_x(XR[1], 1, 1);
        load_b_part(B0, 0, 0, 4);
        load_b_part(B1, 1, 0, 4);
        load_b_part(B2, 2, 0, 4);
        wait_vm<0>();
        widen(x0, 0, 0);
        widen(x1, 0, 1);
    }
    __syncthreads();

    // Two groups of four waves (one wave per SIMD each: waves w and w + 4 share a SIMD) run half a block apart.  In
    // its load phase a wave reads the block's 16 A fragments into registers, widens its share of the NEXT pair of
    // blocks into the ring and requests the pair after that; in its matrix phase it issues the block's P x 16 MFMAs
    // from registers only, each group of 4 followed by the request for the weight fragment it just consumed, three
    // tiles ahead.  The phases of the two groups alternate between the same barriers, so a SIMD's matrix pipe is
    // fed by one wave while its partner moves data.
    const int grp = w >> 2;
    if (grp == 1) phase_barrier();
    constexpr int NBLK = (P == 3) ? 2 : 6;                      // unrolled blocks: register set and block parity static
    auto block = [&](int bb, auto alc) {
        constexpr int al = decltype(alc)::value;
        const int ai = bb + al;
        const uint32_t so = (ai & 3) * GM_AIMG;
        bf16x8 a[4][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) rd4(a[kk][0], a[kk][1], a[kk][2], a[kk][3], aoff[al & 1][kk] + so);
        // vmcnt is in order.  A block issues 1 genotype request, then 4 weight requests per tile.  The genotypes
        // widened now were requested two blocks ago: 4P + 1 + 4P younger requests may stay in flight.
        wait_vm<8 * P + 1>();
        widen(XR[al & 1], (ai >> 1) + 1, al & 1);
        load_x(XR[al & 1], (ai >> 1) + 2, al & 1);
        phase_barrier();
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int u = al * P + p;
            bregs& Bc = (u % 3 == 0) ? B0 : (u % 3 == 1) ? B1 : B2;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                // fragment kk of this tile was requested three tiles ago: 11 weight requests since, plus one genotype
                // request per block boundary crossed on the way (p folds after unrolling)
                if (P == 1) wait_vm<14>();
                else if (P == 2 && p == 0) wait_vm<13>();
                else wait_vm<12>();
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][tm], Bc.b[kk], acc[tm], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                load_b_part(Bc, bb * P + u + 3, kk, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        phase_barrier();
    };
    // whole bodies without a branch inside (the compiler's vmcnt bookkeeping stays exact), then the remainder
    int bb = 0;
    for (; bb + NBLK <= cnt; bb += NBLK) {
        block(bb, std::integral_constant<int, 0>{});
        block(bb, std::integral_constant<int, 1>{});
        if (NBLK == 6) {
            block(bb, std::integral_constant<int, 2>{});
            block(bb, std::integral_constant<int, 3>{});
            block(bb, std::integral_constant<int, 4>{});
            block(bb, std::integral_constant<int, 5>{});
        }
    }
    if (NBLK == 6 && bb < cnt) {                                // cnt is even: 2 or 4 blocks left
        block(bb, std::integral_constant<int, 0>{});
        block(bb, std::integral_constant<int, 1>{});
        if (bb + 2 < cnt) {
            block(bb, std::integral_constant<int, 2>{});
            block(bb, std::integral_constant<int, 3>{});
        }
    }
    // requests past the end (clamped, never used) are still landing: drain them while their registers are still
    // allocated - the empty asm after the wait is what keeps the compiler from re-using one of them before it
    wait_vm<0>();
    asm volatile("" ::"v"(B0.b[0]), "v"(B0.b[1]), "v"(B0.b[2]), "v"(B0.b[3]), "v"(B1.b[0]), "v"(B1.b[1]), "v"(B1.b[2]),
                 "v"(B1.b[3]), "v"(B2.b[0]), "v"(B2.b[1]), "v"(B2.b[2]), "v"(B2.b[3]), "v"(XR[0]), "v"(XR[1]));
    if (grp == 0) phase_barrier();

    // D[i = row][j = unit]: wave tile 128 rows x 32 units through a wave-private LDS image, then 16-byte stores
    float* const ep = reinterpret_cast<float*>(gm_smem) + w * 4096;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) ep[(tm * 32 + rowmap(r, hi)) * 32 + jl] = acc[tm][r];
    wait_lgkm0();
    float* pout = partial + ((int64_t)g * Mp + mt * GM_BM) * GM_HP + w * 32;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = i * 8 + (lane >> 3), c4 = (lane & 7) * 4;
        *reinterpret_cast<f32x4*>(pout + (int64_t)row * GM_HP + c4) = *reinterpret_cast<const f32x4*>(ep + row * 32 + c4);
    }
}

// a1[m][h] = ELU(sum_g partial[g][m][h] + cvec[h] + b1[h]).  Block = 64 float4 positions x 4 quarters of the SNP
// groups: every thread has its G/4 loads in flight at once, the quarters are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void l1_gemm_reduce_kernel(const float* __restrict__ partial, int G, int64_t MH,
                                                             const float* __restrict__ cvec8,
                                                             const float* __restrict__ b1, float* __restrict__ a1) {
    __shared__ f32x4 red[4][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i4 = ((int64_t)blockIdx.x * 64 + o) * 4;
    const int gq = (G + 3) / 4, g0 = q * gq, g1 = g0 + gq < G ? g0 + gq : G;
    f32x4 s = f32x4{0};
    int g = g0;
    for (; g + 8 <= g1; g += 8) {
        f32x4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const f32x4*>(partial + (int64_t)(g + e) * MH + i4);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = s + v[e];
    }
    for (; g < g1; ++g) s = s + *reinterpret_cast<const f32x4*>(partial + (int64_t)g * MH + i4);
    red[q][o] = s;
    __syncthreads();
    if (q == 0) {
        const f32x4 z = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
        const int h = (int)(i4 % GM_HP);
        f32x4 c = *reinterpret_cast<const f32x4*>(cvec8 + h);
#pragma unroll
        for (int sl = 1; sl < 8; ++sl) c = c + *reinterpret_cast<const f32x4*>(cvec8 + sl * GM_HP + h);
        const f32x4 b = *reinterpret_cast<const f32x4*>(b1 + h);
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = elu_f(z[e] + (c[e] + b[e]));
        *reinterpret_cast<f32x4*>(a1 + i4) = out;
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
// 64-SNP blocks of the image, rounded up to whole pairs (a zero-weight tile pads an odd count)
static int gm_nkt64(const loc_dims* d) { return ((d->Kp + GM_BK - 1) / GM_BK + 1) & ~1; }

extern "C" int loc_l1_gemm_supported(int Hp, int pieces) { return Hp == GM_HP && pieces >= 1 && pieces <= 3; }

// image = [cvec8: 8*Hp floats][cpart: nkt64*Hp floats][tiles: nkt64*pieces tiles of 32 KB], 1 KB aligned sections
static int64_t gm_cpart_off(const loc_dims* d) { return 8 * GM_HP * 4; }
static int64_t gm_tiles_off(const loc_dims* d) {
    return (gm_cpart_off(d) + (int64_t)gm_nkt64(d) * GM_HP * 4 + 1023) / 1024 * 1024;
}
extern "C" int64_t loc_l1_image_bytes(const loc_dims* d, int pieces) {
    if (!loc_l1_gemm_supported(d->Hp, pieces)) return 0;
    return gm_tiles_off(d) + (int64_t)gm_nkt64(d) * pieces * GM_BTILE;
}

extern "C" int loc_l1_image_build(const loc_dims* d, const float* scale_shift, const float* w1s, int pieces,
                                  void* image, void* stream) {
    if (!loc_l1_gemm_supported(d->Hp, pieces)) {
        loc_set_error("loc_l1_image_build: width %d / %d pieces unsupported (needs padded width 256)", d->Hp, pieces);
        return -1;
    }
    unsigned char* base = static_cast<unsigned char*>(image);
    float* cvec = reinterpret_cast<float*>(base);
    float* cpart = reinterpret_cast<float*>(base + gm_cpart_off(d));
    unsigned char* tiles = base + gm_tiles_off(d);
    const int nkt = gm_nkt64(d);
    hipStream_t st = (hipStream_t)stream;
    switch (pieces) {
        case 1: hipLaunchKernelGGL(l1_image_kernel<1>, dim3(nkt), dim3(GM_HP), 0, st, w1s, scale_shift, d->Kp, tiles, cpart); break;
        case 2: hipLaunchKernelGGL(l1_image_kernel<2>, dim3(nkt), dim3(GM_HP), 0, st, w1s, scale_shift, d->Kp, tiles, cpart); break;
        default: hipLaunchKernelGGL(l1_image_kernel<3>, dim3(nkt), dim3(GM_HP), 0, st, w1s, scale_shift, d->Kp, tiles, cpart); break;
    }
    LOC_CHECK_LAUNCH();
    return gm_launch_cvec(cpart, nkt, cvec, stream);
}

// shared with l1_gemm_i8.hip (kernels are launched from the translation unit that defines them)
int gm_launch_cvec(const float* cpart, int nkt64, float* cvec8, void* stream) {
    hipLaunchKernelGGL(l1_image_cvec_kernel, dim3(GM_HP / 64, 8), dim3(1024), 0, (hipStream_t)stream, cpart, nkt64, cvec8);
    LOC_CHECK_LAUNCH();
    return 0;
}
int gm_launch_reduce(const float* partial, int G, int64_t MH, const float* cvec8, const float* b1, float* a1,
                     void* stream) {
    hipLaunchKernelGGL(l1_gemm_reduce_kernel, dim3((unsigned)(MH / 256)), dim3(256), 0, (hipStream_t)stream, partial, G,
                       MH, cvec8, b1, a1);
    LOC_CHECK_LAUNCH();
    return 0;
}

// the dynamic-LDS limit is a per-device attribute of the function: set once per (kernel, device)
template <typename F>
static int gm_set_lds(F* func) {
    static bool done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && done[dev]) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(func), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       GM_LDS);
    if (e != hipSuccess) { loc_set_error("hipFuncSetAttribute(%d): %s", GM_LDS, hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < 64) done[dev] = true;
    return 0;
}

extern "C" int loc_l1_forward_gemm(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n, const loc_dims* d,
                                   const void* image, int pieces, const float* b1, float* partial,
                                   int64_t partial_floats, float* a1, int target_blocks, void* stream) {
    if (n < 1) { loc_set_error("loc_l1_forward_gemm: n=%d", n); return -1; }
    if (!loc_l1_gemm_supported(d->Hp, pieces)) {
        loc_set_error("loc_l1_forward_gemm: width %d / %d pieces unsupported", d->Hp, pieces);
        return -1;
    }
    if (d->Kp < 16 || d->Kp % 16 || x_pitch % 16 || x_pitch < d->Kp || ((uintptr_t)X & 15)) {
        loc_set_error("loc_l1_forward_gemm: needs a 16-byte aligned X, Kp %% 16 == 0 and a 16-byte row pitch >= Kp");
        return -1;
    }
    const int nkt = gm_nkt64(d);
    const int n_mt = (n + GM_BM - 1) / GM_BM, Mp = n_mt * GM_BM;
    if (target_blocks < 1) target_blocks = 256;
    int G = target_blocks / n_mt;
    const int64_t cap = partial_floats / ((int64_t)Mp * GM_HP);
    if (G > cap) G = (int)cap;
    if (G > nkt / 2) G = nkt / 2;
    if (G >= 8) G &= ~7;
    if (G < 1) { loc_set_error("loc_l1_forward_gemm: scratch too small for %d rows", n); return -1; }
    const unsigned char* base = static_cast<const unsigned char*>(image);
    const float* cvec = reinterpret_cast<const float*>(base);
    const unsigned char* tiles = base + gm_tiles_off(d);
    hipStream_t st = (hipStream_t)stream;
#define GM_LAUNCH(PP)                                                                                          \
    {                                                                                                          \
        int rc = gm_set_lds(l1_gemm_kernel<PP>);                                                               \
        if (rc) return rc;                                                                                     \
        hipLaunchKernelGGL(l1_gemm_kernel<PP>, dim3(n_mt * G), dim3(GM_NT), GM_LDS, st, X, x_pitch, rows, n,   \
                           d->Kp, tiles, partial, G, n_mt, nkt / 2);                                           \
    }
    switch (pieces) {
        case 1: GM_LAUNCH(1) break;
        case 2: GM_LAUNCH(2) break;
        default: GM_LAUNCH(3) break;
    }
#undef GM_LAUNCH
    LOC_CHECK_LAUNCH();
    return gm_launch_reduce(partial, G, (int64_t)Mp * GM_HP, cvec, b1, a1, stream);
}
