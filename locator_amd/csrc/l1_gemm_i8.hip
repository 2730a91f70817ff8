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
//   l1_colmax_kernel     per-unit max_k |s_k W1[k][h]| (order-independent: max over non-negative floats as uints)
//   l1_image_i8_kernel   W1S (fp32, swizzled) x BN scale -> HBM image of digit planes, one 16 KB tile per (64-SNP block,
//                        digit plane), laid out [16-SNP chunk c][unit n][16 SNPs]: a wave's MFMA B operand for one
//                        32-SNP step is two contiguous 512-byte runs.  Plane 0 is the most significant digit.  Also
//                        the shift term's per-block partial sums (shared with l1_gemm.hip's cvec reduction).
//   l1_gemm_i8_kernel    workgroup = 8 waves on a 128-row x 256-unit tile, split over SNP groups exactly like
//                        l1_gemm_kernel: a wave owns 32 units and ALL 128 rows, its digit fragments go HBM/L2 -> VGPRs
//                        (a ring of 6 tiles = 3 blocks ahead), only the raw genotype block is shared through a 4-slot
//                        LDS ring (8 KB per block, XOR-placed so that the 8-lane store groups and the 16-lane read
//                        groups are bank-conflict free).  The two waves of each SIMD run half a block apart between
//                        two barriers per block: one issues the block's 8 MFMAs per digit from registers while its
//                        partner reads the next block's 8 A fragments, stores its 16 genotype bytes and requests.
//                        There is NO vector-ALU work in the loop.  Two digits accumulate side by side (2 x 64
//                        accumulator registers); the exact mode folds them into fp32 and walks the K range a second
//                        time for the least significant plane (genotypes re-read from L2, 8 MFMAs per block).
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
#define G8_LDS 131072             /* 4 genotype images in the loop; the epilogue stages 8 x 16 KB of partials */
#define G8_RING 6                 /* digit tiles in flight per wave (48 registers)  */


// ---------------------------------------------------------------------------------------------------------
// per-unit scale and digit image
// ---------------------------------------------------------------------------------------------------------
// One workgroup per 64-SNP block, thread = unit n (the addressing of l1_image_kernel).  colmax[n] = max |s_k W1[k][n]|
// as the bit pattern of a non-negative float: atomicMax on it is order-independent, so the result is deterministic.
__global__ __launch_bounds__(G8_HP) void l1_colmax_kernel(const float* __restrict__ w1s, const float* __restrict__ ss4,
                                                          int Kp, uint32_t* __restrict__ colmax) {
    constexpr int nht = G8_HP / 32;
    const int kt64 = blockIdx.x, n = threadIdx.x;
    const int ht = n >> 5, hl = n & 31, q = hl >> 3, hi = (hl >> 2) & 1, c4 = hl & 3;
    float mx = 0.f;
#pragma unroll
    for (int h32 = 0; h32 < 2; ++h32) {
        const int kt32 = 2 * kt64 + h32;
        if (kt32 * 32 < Kp) {
            const float* src = w1s + ((int64_t)(kt32 * nht + ht) * 4 + q) * 256 + hi * 128 + c4;
#pragma unroll
            for (int kl = 0; kl < 32; ++kl) mx = fmaxf(mx, fabsf(src[kl * 4] * ss4[kt32 * 32 + kl]));
        }
    }
    if (mx > 0.f) atomicMax(colmax + n, fbits(mx));
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
// k = kt64*64 + c*16 + e;  cpart[kt64][n] = sum over the block's SNPs of t_k W1[k][n];  delta[n] written by block 0.
template <int DT>
__global__ __launch_bounds__(G8_HP) void l1_image_i8_kernel(const float* __restrict__ w1s, const float* __restrict__ ss4,
                                                            int Kp, const uint32_t* __restrict__ colmax,
                                                            float* __restrict__ delta, unsigned char* __restrict__ tiles,
                                                            float* __restrict__ cpart) {
    constexpr int nht = G8_HP / 32;
    const int kt64 = blockIdx.x, n = threadIdx.x;
    const int ht = n >> 5, hl = n & 31, q = hl >> 3, hi = (hl >> 2) & 1, c4 = hl & 3;
    const float* scale = ss4;
    const float* shift = ss4 + Kp;
    const float dl = digit_delta<DT>(bitsf(colmax[n]));
    const float inv = 1.0f / dl;                          // a power of two: exact
    if (kt64 == 0) delta[n] = dl;
    float csum = 0.f;
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
                csum = fmaf(shift[k], w, csum);
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
    cpart[(int64_t)kt64 * G8_HP + n] = csum;
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
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\t"
                 "ds_read_b128 %2, %4 offset:1024\n\tds_read_b128 %3, %4 offset:1536"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr) : "memory");
}
// Global loads as asm with hand-counted s_waitcnt vmcnt (see l1_gemm.hip: the compiler's own counts collapse at the
// loop header).  COUNT TABLE, per wave, requests in program order:
//   load phase of a block     1 genotype request (16 bytes per lane)
//   matrix phase of a block   D digit tiles x 2 fragment requests, each issued right after the 4 MFMAs that consumed
//                             the fragment it replaces, G8_RING tiles ahead
//   => a fragment about to be consumed was requested G8_RING tiles ago; younger than it are 2*G8_RING - 1 fragment
//      requests and the G8_RING / D genotype requests of the blocks crossed:      vmcnt(2*G8_RING - 1 + G8_RING/D)
//   => the genotype bytes stored in a load phase were requested two blocks ago; younger than them are the 2*2*D fragment
//      requests of two matrix phases and one genotype request:                    vmcnt(4*D + 1)
template <typename T>
__device__ __forceinline__ void gload16_i8(T& r, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_i8() {
#ifdef LOC_GEMM_DEBUG_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // parity-debug build: every count replaced by a full drain
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void phase_barrier_i8() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

struct g8_ctx {
    const uint8_t* xsrc[2];
    const unsigned char* tiles;
    unsigned char* As;
    int g, G, cntp, Kp, q, xr, w, jl, hi;
};

// One walk over this workgroup's SNP blocks with D digit planes [plane0, plane0 + D) of the DT planes in the image.
template <int D, int DT>
__device__ __forceinline__ void g8_sweep(const g8_ctx& c, int plane0, i32x16 (&acc)[D][4]) {
    static_assert(G8_RING % D == 0 && (6 * D) % G8_RING == 0, "ring / unroll shapes");
    const int cnt = 2 * c.cntp, nT = cnt * D;
    auto load_x = [&](u32x4& R, int pc, int i) {
        const int cc = pc < c.cntp ? pc : c.cntp - 1;
        int koff = (c.g + cc * c.G) * (2 * G8_BK) + 16 * c.q;
        if (koff > c.Kp - 16) koff = c.Kp - 16;            // only in the zero-weight padding of the last pair
        gload16_i8(R, c.xsrc[i] + koff);
    };
    // 16 bytes = chunk c4 = q & 3 of block e = q >> 2 of the pair, row xr + 64 i: slot (2 pc + e) & 3, at
    // c4*2048 + ((row ^ q) << 4).  The 8 lanes of a store group (one row, q = 0..7) land on 8 different 16-byte
    // bank groups, and so do the 16 lanes of a read group (one chunk, 16 rows that differ in their low 4 bits).
    const uint32_t woff = (c.q >> 2) * G8_AIMG + (c.q & 3) * 2048;
    auto stage = [&](const u32x4& R, int pc, int i) {
        const int row = c.xr + 64 * i;
        unsigned char* ad = c.As + woff + ((2 * pc) & 3) * G8_AIMG + ((row ^ c.q) << 4);
        *reinterpret_cast<u32x4*>(ad) = R;
    };
    const int b_lane = c.hi * 4096 + (c.w * 32 + c.jl) * 16;
    auto load_b = [&](i32x4& R, int j, int kk) {           // fragment kk of digit tile j = block * D + p
        const int jj = j < nT ? j : nT - 1;
        const int a = jj / D, p = jj - a * D;
        const int kt = 2 * (c.g + (a >> 1) * c.G) + (a & 1);
        gload16_i8(R, c.tiles + ((int64_t)kt * DT + plane0 + p) * G8_TILE + b_lane + kk * 8192);
    };
    uint32_t aoff[2][2];                                    // A-fragment addresses for even / odd blocks
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            aoff[e][kk] = lds_addr32_i8(c.As) + (2 * kk + c.hi) * 2048 + ((c.jl ^ (4 * e + 2 * kk + c.hi)) << 4);

    // prologue: pair 0 stored at once, pair 1 and digit tiles 0..RING-1 requested
    u32x4 XR[2];
    i32x4 B[G8_RING][2];
    {
        u32x4 x0, x1;
        load_x(x0, 0, 0);
        load_x(x1, 0, 1);
        load_x(XR[0], 1, 0);
        load_x(XR[1], 1, 1);
#pragma unroll
        for (int j = 0; j < G8_RING; ++j) { load_b(B[j][0], j, 0); load_b(B[j][1], j, 1); }
        wait_vm_i8<0>();
        stage(x0, 0, 0);
        stage(x1, 0, 1);
    }
    __syncthreads();

    const int grp = c.w >> 2;
    if (grp == 1) phase_barrier_i8();
    auto block = [&](int bb, auto alc) {
        constexpr int al = decltype(alc)::value;
        const int ai = bb + al;
        const uint32_t so = (ai & 3) * G8_AIMG;
        i32x4 a[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) rd4_i8(a[kk][0], a[kk][1], a[kk][2], a[kk][3], aoff[al & 1][kk] + so);
        wait_vm_i8<4 * D + 1>();
        stage(XR[al & 1], (ai >> 1) + 1, al & 1);
        load_x(XR[al & 1], (ai >> 1) + 2, al & 1);
        phase_barrier_i8();
#pragma unroll
        for (int p = 0; p < D; ++p) {
            const int u = al * D + p;                       // static after unrolling
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                wait_vm_i8<2 * G8_RING - 1 + G8_RING / D>();
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[p][tm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kk][tm], B[u % G8_RING][kk], acc[p][tm], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                load_b(B[u % G8_RING][kk], bb * D + u + G8_RING, kk);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        phase_barrier_i8();
    };
    // whole bodies of 6 blocks (ring slot and block parity static), then the even remainder
    int bb = 0;
    for (; bb + 6 <= cnt; bb += 6) {
        block(bb, std::integral_constant<int, 0>{});
        block(bb, std::integral_constant<int, 1>{});
        block(bb, std::integral_constant<int, 2>{});
        block(bb, std::integral_constant<int, 3>{});
        block(bb, std::integral_constant<int, 4>{});
        block(bb, std::integral_constant<int, 5>{});
    }
    if (bb < cnt) {                                         // cnt is even: 2 or 4 blocks left
        block(bb, std::integral_constant<int, 0>{});
        block(bb, std::integral_constant<int, 1>{});
        if (bb + 2 < cnt) {
            block(bb, std::integral_constant<int, 2>{});
            block(bb, std::integral_constant<int, 3>{});
        }
    }
    // requests past the end (clamped, never used) are still landing: drain them while their registers are allocated
    wait_vm_i8<0>();
#pragma unroll
    for (int j = 0; j < G8_RING; ++j) asm volatile("" ::"v"(B[j][0]), "v"(B[j][1]));
    asm volatile("" ::"v"(XR[0]), "v"(XR[1]));
    if (grp == 0) phase_barrier_i8();
}

template <int DT>
__global__ __launch_bounds__(G8_NT) void l1_gemm_i8_kernel(const uint8_t* __restrict__ X, int64_t pitch,
                                                            const int32_t* __restrict__ rows, int n, int Kp,
                                                            const unsigned char* __restrict__ tiles,
                                                            const float* __restrict__ delta,
                                                            float* __restrict__ partial, int G, int n_mt, int npairs) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char g8_smem[];
    const int t = threadIdx.x, lane = t & 63;
    g8_ctx c;
    c.w = __builtin_amdgcn_readfirstlane(t >> 6);
    c.jl = lane & 31;
    c.hi = lane >> 5;
    int mt;
    if ((G & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        c.g = xcd + 8 * (idx / n_mt);
    } else {
        c.g = blockIdx.x % G;
        mt = blockIdx.x / G;
    }
    c.G = G;
    c.Kp = Kp;
    c.tiles = tiles;
    c.As = g8_smem;
    c.cntp = (npairs - c.g + G - 1) / G;
    c.xr = t >> 3;
    c.q = t & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = mt * G8_BM + c.xr + 64 * i;
        if (r > n - 1) r = n - 1;
        c.xsrc[i] = X + (int64_t)rows[r] * pitch;
    }
    const int Mp = n_mt * G8_BM;

    i32x16 acc[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[p][tm] = i32x16{0};
    g8_sweep<2, DT>(c, 0, acc);

    // D[i = row][j = unit] x delta_j: wave tile 128 rows x 32 units through a wave-private LDS image, then 16-byte stores.
    // Three planes: the two leading ones go out first (65536 * plane 0 + 256 * plane 1, in units of delta), the K range
    // is walked again for the least significant plane, and the same thread adds it to what it stored - delta is a power
    // of two, so the sum equals the one a 192-register accumulator set would have produced, without the spills that
    // set costs at 256 registers per wave.
    const float dl = delta[c.w * 32 + c.jl];
    float* const ep = reinterpret_cast<float*>(g8_smem) + c.w * 4096;
    float* const pout = partial + ((int64_t)c.g * Mp + mt * G8_BM) * G8_HP + c.w * 32;
    auto emit = [&](const i32x16 (&hi_p)[4], const i32x16 (&lo_p)[4], float s_hi, float s_lo, bool add) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[(tm * 32 + rowmap(r, c.hi)) * 32 + c.jl] = (s_hi * (float)hi_p[tm][r] + s_lo * (float)lo_p[tm][r]) * dl;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = i * 8 + (lane >> 3), c4 = (lane & 7) * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * 32 + c4);
            f32x4* dst = reinterpret_cast<f32x4*>(pout + (int64_t)row * G8_HP + c4);
            if (add) v = v + *dst;
            *dst = v;
        }
    };
    if (DT == 2) {
        emit(acc[0], acc[1], 256.f, 1.f, false);
    } else {
        emit(acc[0], acc[1], 65536.f, 256.f, false);
        __syncthreads();                                    // the staging images overlap the genotype ring
        i32x16 lo[1][4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) lo[0][tm] = i32x16{0};
        g8_sweep<1, DT>(c, 2, lo);
        emit(lo[0], lo[0], 1.f, 0.f, true);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static int g8_nkt64(const loc_dims* d) { return ((d->Kp + G8_BK - 1) / G8_BK + 1) & ~1; }

extern "C" int loc_l1_gemm_i8_supported(int Hp, int digits) { return Hp == G8_HP && (digits == 2 || digits == 3); }

// image = [cvec8: 8*Hp floats][delta: Hp floats][colmax: Hp uints][cpart: nkt64*Hp floats][tiles], 1 KB aligned sections
static int64_t g8_delta_off() { return 8 * G8_HP * 4; }
static int64_t g8_colmax_off() { return g8_delta_off() + G8_HP * 4; }
static int64_t g8_cpart_off() { return g8_colmax_off() + G8_HP * 4; }
static int64_t g8_tiles_off(const loc_dims* d) {
    return (g8_cpart_off() + (int64_t)g8_nkt64(d) * G8_HP * 4 + 1023) / 1024 * 1024;
}
extern "C" int64_t loc_l1_image_i8_bytes(const loc_dims* d, int digits) {
    if (!loc_l1_gemm_i8_supported(d->Hp, digits)) return 0;
    return g8_tiles_off(d) + (int64_t)g8_nkt64(d) * digits * G8_TILE;
}

extern "C" int loc_l1_image_i8_build(const loc_dims* d, const float* scale_shift, const float* w1s, int digits,
                                     void* image, void* stream) {
    if (!loc_l1_gemm_i8_supported(d->Hp, digits)) {
        loc_set_error("loc_l1_image_i8_build: width %d / %d digits unsupported (needs padded width 256, 2 or 3 digits)",
                      d->Hp, digits);
        return -1;
    }
    unsigned char* base = static_cast<unsigned char*>(image);
    float* cvec = reinterpret_cast<float*>(base);
    float* delta = reinterpret_cast<float*>(base + g8_delta_off());
    uint32_t* colmax = reinterpret_cast<uint32_t*>(base + g8_colmax_off());
    float* cpart = reinterpret_cast<float*>(base + g8_cpart_off());
    unsigned char* tiles = base + g8_tiles_off(d);
    const int nkt = g8_nkt64(d);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(colmax, 0, G8_HP * 4, st);
    if (e != hipSuccess) { loc_set_error("loc_l1_image_i8_build: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(l1_colmax_kernel, dim3(nkt), dim3(G8_HP), 0, st, w1s, scale_shift, d->Kp, colmax);
    LOC_CHECK_LAUNCH();
    if (digits == 2)
        hipLaunchKernelGGL(l1_image_i8_kernel<2>, dim3(nkt), dim3(G8_HP), 0, st, w1s, scale_shift, d->Kp, colmax, delta, tiles, cpart);
    else
        hipLaunchKernelGGL(l1_image_i8_kernel<3>, dim3(nkt), dim3(G8_HP), 0, st, w1s, scale_shift, d->Kp, colmax, delta, tiles, cpart);
    LOC_CHECK_LAUNCH();
    return gm_launch_cvec(cpart, nkt, cvec, stream);
}

extern "C" int loc_l1_forward_gemm_i8(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n, const loc_dims* d,
                                      const void* image, int digits, int x_max, const float* b1, float* partial,
                                      int64_t partial_floats, float* a1, int target_blocks, void* stream) {
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
    if (d->Kp < 16 || d->Kp % 16 || x_pitch % 16 || x_pitch < d->Kp || ((uintptr_t)X & 15)) {
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
    if (G >= 8) G &= ~7;
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
    if (digits == 2) {
        LOC_ENSURE_LDS((l1_gemm_i8_kernel<2>), G8_LDS);
        hipLaunchKernelGGL(l1_gemm_i8_kernel<2>, dim3(n_mt * G), dim3(G8_NT), G8_LDS, st, X, x_pitch, rows, n, d->Kp,
                           tiles, delta, partial, G, n_mt, nkt / 2);
    } else {
        LOC_ENSURE_LDS((l1_gemm_i8_kernel<3>), G8_LDS);
        hipLaunchKernelGGL(l1_gemm_i8_kernel<3>, dim3(n_mt * G), dim3(G8_NT), G8_LDS, st, X, x_pitch, rows, n, d->Kp,
                           tiles, delta, partial, G, n_mt, nkt / 2);
    }
    LOC_CHECK_LAUNCH();
    return gm_launch_reduce(partial, G, (int64_t)Mp * G8_HP, cvec, b1, a1, stream);
}
