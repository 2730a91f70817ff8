// Large-M first-layer genotype GEMM on the bf16 matrix pipe, weights converted ONCE per sweep
// (reference: model.predict on predgen / testgen and the --jacknife replicate predictions,
// /root/reference/locator/locator.py:414, :441, :683-747).
//
//     z1[m][h] = sum_k x[m][k] (s_k W1[k][h])  +  sum_k t_k W1[k][h]          (BatchNorm in inference form)
//
// l1_rows.hip converts the fp32 weight tile (scale, split into bf16 pieces, pack) inside the K loop of EVERY
// 128-row tile, which made it vector-ALU bound (PMC, round 1: matrix pipe 43 % busy, VALU 28 %, waits 31 %).
// Here the conversion is its own streaming pass:
//
//   l1_image_kernel      W1S (fp32, swizzled) x BN scale  ->  HBM image of bf16 tiles, one 32 KB tile per
//                        (64-SNP block, piece), stored EXACTLY as the GEMM wants it in the LDS
//                        ([8-SNP chunk c][unit n][8 SNPs]: the MFMA B-operand read of a 32-unit group is 512
//                        contiguous bytes, conflict-free, no swizzle), plus the per-unit shift term
//                        c[h] = sum_k t_k W1[k][h] (per-block partial sums, then one fixed-order reduction).
//                        3 pieces = the exact 8+8+8-bit truncation split of l1_rows.hip (fp32-exact products).
//   l1_gemm_kernel       pure matrix-pipe K loop: weight tiles go HBM/L2 -> LDS by global_load_lds (no VGPRs, no
//                        VALU), genotypes u8 -> LDS (raw, global_load_lds) -> bf16 image (the only VALU work:
//                        1 cvt + 1/2 perm per genotype, once per workgroup and 64-SNP block), MFMA
//                        32x32x16 bf16 from LDS fragments.  Workgroup = 8 waves on a 128-row x 256-unit tile
//                        (wave = 64 x 64), split over SNP blocks (strided) like l1_rows; raw s_barrier and
//                        counted vmcnt so two weight tiles and three genotype tiles stay in flight across the
//                        barriers; 160 KB LDS = 3 weight slots + 2 genotype images + 4 raw genotype slots.
//   l1_gemm_reduce_kernel  fixed-order sum of the SNP-group partials + shift term + b1, ELU.
//
// Blocks that share a SNP group (same weight tiles, different row tiles) are placed on one XCD (block b runs on
// XCD b % 8), so a weight tile leaves HBM once and the other row tiles hit that XCD's L2.  Placement is a speed
// hint only.
#include "common.h"

#define GM_BM 128
#define GM_BK 64
#define GM_NT 512
#define GM_HP 256
#define GM_BTILE (GM_HP * GM_BK * 2) /* 32768: one (SNP block, piece) weight tile, bf16 */
#define GM_AIMG (GM_BM * GM_BK * 2)  /* 16384: bf16 genotype image of one SNP block    */
#define GM_ARAW (GM_BM * GM_BK)      /*  8192: the same block as uint8                 */
#define GM_NB 3
#define GM_NR 4
#define GM_LDS (GM_NB * GM_BTILE + 2 * GM_AIMG + GM_NR * GM_ARAW) /* 163840 = 160 KB */

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* gbl_vp;

// 16 bytes per lane, global -> LDS at (wave-uniform) lds_base + lane*16; tracked by vmcnt
__device__ __forceinline__ void glds16(const void* g, void* lds_base) {
    __builtin_amdgcn_global_load_lds((gbl_vp)g, (lds_vp)lds_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// timing probe of the experiment variants (V & 128): workgroup 0 stamps s_memtime per phase into gm_dbg[wave][slot]
__device__ unsigned long long gm_dbg[8 * 1024];
template <int V>
__device__ __forceinline__ void gm_stamp(int w, int& slot) {
    if ((V & 128) && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && slot < 1024) gm_dbg[w * 1024 + slot++] = __builtin_amdgcn_s_memtime();
}
extern "C" int loc_l1_gemm_debug_read(unsigned long long* h_out) {
    return (int)hipMemcpyFromSymbol(h_out, HIP_SYMBOL(gm_dbg), sizeof(unsigned long long) * 8 * 1024);
}


// Fragment reads under manual control (the compiler folds a source-level double buffer back into
// "4 reads -> lgkmcnt(0) -> 4 MFMAs" per k-step, exposing the LDS latency four times per tile: 1,200 cycles for
// 16 MFMAs measured).  One k-step = 2 genotype + 2 weight fragments of 16 bytes per lane.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
template <int OB>
__device__ __forceinline__ void rd_frags(bf16x8& a0, bf16x8& a1, bf16x8& b0, bf16x8& b1, uint32_t aa, uint32_t ba) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\t"
                 "ds_read_b128 %2, %5 offset:%6\n\tds_read_b128 %3, %5 offset:%7"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1)
                 : "v"(aa), "v"(ba), "n"(OB), "n"(OB + 512)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);      // MFMAs must not be hoisted above the wait (they are not memory operations)
}

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
// V: experiment flags (0 = product).  1: pin the fragment reads of k-step kk+1 ahead of the MFMAs of kk;
// ablations (wrong results, timing only): 2 skip the genotype conversion, 4 skip the weight-tile loads,
// 8 skip the fragment reads, 16 skip the MFMAs.
template <int P, int V>
__global__ __launch_bounds__(GM_NT) void l1_gemm_kernel(const uint8_t* __restrict__ X, int64_t pitch,
                                                        const int32_t* __restrict__ rows, int n, int Kp,
                                                        const unsigned char* __restrict__ tiles,
                                                        float* __restrict__ partial, int G, int n_mt, int nkt64) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char gm_smem[];
    unsigned char* const Bs = gm_smem;
    unsigned char* const As = gm_smem + GM_NB * GM_BTILE;
    unsigned char* const Rs = As + 2 * GM_AIMG;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int jl = lane & 31, hi = lane >> 5;
    int g, mt;
    if ((G & 15) == 0) {
        // SNP groups 2x, 2x+1 (mod 16) on XCD x with all of their row tiles: the weight tiles of a group leave HBM
        // once, and the two groups read the two 64-byte halves of the same 128-byte genotype lines at about the same
        // time, so each line is fetched into that XCD's L2 once (groups g, g+1 on different XCDs fetched it twice:
        // 40 us for the genotype stream alone)
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        const int r = idx / n_mt;
        g = 2 * xcd + (r & 1) + 16 * (r >> 1);
    } else if ((G & 7) == 0) {     // SNP group g on XCD g % 8 with all of its row tiles
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        g = xcd + 8 * (idx / n_mt);
    } else {
        g = blockIdx.x % G;
        mt = blockIdx.x / G;
    }
    const int Mp = n_mt * GM_BM;
    const int cnt = (nkt64 - g + G - 1) / G;       // SNP blocks g, g+G, ... of this workgroup (>= 1)
    const int nB = cnt * P;

    // genotype staging role: row xm of the tile, bytes 16*xj .. of the 64-byte block (4 lanes = one 64-byte run)
    const int xm = t >> 2, xj = t & 3;
    int xrow_i = mt * GM_BM + xm;
    if (xrow_i > n - 1) xrow_i = n - 1;            // padded rows repeat the last one (finite values, never read back)
    const uint8_t* const xsrc = X + (int64_t)rows[xrow_i] * pitch;
    auto issue_raw = [&](int ai, int slot) {       // tile ai of this workgroup -> raw slot
        int a = ai < cnt ? ai : cnt - 1;           // past the end: a valid tile into a dead slot (static vmcnt counts)
        if (V & 32) a = 0;
        int koff = (g + a * G) * GM_BK + 16 * xj;
        if (koff > Kp - 16) koff = Kp - 16;        // Kp % 64 == 32: the image holds zeros there, any genotype will do
        glds16(xsrc + koff, Rs + slot * GM_ARAW + w * 1024);
    };
    auto issue_b = [&](int j, int slot) {          // weight tile j = (SNP block j / P, piece j % P) -> B slot
        int jj = j < nB ? j : nB - 1;
        if (V & 32) jj = 0;
        const int kt = g + (jj / P) * G, p = jj % P;
        const unsigned char* src = tiles + ((int64_t)kt * P + p) * GM_BTILE + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(src + (i * 8 + w) * 1024, Bs + slot * GM_BTILE + (i * 8 + w) * 1024);
    };
    // raw slot -> bf16 image: this thread's own 16 bytes (so only its own vmcnt orders the read), two 16-byte
    // chunks c = 2 xj, 2 xj + 1 of row xm at  c*2048 + ((xm ^ (c & 6)) * 16)  (conflict-free writes and reads)
    auto convert = [&](int rslot, int islot) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(Rs + rslot * GM_ARAW + t * 16);
        u32x4 o0, o1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t b = raw[d];
            const uint32_t lo = pack_top16(fbits((float)(b & 255u)), fbits((float)((b >> 8) & 255u)));
            const uint32_t hh = pack_top16(fbits((float)((b >> 16) & 255u)), fbits((float)(b >> 24)));
            if (d < 2) { o0[2 * d] = lo; o0[2 * d + 1] = hh; }
            else { o1[2 * (d - 2)] = lo; o1[2 * (d - 2) + 1] = hh; }
        }
        unsigned char* dst = As + islot * GM_AIMG + ((xm ^ (2 * xj)) << 4);
        *reinterpret_cast<u32x4*>(dst + (2 * xj) * 2048) = o0;
        *reinterpret_cast<u32x4*>(dst + (2 * xj + 1) * 2048) = o1;
    };

    const int wm = w & 1, wn = w >> 1;             // 2 row halves x 4 unit quarters
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = f32x16{0};

    const int a_row = wm * 64 + jl;                // + 32 tm
    const int b_lane = hi * 4096 + (wn * 64 + jl) * 16;     // + kk*8192 + tn*512

    auto mma_tile = [&](const unsigned char* Ab, const unsigned char* Bb) {
        bf16x8 a[2][2], b[2][2];
        auto rd = [&](int kk, int s) {
            const int ao = (2 * kk + hi) * 2048 + ((a_row ^ (2 * kk)) << 4);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) a[s][tm] = *reinterpret_cast<const bf16x8*>(Ab + ao + tm * 512);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) b[s][tn] = *reinterpret_cast<const bf16x8*>(Bb + b_lane + kk * 8192 + tn * 512);
        };
        if (V & 8) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) { a[s][i] = bf16x8{0}; b[s][i] = bf16x8{0}; asm volatile("" : "+v"(a[s][i]), "+v"(b[s][i])); }
        } else rd(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (!(V & 8) && kk + 1 < 4) rd(kk + 1, (kk + 1) & 1);
            if (V & 1) __builtin_amdgcn_sched_barrier(0);
            if (!(V & 16)) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1][tm], b[kk & 1][tn], acc[tm][tn], 0, 0, 0);
            } else {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) asm volatile("" :: "v"(a[kk & 1][tm]), "v"(b[kk & 1][tm]));
            }
            if (V & 1) __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue = pseudo-iteration -1: three raw genotype tiles and two weight tiles in flight, tile 0 converted
    issue_raw(0, 0);
    issue_raw(1, 1);
    issue_raw(2, 2);
    issue_b(0, 0);
    issue_b(1, 1);
    wait_vm<10>();                    // raw tile 0 landed (behind it: 2 raw + 8 weight pieces)
    convert(0, 0);
    issue_raw(3, 3);
    wait_vm<5>();                     // weight tile 0 landed (behind it: weight tile 1 + raw tile 3)
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();

    int bs = 0;                       // B slot of tile j
    for (int ai = 0; ai < cnt; ++ai) {
        const unsigned char* Ab = As + (ai & 1) * GM_AIMG;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int j = ai * P + p;
            int bs2 = bs + 2; if (bs2 >= GM_NB) bs2 -= GM_NB;
            if (!(V & 4)) issue_b(j + 2, bs2);                // slot of tile j-1: every wave is past its reads
            if (p == P - 1) {
                // genotype tile ai+1 for the next iteration: its raw bytes were requested three tiles ago
                wait_vm<10 + 4 * P>();
                if (!(V & 2)) convert((ai + 1) & 3, (ai + 1) & 1);
                issue_raw(ai + 4, ai & 3);
            }
            mma_tile(Ab, Bs + bs * GM_BTILE);
            // weight tile j+1 landed; behind it: this iteration's 4 pieces and the raw tiles requested at the end of
            // the previous / this iteration (when those were the last piece of their genotype tile)
            // (p is a constant after unrolling: the untaken waits fold away)
            if (p == 0 && p == P - 1) wait_vm<6>();
            else if (p == 0 || p == P - 1) wait_vm<5>();
            else wait_vm<4>();
            wait_lgkm0();
            __builtin_amdgcn_s_barrier();
            bs = bs + 1; if (bs >= GM_NB) bs -= GM_NB;
        }
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();          // every wave is past its last LDS read and DMA: the LDS is free

    // D[i = row][j = unit]: lane holds unit jl of its tile, rows rowmap(r, hi).  Through a wave-private 64 x 64 fp32
    // LDS image so the partial tile leaves as 16-byte stores (16 per lane instead of 64 dword stores)
    float* const ep = reinterpret_cast<float*>(gm_smem) + w * 4096;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) ep[(tm * 32 + rowmap(r, hi)) * 64 + tn * 32 + jl] = acc[tm][tn][r];
    wait_lgkm0();
    float* pout = partial + ((int64_t)g * Mp + mt * GM_BM + wm * 64) * GM_HP + wn * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = i * 4 + (lane >> 4), c4 = (lane & 15) * 4;
        *reinterpret_cast<f32x4*>(pout + (int64_t)row * GM_HP + c4) = *reinterpret_cast<const f32x4*>(ep + row * 64 + c4);
    }
}

// ---- two-group schedule ------------------------------------------------------------------------------------
// Same tiles, LDS layout and arithmetic as l1_gemm_kernel, but the eight waves run as two groups of four (one
// wave of each group per SIMD) half an iteration apart, with a barrier between the halves:
//
//   phase 2j    group 0: widen genotype tile (first piece of a SNP block only)   group 1: MFMAs of tile j-1, wait B(j)
//   phase 2j+1  group 0: MFMAs of tile j                                         group 1: request weight tile j+2
//
// so on every SIMD one wave is in its matrix segment while its partner issues LDS-DMA / does the vector work,
// instead of all eight doing the same thing at the same time (measured on the lockstep kernel: genotype widening,
// weight requests and fragment reads each ADD to the matrix time).  Group 0 owns the genotype stream (its vmcnt
// queue holds only raw genotype requests: three SNP blocks ahead), group 1 the weight stream (two tiles ahead).
template <int P, int V>
__global__ __launch_bounds__(GM_NT) void l1_gemm2_kernel(const uint8_t* __restrict__ X, int64_t pitch,
                                                         const int32_t* __restrict__ rows, int n, int Kp,
                                                         const unsigned char* __restrict__ tiles,
                                                         float* __restrict__ partial, int G, int n_mt, int nkt64) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char gm_smem[];
    unsigned char* const Bs = gm_smem;
    unsigned char* const As = gm_smem + GM_NB * GM_BTILE;
    unsigned char* const Rs = As + 2 * GM_AIMG;

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = w >> 2, wl = w & 3, tl = t & 255;
    const int jl = lane & 31, hi = lane >> 5;
    int g, mt;
    if ((G & 15) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        const int r = idx / n_mt;
        g = 2 * xcd + (r & 1) + 16 * (r >> 1);
    } else if ((G & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        g = xcd + 8 * (idx / n_mt);
    } else {
        g = blockIdx.x % G;
        mt = blockIdx.x / G;
    }
    const int Mp = n_mt * GM_BM;
    const int cnt = (nkt64 - g + G - 1) / G;
    const int nB = cnt * P;

    // group 0: rows xm, xm + 64 of the tile, bytes 16*xj .. of the 64-byte block; a thread widens the bytes it requested
    const int xm = tl >> 2, xj = tl & 3;
    int r0 = mt * GM_BM + xm, r1 = r0 + 64;
    if (r0 > n - 1) r0 = n - 1;
    if (r1 > n - 1) r1 = n - 1;
    const uint8_t* const xsrc0 = X + (int64_t)rows[r0] * pitch;
    const uint8_t* const xsrc1 = X + (int64_t)rows[r1] * pitch;
    auto issue_raw = [&](int ai, int slot) {
        int a = ai < cnt ? ai : cnt - 1;
        if (V & 32) a = 0;
        int koff = (g + a * G) * GM_BK + 16 * xj;
        if (koff > Kp - 16) koff = Kp - 16;
        glds16(xsrc0 + koff, Rs + slot * GM_ARAW + wl * 1024);
        glds16(xsrc1 + koff, Rs + slot * GM_ARAW + 4096 + wl * 1024);
    };
    auto convert = [&](int rslot, int islot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(Rs + rslot * GM_ARAW + i * 4096 + tl * 16);
            u32x4 o0, o1;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t b = raw[d];
                const uint32_t lo = pack_top16(fbits((float)(b & 255u)), fbits((float)((b >> 8) & 255u)));
                const uint32_t hh = pack_top16(fbits((float)((b >> 16) & 255u)), fbits((float)(b >> 24)));
                if (d < 2) { o0[2 * d] = lo; o0[2 * d + 1] = hh; }
                else { o1[2 * (d - 2)] = lo; o1[2 * (d - 2) + 1] = hh; }
            }
            const int m = xm + 64 * i;
            unsigned char* dst = As + islot * GM_AIMG + ((m ^ (2 * xj)) << 4);
            *reinterpret_cast<u32x4*>(dst + (2 * xj) * 2048) = o0;
            *reinterpret_cast<u32x4*>(dst + (2 * xj + 1) * 2048) = o1;
        }
    };
    // group 1: the whole 32 KB weight tile, 8 pieces of 1 KB per wave
    auto issue_b = [&](int j, int slot) {
        int jj = j < nB ? j : nB - 1;
        if (V & 32) jj = 0;
        const int kt = g + (jj / P) * G, p = jj % P;
        const unsigned char* src = tiles + ((int64_t)kt * P + p) * GM_BTILE + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) glds16(src + (i * 4 + wl) * 1024, Bs + slot * GM_BTILE + (i * 4 + wl) * 1024);
    };

    const int wm = w & 1, wn = w >> 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = f32x16{0};
    const int a_row = wm * 64 + jl;
    const int b_lane = hi * 4096 + (wn * 64 + jl) * 16;

    int ds2 = 0;
    uint32_t aoff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) aoff[kk] = lds_addr(As) + (2 * kk + hi) * 2048 + ((a_row ^ (2 * kk)) << 4);
    const uint32_t boff = lds_addr(Bs) + b_lane;
    // all four k-steps' fragments are requested before the first MFMA (two steps up front, the next two as the
    // first ones are consumed: the LGKM counter holds 15)
    auto mma_tile = [&](int aslot, int bslot) {
        bf16x8 a[4][2], b[4][2];
        const uint32_t ab = aslot * GM_AIMG, bb = boff + bslot * GM_BTILE;
        wait_lgkm0();                  // counted waits below: nothing else (LDS or scalar loads) may be in flight
        rd_frags<0>(a[0][0], a[0][1], b[0][0], b[0][1], aoff[0] + ab, bb);
        rd_frags<8192>(a[1][0], a[1][1], b[1][0], b[1][1], aoff[1] + ab, bb);
        rd_frags<16384>(a[2][0], a[2][1], b[2][0], b[2][1], aoff[2] + ab, bb);
        if (V & 2) __builtin_amdgcn_s_setprio(1);
        unsigned long long tm0 = 0, tm1 = 0, tm2 = 0, tm3 = 0, tm4 = 0;
        if (V & 256) tm0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk == 0) { wait_lgkm<8>(); if (V & 256) tm1 = __builtin_amdgcn_s_memtime(); }
            if (kk == 1) { rd_frags<24576>(a[3][0], a[3][1], b[3][0], b[3][1], aoff[3] + ab, bb); wait_lgkm<8>(); if (V & 256) tm2 = __builtin_amdgcn_s_memtime(); }
            if (kk == 2) { wait_lgkm<4>(); if (V & 256) tm3 = __builtin_amdgcn_s_memtime(); }
            if (kk == 3) { wait_lgkm<0>(); if (V & 256) tm4 = __builtin_amdgcn_s_memtime(); }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][tm], b[kk][tn], acc[tm][tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (V & 2) __builtin_amdgcn_s_setprio(0);
        if ((V & 256) && blockIdx.x == 0 && lane == 0 && ds2 < 1000) {
            unsigned long long te = __builtin_amdgcn_s_memtime();
            gm_dbg[w * 1024 + ds2++] = tm1 - tm0; gm_dbg[w * 1024 + ds2++] = tm2 - tm1; gm_dbg[w * 1024 + ds2++] = tm3 - tm2;
            gm_dbg[w * 1024 + ds2++] = tm4 - tm3; gm_dbg[w * 1024 + ds2++] = te - tm4;
        }
    };

    if (grp == 0) { issue_raw(0, 0); issue_raw(1, 1); issue_raw(2, 2); issue_raw(3, 3); }
    else { issue_b(0, 0); issue_b(1, 1); }

    int bs = 0, bprev = 0;            // B slots of tile j and of tile j-1
    int ds = 0;
    gm_stamp<V>(w, ds);
    for (int ai = 0; ai < cnt; ++ai) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int j = ai * P + p;
            // ---- phase 2j
            if (grp == 0) {
                if (p == 0) {
                    wait_vm<6>();                                   // this SNP block's raw bytes (3 blocks stay in flight)
                    gm_stamp<V>(w, ds);
                    convert(ai & 3, ai & 1);
                    issue_raw(ai + 4, ai & 3);
                    wait_lgkm0();
                }
            } else {
                if (j > 0) mma_tile((p == 0 ? ai - 1 : ai) & 1, bprev);
                gm_stamp<V>(w, ds);
                wait_vm<8>();                                       // weight tile j landed (tile j+1 stays in flight)
            }
            gm_stamp<V>(w, ds);
            __builtin_amdgcn_s_barrier();
            gm_stamp<V>(w, ds);
            // ---- phase 2j+1
            if (grp == 0) {
                mma_tile(ai & 1, bs);
            } else {
                int bs2 = bs + 2; if (bs2 >= GM_NB) bs2 -= GM_NB;
                issue_b(j + 2, bs2);                                // slot of tile j-1: both groups are past it
            }
            wait_lgkm0();
            gm_stamp<V>(w, ds);
            __builtin_amdgcn_s_barrier();
            gm_stamp<V>(w, ds);
            bprev = bs;
            bs = bs + 1; if (bs >= GM_NB) bs -= GM_NB;
        }
    }
    if (grp == 1) mma_tile((cnt - 1) & 1, bprev);
    wait_vm<0>();
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();

    float* const ep = reinterpret_cast<float*>(gm_smem) + w * 4096;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) ep[(tm * 32 + rowmap(r, hi)) * 64 + tn * 32 + jl] = acc[tm][tn][r];
    wait_lgkm0();
    float* pout = partial + ((int64_t)g * Mp + mt * GM_BM + wm * 64) * GM_HP + wn * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = i * 4 + (lane >> 4), c4 = (lane & 15) * 4;
        *reinterpret_cast<f32x4*>(pout + (int64_t)row * GM_HP + c4) = *reinterpret_cast<const f32x4*>(ep + row * 64 + c4);
    }
}

// ---- register-staged schedule -----------------------------------------------------------------------------
// Measured on the LDS-DMA kernels above: one global_load_lds costs the issuing wave ~150 cycles (4 waves x 8
// pieces = 1,213 cycles, 8 waves x 4 = 580), so the 40 pieces of a 128 x 256 x 64 step cost more wave time than its
// 16 MFMAs per wave.  Here the tiles go HBM/L2 -> VGPRs (plain 16-byte loads, two iterations ahead, two static
// register sets) -> LDS (ds_write_b128, linear copy of the image tile; genotypes are widened on the way), which
// issues in ~35 cycles per KB.  One barrier per 64-SNP step; the two halves of the workgroup run the step's two
// parts in opposite order (waves 0-3: stage tile j+1, then multiply tile j; waves 4-7: multiply, then stage), so
// each SIMD has one wave in its matrix segment while the other moves data.
template <int P, int V>
__global__ __launch_bounds__(GM_NT) void l1_gemm3_kernel(const uint8_t* __restrict__ X, int64_t pitch,
                                                         const int32_t* __restrict__ rows, int n, int Kp,
                                                         const unsigned char* __restrict__ tiles,
                                                         float* __restrict__ partial, int G, int n_mt, int nkt64) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char gm_smem[];
    unsigned char* const Bs = gm_smem;                         // 2 x 32 KB
    unsigned char* const As = gm_smem + 2 * GM_BTILE;          // 2 x 16 KB

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = w >> 2;
    const int jl = lane & 31, hi = lane >> 5;
    int g, mt;
    if ((G & 15) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        const int r = idx / n_mt;
        g = 2 * xcd + (r & 1) + 16 * (r >> 1);
    } else if ((G & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        g = xcd + 8 * (idx / n_mt);
    } else {
        g = blockIdx.x % G;
        mt = blockIdx.x / G;
    }
    const int Mp = n_mt * GM_BM;
    const int cnt = (nkt64 - g + G - 1) / G;
    const int nB = cnt * P;

    const int xm = t >> 2, xj = t & 3;
    int xrow_i = mt * GM_BM + xm;
    if (xrow_i > n - 1) xrow_i = n - 1;
    const uint8_t* const xsrc = X + (int64_t)rows[xrow_i] * pitch;

    struct stage_regs { u32x4 b[4]; u32x4 x; };
    // tile j = (SNP block j / P, piece j % P) of this workgroup; past the end: the last tile again (never multiplied)
    auto load_tile = [&](stage_regs& R, int j, bool with_x) {
        const int jj = j < nB ? j : nB - 1;
        const int a = jj / P, p = jj % P;
        const int kt = g + a * G;
        const u32x4* src = reinterpret_cast<const u32x4*>(tiles + ((int64_t)kt * P + p) * GM_BTILE) + t;
#pragma unroll
        for (int i = 0; i < 4; ++i) R.b[i] = src[i * 512];
        if (with_x) {
            int koff = kt * GM_BK + 16 * xj;
            if (koff > Kp - 16) koff = Kp - 16;
            R.x = *reinterpret_cast<const u32x4*>(xsrc + koff);
        }
    };
    auto store_tile = [&](const stage_regs& R, int bslot, bool with_x, int islot) {
        u32x4* dst = reinterpret_cast<u32x4*>(Bs + bslot * GM_BTILE) + t;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i * 512] = R.b[i];
        if (with_x) {
            u32x4 o0, o1;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t b = R.x[d];
                const uint32_t lo = pack_top16(fbits((float)(b & 255u)), fbits((float)((b >> 8) & 255u)));
                const uint32_t hh = pack_top16(fbits((float)((b >> 16) & 255u)), fbits((float)(b >> 24)));
                if (d < 2) { o0[2 * d] = lo; o0[2 * d + 1] = hh; }
                else { o1[2 * (d - 2)] = lo; o1[2 * (d - 2) + 1] = hh; }
            }
            unsigned char* ad = As + islot * GM_AIMG + ((xm ^ (2 * xj)) << 4);
            *reinterpret_cast<u32x4*>(ad + (2 * xj) * 2048) = o0;
            *reinterpret_cast<u32x4*>(ad + (2 * xj + 1) * 2048) = o1;
        }
    };

    const int wm = w & 1, wn = w >> 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = f32x16{0};
    const int a_row = wm * 64 + jl;
    const int b_lane = hi * 4096 + (wn * 64 + jl) * 16;

    auto mma_tile = [&](const unsigned char* Ab, const unsigned char* Bb) {
        bf16x8 a[2][2], b[2][2];
        auto rd = [&](int kk, int s) {
            const int ao = (2 * kk + hi) * 2048 + ((a_row ^ (2 * kk)) << 4);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) a[s][tm] = *reinterpret_cast<const bf16x8*>(Ab + ao + tm * 512);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) b[s][tn] = *reinterpret_cast<const bf16x8*>(Bb + b_lane + kk * 8192 + tn * 512);
        };
        rd(0, 0);
        if (V & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) rd(kk + 1, (kk + 1) & 1);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1][tm], b[kk & 1][tn], acc[tm][tn], 0, 0, 0);
        }
        if (V & 2) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: tiles 0, 1 requested; tile 0 into the LDS; tile 2 requested
    int ds = 0;
    stage_regs R0, R1;
    load_tile(R0, 0, true);
    load_tile(R1, 1, 1 % P == 0);
    store_tile(R0, 0, true, 0);
    load_tile(R0, 2, 2 % P == 0);
    __syncthreads();

    // iteration j: LDS slot j&1 holds tile j, register set (j+1)&1 holds tile j+1, the other set tile j+2 (in flight).
    // Unrolled by 2P so that every slot / register set / "first piece of a SNP block" decision is static.
    for (int base = 0; base < nB; base += 2 * P) {
#pragma unroll
        for (int u = 0; u < 2 * P; ++u) {
            const int j = base + u;
            const bool first_next = ((u + 1) % P) == 0;          // tile j+1 is the first piece of its SNP block
            const bool first_3 = ((u + 3) % P) == 0;             // so is tile j+3
            const int islot_next = ((u + 1) / P) & 1;            // genotype image slot of tile j+1's SNP block
            const int islot_cur = (u / P) & 1;
            auto stage = [&]() {
                if ((u & 1) == 0) { store_tile(R1, 1, first_next, islot_next); load_tile(R1, j + 3, first_3); }
                else { store_tile(R0, 0, first_next, islot_next); load_tile(R0, j + 3, first_3); }
            };
            auto mma = [&]() { if (j < nB) mma_tile(As + islot_cur * GM_AIMG, Bs + (u & 1) * GM_BTILE); };
            gm_stamp<V>(w, ds);
            if ((V & 4) || grp == 0) { stage(); gm_stamp<V>(w, ds); mma(); }
            else { mma(); gm_stamp<V>(w, ds); stage(); }
            gm_stamp<V>(w, ds);
            __syncthreads();
        }
    }

    float* const ep = reinterpret_cast<float*>(gm_smem) + w * 4096;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) ep[(tm * 32 + rowmap(r, hi)) * 64 + tn * 32 + jl] = acc[tm][tn][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    wait_lgkm0();
    float* pout = partial + ((int64_t)g * Mp + mt * GM_BM + wm * 64) * GM_HP + wn * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = i * 4 + (lane >> 4), c4 = (lane & 15) * 4;
        *reinterpret_cast<f32x4*>(pout + (int64_t)row * GM_HP + c4) = *reinterpret_cast<const f32x4*>(ep + row * 64 + c4);
    }
}

// ---- weights straight to registers ------------------------------------------------------------------------
// In-kernel stamps of the schedules above show the LDS as the contended resource: per 64-SNP step the workgroup
// writes the 32 KB weight tile + 16 KB genotype image and reads 128 KB of fragments, and a wave's matrix segment
// stretches from 750 to 1,100-1,700 cycles whenever its partner stages data.  Here a wave owns 32 units and ALL 128
// rows (4 row tiles x 1 unit tile), so its weight fragments are private: they go HBM/L2 -> VGPRs in exactly the
// MFMA B-operand layout (the image's [chunk][unit][8 SNPs] order makes a fragment two 512-byte runs) and never
// touch the LDS.  Only the genotype image is shared through the LDS (16 KB written, 128 KB read per step), and the
// workgroup barrier is needed once per SNP block (every P weight tiles), not once per tile.
template <int P, int V>
__global__ __launch_bounds__(GM_NT) void l1_gemm4_kernel(const uint8_t* __restrict__ X, int64_t pitch,
                                                         const int32_t* __restrict__ rows, int n, int Kp,
                                                         const unsigned char* __restrict__ tiles,
                                                         float* __restrict__ partial, int G, int n_mt, int nkt64) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char gm_smem[];
    unsigned char* const As = gm_smem;                         // 2 x 16 KB (the epilogue reuses 8 x 16 KB)

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int jl = lane & 31, hi = lane >> 5;
    int g, mt;
    if ((G & 15) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        const int r = idx / n_mt;
        g = 2 * xcd + (r & 1) + 16 * (r >> 1);
    } else if ((G & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        mt = idx % n_mt;
        g = xcd + 8 * (idx / n_mt);
    } else {
        g = blockIdx.x % G;
        mt = blockIdx.x / G;
    }
    const int Mp = n_mt * GM_BM;
    const int cnt = (nkt64 - g + G - 1) / G;
    const int nB = cnt * P;

    const int xm = t >> 2, xj = t & 3;
    int xrow_i = mt * GM_BM + xm;
    if (xrow_i > n - 1) xrow_i = n - 1;
    const uint8_t* const xsrc = X + (int64_t)rows[xrow_i] * pitch;
    auto load_x = [&](u32x4& R, int ai) {
        const int a = ai < cnt ? ai : cnt - 1;
        int koff = (g + a * G) * GM_BK + 16 * xj;
        if (koff > Kp - 16) koff = Kp - 16;
        R = *reinterpret_cast<const u32x4*>(xsrc + koff);
    };
    auto widen = [&](const u32x4& R, int islot) {
        u32x4 o0, o1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t b = R[d];
            const uint32_t lo = pack_top16(fbits((float)(b & 255u)), fbits((float)((b >> 8) & 255u)));
            const uint32_t hh = pack_top16(fbits((float)((b >> 16) & 255u)), fbits((float)(b >> 24)));
            if (d < 2) { o0[2 * d] = lo; o0[2 * d + 1] = hh; }
            else { o1[2 * (d - 2)] = lo; o1[2 * (d - 2) + 1] = hh; }
        }
        unsigned char* ad = As + islot * GM_AIMG + ((xm ^ (2 * xj)) << 4);
        *reinterpret_cast<u32x4*>(ad + (2 * xj) * 2048) = o0;
        *reinterpret_cast<u32x4*>(ad + (2 * xj + 1) * 2048) = o1;
    };
    // this wave's weight fragments of tile j: chunk 2 kk + hi, unit w*32 + jl
    struct bregs { bf16x8 b[4]; };
    const int b_lane = hi * 4096 + (w * 32 + jl) * 16;
    auto load_b = [&](bregs& R, int j) {
        const int jj = j < nB ? j : nB - 1;
        const int kt = g + (jj / P) * G, p = jj % P;
        const unsigned char* src = tiles + ((int64_t)kt * P + p) * GM_BTILE + b_lane;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) R.b[kk] = *reinterpret_cast<const bf16x8*>(src + ((V & 8) ? 0 : kk * 8192));
    };

    f32x16 acc[4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) acc[tm] = f32x16{0};
    // V & 32: the SNP block's 16 genotype fragments are read once and kept in registers for all P weight pieces
    bf16x8 af[4][4];
    auto read_a = [&](const unsigned char* Ab) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const unsigned char* ap = Ab + (2 * kk + hi) * 2048 + ((jl ^ (2 * kk)) << 4);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) af[kk][tm] = *reinterpret_cast<const bf16x8*>(ap + tm * 512);
        }
    };
    auto mma_tile = [&](const unsigned char* Ab, const bregs& R) {
        if (V & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (V & 32) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][tm], R.b[kk], acc[tm], 0, 0, 0);
            } else {
            const int kq = (V & 16) ? 0 : kk;
            const unsigned char* ap = Ab + (2 * kq + hi) * 2048 + ((jl ^ (2 * kq)) << 4);
            bf16x8 a[4];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) a[tm] = *reinterpret_cast<const bf16x8*>(ap + tm * 512);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], R.b[kk], acc[tm], 0, 0, 0);
            }
        }
        if (V & 2) __builtin_amdgcn_s_setprio(0);
    };

    // prologue: genotype blocks 0 (widened at once), 1, 2 and weight tiles 0, 1 requested
    u32x4 X0, X1;
    bregs B0, B1, B2;
    {
        u32x4 xt;
        load_x(xt, 0);
        load_x(X1, 1);
        load_x(X0, 2);
        load_b(B0, 0);
        load_b(B1, 1);
        widen(xt, 0);
    }
    __syncthreads();

    // U tiles per unrolled body: register set of tile j = j % 3, genotype image slot = (j / P) % 2, all static
    constexpr int U = (P == 2) ? 12 : 6;
    for (int base = 0; base < nB; base += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base + u;
            const int p = u % P, al = u / P;                  // SNP block of tile j = base / P + al, al & 1 = its image slot
            const int ai = base / P + al;
            // weight tile j + 2 into the set tile j - 1 used
            if (u % 3 == 0) load_b(B2, j + 2);
            else if (u % 3 == 1) load_b(B0, j + 2);
            else load_b(B1, j + 2);
            if (p == 0) {
                // the NEXT SNP block's genotypes: widen into the other image slot (free since the last barrier), then
                // request the block after the next two into the registers just consumed
                if ((al & 1) == 0) { widen(X1, 1); load_x(X1, ai + 3); }
                else { widen(X0, 0); load_x(X0, ai + 3); }
            }
            if (j < nB) {
                const unsigned char* Ab = As + (al & 1) * GM_AIMG;
                if ((V & 32) && p == 0) read_a(Ab);
                if (u % 3 == 0) mma_tile(Ab, B0);
                else if (u % 3 == 1) mma_tile(Ab, B1);
                else mma_tile(Ab, B2);
            }
            if (p == P - 1) __syncthreads();
        }
    }

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
static int gm_nkt64(const loc_dims* d) { return (d->Kp + GM_BK - 1) / GM_BK; }

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
    hipLaunchKernelGGL(l1_image_cvec_kernel, dim3(GM_HP / 64, 8), dim3(1024), 0, st, cpart, nkt, cvec);
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
    if (d->Kp < 16 || x_pitch % 16) { loc_set_error("loc_l1_forward_gemm: needs Kp >= 16 and a 16-byte row pitch"); return -1; }
    const int nkt = gm_nkt64(d);
    const int n_mt = (n + GM_BM - 1) / GM_BM, Mp = n_mt * GM_BM;
    const int variant = target_blocks >> 16;      // experiment selector (see l1_gemm_kernel), 0 = product
    target_blocks &= 0xFFFF;
    if (target_blocks < 1) target_blocks = 256;
    int G = target_blocks / n_mt;
    const int64_t cap = partial_floats / ((int64_t)Mp * GM_HP);
    if (G > cap) G = (int)cap;
    if (G > nkt) G = nkt;
    if (G >= 8) G &= ~7;
    if (G < 1) { loc_set_error("loc_l1_forward_gemm: scratch too small for %d rows", n); return -1; }
    const unsigned char* base = static_cast<const unsigned char*>(image);
    const float* cvec = reinterpret_cast<const float*>(base);
    const unsigned char* tiles = base + gm_tiles_off(d);
    hipStream_t st = (hipStream_t)stream;
#define GM_LAUNCH(PP, VV)                                                                                      \
    {                                                                                                          \
        int rc = gm_set_lds(l1_gemm_kernel<PP, VV>);                                                           \
        if (rc) return rc;                                                                                     \
        hipLaunchKernelGGL((l1_gemm_kernel<PP, VV>), dim3(n_mt * G), dim3(GM_NT), GM_LDS, st, X, x_pitch, rows, n, \
                           d->Kp, tiles, partial, G, n_mt, nkt);                                               \
    }
#define GM_LAUNCH2(PP, VV)                                                                                     \
    {                                                                                                          \
        int rc = gm_set_lds(l1_gemm2_kernel<PP, VV>);                                                          \
        if (rc) return rc;                                                                                     \
        hipLaunchKernelGGL((l1_gemm2_kernel<PP, VV>), dim3(n_mt * G), dim3(GM_NT), GM_LDS, st, X, x_pitch, rows, n, \
                           d->Kp, tiles, partial, G, n_mt, nkt);                                               \
    }
#define GM_LAUNCH3(PP, VV)                                                                                     \
    {                                                                                                          \
        int rc = gm_set_lds(l1_gemm3_kernel<PP, VV>);                                                          \
        if (rc) return rc;                                                                                     \
        hipLaunchKernelGGL((l1_gemm3_kernel<PP, VV>), dim3(n_mt * G), dim3(GM_NT), 131072, st, X, x_pitch, rows, n, \
                           d->Kp, tiles, partial, G, n_mt, nkt);                                               \
    }
#define GM_LAUNCH4(PP, VV)                                                                                     \
    {                                                                                                          \
        int rc = gm_set_lds(l1_gemm4_kernel<PP, VV>);                                                          \
        if (rc) return rc;                                                                                     \
        hipLaunchKernelGGL((l1_gemm4_kernel<PP, VV>), dim3(n_mt * G), dim3(GM_NT), 131072, st, X, x_pitch, rows, n, \
                           d->Kp, tiles, partial, G, n_mt, nkt);                                               \
    }
#define GM_VARIANTS(PP)                                                                                        \
    switch (variant) {                                                                                         \
        case 1032: GM_LAUNCH4(PP, 8) break;                                                                    \
        case 1040: GM_LAUNCH4(PP, 16) break;                                                                   \
        case 1048: GM_LAUNCH4(PP, 24) break;                                                                   \
        case 1056: GM_LAUNCH4(PP, 32) break;                                                                   \
        case 1058: GM_LAUNCH4(PP, 34) break;                                                                   \
        case 1024: GM_LAUNCH4(PP, 0) break;                                                                    \
        case 1026: GM_LAUNCH4(PP, 2) break;                                                                    \
        case 640: GM_LAUNCH3(PP, 128) break;                                                                   \
        case 644: GM_LAUNCH3(PP, 132) break;                                                                   \
        case 512: GM_LAUNCH3(PP, 0) break;                                                                     \
        case 514: GM_LAUNCH3(PP, 2) break;                                                                     \
        case 516: GM_LAUNCH3(PP, 4) break;                                                                     \
        case 32: GM_LAUNCH(PP, 32) break;                                                                      \
        case 96: GM_LAUNCH2(PP, 32) break;                                                                     \
        case 98: GM_LAUNCH2(PP, 34) break;                                                                     \
        case 192: GM_LAUNCH2(PP, 128) break;                                                                   \
        case 320: GM_LAUNCH2(PP, 256) break;                                                                   \
        case 64: GM_LAUNCH2(PP, 0) break;                                                                      \
        case 65: GM_LAUNCH2(PP, 1) break;                                                                      \
        case 66: GM_LAUNCH2(PP, 2) break;                                                                      \
        case 67: GM_LAUNCH2(PP, 3) break;                                                                      \
        case 1: GM_LAUNCH(PP, 1) break;                                                                        \
        case 2: GM_LAUNCH(PP, 2) break;                                                                        \
        case 4: GM_LAUNCH(PP, 4) break;                                                                        \
        case 6: GM_LAUNCH(PP, 6) break;                                                                        \
        case 14: GM_LAUNCH(PP, 14) break;                                                                      \
        case 30: GM_LAUNCH(PP, 30) break;                                                                      \
        case 22: GM_LAUNCH(PP, 22) break;                                                                      \
        default: GM_LAUNCH(PP, 0) break;                                                                       \
    }
    switch (pieces) {
        case 1: GM_VARIANTS(1) break;
        case 2: GM_LAUNCH(2, 0) break;
        default: GM_VARIANTS(3) break;
    }
#undef GM_LAUNCH
    LOC_CHECK_LAUNCH();
    const int64_t MH = (int64_t)Mp * GM_HP;
    hipLaunchKernelGGL(l1_gemm_reduce_kernel, dim3((unsigned)(MH / 256)), dim3(256), 0, st, partial, G, MH, cvec, b1, a1);
    LOC_CHECK_LAUNCH();
    return 0;
}
