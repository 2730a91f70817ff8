// Large-M layer-1 forward for the inference sweeps (reference: model.predict on predgen / testgen,
// /root/reference/locator/locator.py:414, :441; the per-epoch validation pass of model.fit, :367-376;
// the --jacknife replicate predictions, :683-747).
//
//     z1[m][h] = sum_k xhat[m][k] W1[k][h],   xhat = x * s_k + t_k   (BatchNorm in inference form)
//             = sum_k x[m][k] (s_k W1[k][h])  +  sum_k t_k W1[k][h]
//
// The first term is the only large-M contraction on locator's path and it runs on the bf16 matrix
// pipe WITHOUT giving up fp32 results: a genotype (0/1/2, any uint8) is exact in bf16, and an fp32
// weight w' = s_k w splits exactly into three bf16 pieces by truncation (8 + 8 + 8 significand bits:
// hi = top 16 bits of w', mid = top 16 bits of w' - hi, lo = w' - hi - mid).  Every product
// x * piece is then exact and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so P = 3 pieces give the
// fp32 contraction up to summation order at 3/16 of the fp32-MFMA cycle cost.  P = 1 (round-to-nearest
// bf16 weights, ~2^-9 relative) and P = 2 (~2^-17) trade accuracy for speed and are opt-in.
// The second term is a per-unit constant, accumulated on the vector ALU while the tile is staged.
//
// Workgroup = 512 threads = 8 waves on a 128-row x Hp tile (wave = one 32-row tile x half the unit
// tiles), split over SNP tiles like the training forward: workgroup (mt, g) walks k-tiles g, g+G, ...
// and leaves partial[g][rows of mt][Hp]; l1_reduce_kernel adds the G partials in a fixed order.
// Per k-tile the 32 x Hp fp32 weight tile (one contiguous run of W1S) is scaled, split and written to
// LDS as bf16 [unit][32 k] images (64-byte rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3 so that
// the MFMA operand ds_read_b128 is conflict-free for its non-contiguous 16-lane groups); the 128 x 32
// genotype tile is widened u8 -> bf16 the same way.  Two LDS stages and two register sets: tile i is
// multiplied from one stage while tile i+1 is converted into the other (its vector work dealt out
// between the MFMA slots) and tile i+2 is in flight from HBM/L2; one barrier per tile.
//
// Measured on 1000 rows x 100,000 SNPs x 256 units (tools/rows_gemm_bench.py): 3 pieces 203 us
// (252 TFLOP/s of fp32-exact contraction, 0.30 of the bf16 pipe's issue slots), 1 piece 119 us
// (429 TFLOP/s, 0.17 of peak); the 32-row fp32-MFMA kernel needs 32 launches x 24 us for the same rows.
// PMC (rocprofv3): per wave MFMA-busy 41 %, VALU (the fp32 -> bf16 split) 31 %, waits 28 %, LDS bank
// conflicts 1/3 of LDS cycles before the store-order fix in conv_w.  The conversion, not the matrix pipe, bounds
// this kernel when the same weights serve many row tiles: from LOC_GEMM_MIN_ROWS(pieces) rows on (1152 with three pieces), loc_predict converts once
// per call into an HBM bf16 image and runs l1_gemm.hip instead; this kernel stays the one for few rows (the
// per-epoch validation sweep), where the conversion is used once.
#include "common.h"


#define KT 32

__device__ __forceinline__ uint32_t rne16(uint32_t u) { return u + 0x7FFFu + ((u >> 16) & 1u); }
__device__ __forceinline__ float lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

// RT = row tiles (of 32) per workgroup: 4 (128 rows, one row tile per wave) or 8 (256 rows, two per wave:
// twice the matrix work per staged weight tile and per operand fetched from LDS).
template <int NHT, int RT, int NT>
struct rows_regs {
    f32x4 w[(NHT * 256 + NT - 1) / NT];
    uint32_t x[RT * 256 / NT];
    float s, t;
};

// NT = threads per workgroup: 512 (8 waves, 4 row groups x 2 unit halves) or 256 (4 waves, 2 x 2: one wave
// per SIMD, so each wave may use the whole 512-register file: 256 accumulators + 256 working registers).
template <int NHT, int P, int RT, int NT>
__global__ __launch_bounds__(NT) void l1_rows_partial_kernel(const uint8_t* __restrict__ X, int64_t pitch,
                                                              const int32_t* __restrict__ rows, int n, int Kp,
                                                              const float* __restrict__ ss4,
                                                              const float* __restrict__ w1s,
                                                              float* __restrict__ partial, int G, int Mp) {
    constexpr int Hp = NHT * 32;
    constexpr int RMT = RT * 32;                 // rows per workgroup
    constexpr int WF4 = NHT * 256;
    constexpr int NLD = (WF4 + NT - 1) / NT;
    constexpr int WM = NT / 128;                 // row groups of waves (x 2 unit halves)
    constexpr int TM = RT / WM;                  // row tiles per wave
    constexpr int TH = (NHT + 1) / 2;            // unit tiles per wave
    constexpr int XW = RT * 256 / NT;            // genotype dwords per thread per k-tile (4 bytes each)
    constexpr int XT = 8 / XW;                   // threads per genotype row
    static_assert(XW >= 2 && XW <= 8 && RT % WM == 0, "unsupported tile shape");
    constexpr int A_BYTES = RMT * 64;
    constexpr int B_BYTES = Hp * 64;             // per piece
    constexpr int STAGE = A_BYTES + P * B_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rows[];
    float* cvec = reinterpret_cast<float*>(smem_rows + 2 * STAGE);

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int jl = lane & 31, hi = lane >> 5;
    const int g = blockIdx.x % G, mt = blockIdx.x / G;
    const int nkt = Kp / KT;
    const float* scale = ss4;
    const float* shift = ss4 + Kp;

    // genotype staging role: row xr of the tile, bytes 4*XW*xp .. of the k-tile
    const int xr = t / XT, xp = t % XT;
    const int xm = mt * RMT + xr;
    const bool xvalid = xm < n;
    const int64_t xrow = xvalid ? (int64_t)rows[xm] * pitch : 0;
    // weight staging role: SNP kl of the tile, units given by the float4 index
    const int kl = jl;
    const bool odd = (kl & 1) != 0;

    float csum[NLD][4];
#pragma unroll
    for (int i = 0; i < NLD; ++i) csum[i][0] = csum[i][1] = csum[i][2] = csum[i][3] = 0.f;

    constexpr bool FULL_W = (WF4 % NT) == 0;   // every thread owns NLD float4 of every weight tile
    constexpr bool FULL_H = (NHT % 2) == 0;     // both unit halves hold TH tiles
    const int cnt = g < nkt ? (nkt - g + G - 1) / G : 0;

    // Requests tile number i of this workgroup (k-tile g + i*G) into a register set.  Straight-line: a tile
    // past the end re-reads the last one with its genotypes and BN shift forced to zero, so converting and
    // even multiplying it is harmless.
    auto load_regs = [&](rows_regs<NHT, RT, NT>& R, int i) {
        const bool live = i < cnt;
        const int kt = g + (live ? i : cnt - 1) * G;
        const f32x4* src = reinterpret_cast<const f32x4*>(w1s + (int64_t)kt * NHT * 1024);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int f = t + NT * j;
            if (FULL_W || f < WF4) R.w[j] = src[f];
        }
        R.s = scale[kt * KT + kl];
        const float sh = shift[kt * KT + kl];
        R.t = live ? sh : 0.f;
        const uint32_t* xs = reinterpret_cast<const uint32_t*>(X + xrow + kt * KT + 4 * XW * xp);
        const uint32_t keep = (xvalid && live) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int j = 0; j < XW; ++j) R.x[j] = xs[j] & keep;
    };

    // weight float4 number j of the tile: scale, split into P bf16 pieces, store as [unit][k] images
    auto conv_w = [&](const rows_regs<NHT, RT, NT>& R, int j, unsigned char* st) {
        const int f = t + NT * j;
        if (FULL_W || f < WF4) {
            unsigned char* Bl = st + A_BYTES;
            const int ht = f >> 8, q = (f >> 6) & 3;
            const int hb = ht * 32 + 8 * q + 4 * hi;       // units hb .. hb+3, SNP kl
            const f32x4 wv = R.w[j];
#pragma unroll
            for (int c = 0; c < 4; ++c) csum[j][c] = fmaf(R.t, wv[c], csum[j][c]);
            const float s0 = wv[0] * R.s, s1 = wv[1] * R.s, s2 = wv[2] * R.s, s3 = wv[3] * R.s;
            // lanes (kl, kl^1) trade halves: the even lane ends with units hb, hb+1 and the odd lane with
            // units hb+2, hb+3, each for the SNP pair (kl & ~1, kl | 1) -> one dword per unit and piece
            const float sendA = odd ? s0 : s2, sendB = odd ? s1 : s3;
            const float keepA = odd ? s2 : s0, keepB = odd ? s3 : s1;
            const float recvA = lane_xor1(sendA), recvB = lane_xor1(sendB);
            float aLo = odd ? recvA : keepA, aHi = odd ? keepA : recvA;
            float bLo = odd ? recvB : keepB, bHi = odd ? keepB : recvB;
            const int u = hb + (odd ? 2 : 0);
            const int ke = kl & ~1;
            const int off = u * 64 + ((((ke >> 3) ^ ((u >> 2) & 3))) << 4) + ((ke & 7) << 1);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                uint32_t al = fbits(aLo), ah = fbits(aHi), bl = fbits(bLo), bh = fbits(bHi);
                if (p == P - 1 && P < 3) { al = rne16(al); ah = rne16(ah); bl = rne16(bl); bh = rne16(bh); }
                // ds_write_b32 banks are (addr/4) % 32 per 32-lane half: rows u (even lanes) and u+2 (odd lanes)
                // share a bank half, so odd lanes store their two rows in the opposite order
                const uint32_t d0 = pack_top16(al, ah), d1 = pack_top16(bl, bh);
                *reinterpret_cast<uint32_t*>(Bl + p * B_BYTES + off + (odd ? 64 : 0)) = odd ? d1 : d0;
                *reinterpret_cast<uint32_t*>(Bl + p * B_BYTES + off + (odd ? 0 : 64)) = odd ? d0 : d1;
                if (p < P - 1) {
                    aLo -= bitsf(al & 0xFFFF0000u); aHi -= bitsf(ah & 0xFFFF0000u);
                    bLo -= bitsf(bl & 0xFFFF0000u); bHi -= bitsf(bh & 0xFFFF0000u);
                }
            }
        }
    };
    // genotype bytes of this thread: u8 -> bf16, 16-byte chunks of the [row][k] image
    auto conv_x = [&](const rows_regs<NHT, RT, NT>& R, unsigned char* st) {
#pragma unroll
        for (int j = 0; j < XW; j += 2) {
            const uint32_t b0 = R.x[j], b1 = R.x[j + 1];
            uint4 v;
            v.x = pack_top16(fbits((float)(b0 & 255u)), fbits((float)((b0 >> 8) & 255u)));
            v.y = pack_top16(fbits((float)((b0 >> 16) & 255u)), fbits((float)(b0 >> 24)));
            v.z = pack_top16(fbits((float)(b1 & 255u)), fbits((float)((b1 >> 8) & 255u)));
            v.w = pack_top16(fbits((float)((b1 >> 16) & 255u)), fbits((float)(b1 >> 24)));
            const int chunk = (XW / 2) * xp + j / 2;
            *reinterpret_cast<uint4*>(st + xr * 64 + ((chunk ^ ((xr >> 2) & 3)) << 4)) = v;
        }
    };

    const int wm = w % WM, wh = w / WM;
    f32x16 acc[TM][TH];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int th = 0; th < TH; ++th) acc[tm][th] = f32x16{0};

    // One k-tile: the 2*P (k-half, piece) slots of matrix work on stage `cur`, with the conversion of the
    // next tile (register set `conv`) into stage `nxt` dealt out between the slots: the vector ALU work
    // runs in the shadow of the MFMAs just issued instead of in a phase of its own.
    constexpr int SLOTS = 2 * P;
    auto tile = [&](const unsigned char* cur, unsigned char* nxt, const rows_regs<NHT, RT, NT>& conv) {
        const unsigned char* Al = cur;
        const unsigned char* Bl = cur + A_BYTES;
        bf16x8 a[2][TM], b[2][TH];
        auto read_a = [&](int s2) {
            const int chunk = 2 * s2 + hi;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int arow = (wm * TM + tm) * 32 + jl;
                a[s2][tm] = *reinterpret_cast<const bf16x8*>(Al + arow * 64 + ((chunk ^ ((arow >> 2) & 3)) << 4));
            }
        };
        auto read_b = [&](int slot) {
            const int s2 = slot / P, p = slot % P;
            const int chunk = 2 * s2 + hi;
#pragma unroll
            for (int th = 0; th < TH; ++th) {
                const int brow = (wh * TH + th) * 32 + jl;
                if (FULL_H || wh * TH + th < NHT)
                    b[slot & 1][th] = *reinterpret_cast<const bf16x8*>(
                        Bl + p * B_BYTES + brow * 64 + ((chunk ^ ((brow >> 2) & 3)) << 4));
            }
        };
        read_a(0);
        read_b(0);
        read_a(1);
#pragma unroll
        for (int slot = 0; slot < SLOTS; ++slot) {
            const int s2 = slot / P;
            if (slot + 1 < SLOTS) read_b(slot + 1);      // operands of the next slot are in flight ...
#pragma unroll
            for (int th = 0; th < TH; ++th) {            // ... while this slot's MFMAs issue ...
                if (FULL_H || wh * TH + th < NHT) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        acc[tm][th] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s2][tm], b[slot & 1][th],
                                                                              acc[tm][th], 0, 0, 0);
                }
            }
#pragma unroll
            for (int c = 0; c <= NLD; ++c) {             // ... and a share of the conversion runs in their shadow
                if (c % SLOTS == slot) {
                    if (c < NLD) conv_w(conv, c, nxt);
                    else conv_x(conv, nxt);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Tile i is computed from LDS stage i & 1 while tile i+1 is converted from register set (i+1) & 1 into
    // the other stage; tile i+2 is requested at the top of the step into the set tile i was converted from,
    // so a request has a whole tile of matrix work to land.
    rows_regs<NHT, RT, NT> r0, r1;
    load_regs(r0, 0);
    load_regs(r1, 1);
#pragma unroll
    for (int c = 0; c < NLD; ++c) conv_w(r0, c, smem_rows);
    conv_x(r0, smem_rows);
    __syncthreads();
    unsigned char* const st0 = smem_rows;
    unsigned char* const st1 = smem_rows + STAGE;
    // single-exit loop over pairs of tiles (a second exit inside the pair would make the register allocator
    // rotate the accumulators instead of updating them in place); an odd last tile is finished below, where
    // the conversion it drags along works on a masked (all-zero) tile into the stage nobody reads
    for (int i = 0; i + 1 < cnt; i += 2) {
        load_regs(r0, i + 2); tile(st0, st1, r1); __syncthreads();
        load_regs(r1, i + 3); tile(st1, st0, r0); __syncthreads();
    }
    if (cnt & 1) tile(st0, st1, r1);

    // shift term of this workgroup's k-tiles: reduce over the 32 SNP lanes, one value per unit
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = csum[i][c];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            const int f = t + NT * i;
            if ((FULL_W || f < WF4) && kl == 0) cvec[(f >> 8) * 32 + 8 * ((f >> 6) & 3) + 4 * hi + c] = v;
        }
    }
    __syncthreads();
    // D[i = row][j = unit]: lane holds unit jl of its tile, rows rowmap(r, hi)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        float* pout = partial + ((int64_t)g * Mp + mt * RMT + (wm * TM + tm) * 32) * Hp;
#pragma unroll
        for (int th = 0; th < TH; ++th) {
            const int ht = wh * TH + th;
            if (FULL_H || ht < NHT) {
                const float cv = cvec[ht * 32 + jl];
#pragma unroll
                for (int r = 0; r < 16; ++r) pout[rowmap(r, hi) * Hp + ht * 32 + jl] = acc[tm][th][r] + cv;
            }
        }
    }
}

static size_t rows_lds_bytes(int Hp, int pieces, int rt) {
    return 2 * ((size_t)rt * 32 * 64 + (size_t)pieces * Hp * 64) + (size_t)Hp * 4;
}
// the 256-row tile (4 waves, 2 x 2, 16 accumulator tiles per wave): built for width 256 only
static int rows_big_ok(int Hp, int pieces) { return Hp == 256 && rows_lds_bytes(Hp, pieces, 8) <= 160 * 1024; }

extern "C" int loc_l1_rows_supported(int Hp, int pieces) {
    if (Hp < 32 || Hp > 512 || Hp % 32) return 0;
    if (pieces < 1 || pieces > 3) return 0;
    return rows_lds_bytes(Hp, pieces, 4) <= 160 * 1024 ? 1 : 0;
}

// defined in l1_kernels.hip
int loc_l1_reduce_launch(const float* partial, int G, int rows_p, int Hp, const float* b1, float* a1, void* stream);

#define ROWS_CASE(N, PP, RR, TT)                                                                                 \
    {                                                                                                            \
        LOC_ENSURE_LDS((l1_rows_partial_kernel<N, PP, RR, TT>), lds);                                            \
        hipLaunchKernelGGL((l1_rows_partial_kernel<N, PP, RR, TT>), dim3(n_mt * G), dim3(TT), lds,               \
                           (hipStream_t)stream, X, x_pitch, rows, n, d->Kp, scale_shift, w1s, partial, G, Mp);   \
    }
#define ROWS_SWITCH_BIG(PP)                                                                                       \
    switch (nht) {                                                                                                \
        case 8: ROWS_CASE(8, PP, 8, 256) break;                                                                   \
        default: loc_set_error("loc_l1_forward_rows: width %d unsupported", 32 * nht); return -1;                 \
    }
#define ROWS_SWITCH_SMALL(PP)                                                                                     \
    switch (nht) {                                                                                                \
        case 1: ROWS_CASE(1, PP, 4, 512) break;   case 2: ROWS_CASE(2, PP, 4, 512) break;                         \
        case 3: ROWS_CASE(3, PP, 4, 512) break;   case 4: ROWS_CASE(4, PP, 4, 512) break;                         \
        case 5: ROWS_CASE(5, PP, 4, 512) break;   case 6: ROWS_CASE(6, PP, 4, 512) break;                         \
        case 7: ROWS_CASE(7, PP, 4, 512) break;   case 8: ROWS_CASE(8, PP, 4, 512) break;                         \
        case 9: ROWS_CASE(9, PP, 4, 512) break;   case 10: ROWS_CASE(10, PP, 4, 512) break;                       \
        case 11: ROWS_CASE(11, PP, 4, 512) break; case 12: ROWS_CASE(12, PP, 4, 512) break;                       \
        case 13: ROWS_CASE(13, PP, 4, 512) break; case 14: ROWS_CASE(14, PP, 4, 512) break;                       \
        case 15: ROWS_CASE(15, PP, 4, 512) break; case 16: ROWS_CASE(16, PP, 4, 512) break;                       \
        default: loc_set_error("loc_l1_forward_rows: width %d unsupported", 32 * nht); return -1;                 \
    }

extern "C" int loc_l1_forward_rows(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n, const loc_dims* d,
                                   const float* scale_shift, const float* w1s, const float* b1, float* partial,
                                   int64_t partial_floats, float* a1, int pieces, int target_blocks,
                                   const loc_tuning* tune, void* stream) {
    if (n < 1) { loc_set_error("loc_l1_forward_rows: n=%d", n); return -1; }
    if (!loc_l1_rows_supported(d->Hp, pieces)) {
        loc_set_error("loc_l1_forward_rows: width %d with %d pieces does not fit the LDS", d->Hp, pieces);
        return -1;
    }
    const int nkt = d->Kp / KT, nht = d->Hp / 32;
    if (target_blocks < 1) target_blocks = 256;
    // Default: 128-row tiles, 8 waves.  tune->rows_rt = 8 selects the 256-row / 4-wave tile (half the conversion
    // and LDS operand traffic per flop, accumulators in AGPRs; needs ceil(n/128) even because a1 is written in
    // whole tiles of LOC_ROWS_TILE = 128 rows).  Measured equal within noise on 1000 x 100k x 256
    // (203-209 vs 208-219 us), so it stays a measurement knob.
    const int n128 = (n + LOC_ROWS_TILE - 1) / LOC_ROWS_TILE;
    const bool big = tune && tune->rows_rt == 8 && (n128 % 2) == 0 && rows_big_ok(d->Hp, pieces);
    const int rmt = big ? 256 : 128;
    const int n_mt = (n + rmt - 1) / rmt, Mp = n_mt * rmt;
    int G = target_blocks / n_mt;
    const int64_t cap = partial_floats / ((int64_t)Mp * d->Hp);
    if (G > cap) G = (int)cap;
    if (G > nkt) G = nkt;
    if (G >= 8) G &= ~7;          // blocks (mt, g) of one g share an XCD (block b -> XCD b % 8) and its L2
    if (G < 1) { loc_set_error("loc_l1_forward_rows: scratch too small for %d rows", n); return -1; }
    const size_t lds = rows_lds_bytes(d->Hp, pieces, big ? 8 : 4);
    if (big) {
        switch (pieces) {
            case 1: ROWS_SWITCH_BIG(1) break;
            case 2: ROWS_SWITCH_BIG(2) break;
            default: ROWS_SWITCH_BIG(3) break;
        }
    } else {
        switch (pieces) {
            case 1: ROWS_SWITCH_SMALL(1) break;
            case 2: ROWS_SWITCH_SMALL(2) break;
            default: ROWS_SWITCH_SMALL(3) break;
        }
    }
    LOC_CHECK_LAUNCH();
    return loc_l1_reduce_launch(partial, G, Mp, d->Hp, b1, a1, stream);
}
