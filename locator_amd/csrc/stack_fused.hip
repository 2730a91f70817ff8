// Row-parallel fused hidden stack (reference: /root/reference/locator/locator.py:319-325, :314-315).
//
// The rows of a minibatch are independent through every Dense+ELU layer, the Dropout, the two
// Dense(2) heads and the per-sample Euclidean loss; only the weight gradients reduce over rows.  So
// instead of one launch per layer (latency-bound: ~5 us each, 20 per step), ONE launch carries each
// group of R batch rows through all layers forward, the heads and the loss, and all layers
// backward, with no communication between workgroups at all.  Every workgroup streams each layer's
// 256 KB kernel from L2 (coalesced 16-byte loads; the backward pass reads a transposed copy kept in
// sync by the Adam kernel), contracting on the vector ALU (R rows is far below an MFMA tile).  The
// row-reducing work — dW, db and Adam for all hidden layers and the heads, and the batch loss — runs
// afterwards in ONE wide launch (stack_dw_all_kernel), one workgroup per 32x32 weight tile.
#include <stdlib.h>

#include "common.h"
#include "stack_tail.h"

#ifndef SF_THREADS
#define SF_THREADS 512
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain vmcnt, i.e. wait for
// the weight prefetch that is deliberately kept in flight across the reduction phases.  Inside
// stack_fused_kernel no thread ever reads another thread's GLOBAL writes, so LDS ordering is all the
// barrier has to provide ("memory" keeps the compiler from moving accesses across it).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NHT, int R, bool TRAIN>
__global__ __launch_bounds__(SF_THREADS) void stack_fused_kernel(
    const float* __restrict__ a1_in, const float* __restrict__ Wh, const float* __restrict__ WhT,
    const float* __restrict__ bh, const float* __restrict__ wa, const float* __restrict__ ba,
    const float* __restrict__ wb, const float* __restrict__ bb, const uint8_t* __restrict__ mask, float keep_scale,
    int L, int n_pre, int n_b, const int32_t* __restrict__ rows, const float* __restrict__ Y,
    float* __restrict__ acts, float* __restrict__ adrop, float* __restrict__ dz, float* __restrict__ head_out,
    float* __restrict__ yhat, float* __restrict__ dist, int xcd_stride, int n_work, int slot_rows,
    const float* __restrict__ rd_partial, int rd_G, int64_t rd_MH, const float* __restrict__ rd_cvec8,
    const float* __restrict__ rd_b1) {
    constexpr int Hp = NHT * 32;
    constexpr int C4 = Hp / 4;               // float4 columns per weight row
    constexpr int KG = SF_THREADS / C4;      // k-groups
    constexpr int KPG = Hp / KG;             // k per group
    static_assert(SF_THREADS % C4 == 0 && Hp % KG == 0 && KPG >= 1, "unsupported width for the fused stack");
    __shared__ __attribute__((aligned(16))) float act[R][Hp];          // current layer input (rows of this block)
    __shared__ __attribute__((aligned(16))) float part[KG][R][Hp];     // per-k-group partial sums
    __shared__ float hs[R][8];

    // xcd_stride > 1: only every xcd_stride-th workgroup works, so (with the observed block -> XCD b % 8
    // dispatch) all row groups share ONE XCD's L2 and each weight line crosses the fabric once.  Pure
    // speed hint: results do not depend on where workgroups land.
    if (xcd_stride > 1 && (blockIdx.x % xcd_stride) != 0) return;
    const int li = xcd_stride > 1 ? blockIdx.x / xcd_stride : blockIdx.x;      // logical workgroup index
    if (li >= n_work) {
        // L2 warm-up helper (same XCD as the workers under the observed block -> XCD dispatch): touch every
        // weight line in pass order so the workers' loads hit this XCD's L2 (~110 GB/s per CU) instead of
        // waiting on the fabric (~65 GB/s per CU).  Results cannot depend on it: the values are discarded.
        // with stride 8/n_x the logical workgroups cycle over n_x XCDs: the helpers that share this one's XCD
        // (every n_x-th) split the weight lines between them, so each XCD's L2 sees all of them
        const int n_x = xcd_stride > 1 ? 8 / xcd_stride : 1;
        const int nh_all = (int)(gridDim.x / (xcd_stride > 1 ? xcd_stride : 1)) - n_work;
        const int hid = (li - n_work) / n_x, nh = nh_all / n_x > 0 ? nh_all / n_x : 1;
        const int64_t n4 = (int64_t)(L - 1) * Hp * Hp / 4;
        const f32x4* w4 = reinterpret_cast<const f32x4*>(Wh);
        f32x4 sink = {0.f, 0.f, 0.f, 0.f};
        for (int64_t i = (int64_t)hid * SF_THREADS + threadIdx.x; i < n4; i += (int64_t)nh * SF_THREADS) {
            f32x4 v = w4[i];
            sink[0] += v[0];
        }
        if (TRAIN && WhT != nullptr) {
            const f32x4* t4 = reinterpret_cast<const f32x4*>(WhT);
            for (int64_t i = n4 - 1 - ((int64_t)hid * SF_THREADS + threadIdx.x); i >= 0; i -= (int64_t)nh * SF_THREADS) {
                f32x4 v = t4[i];                      // backward passes walk the layers from last to first
                sink[1] += v[0];
            }
        }
        asm volatile("" ::"v"(sink[0]), "v"(sink[1]));
        return;
    }
    const int t = threadIdx.x, c4 = t % C4, kq = t / C4;
    const int r0 = li * R;
    const int64_t blk = (int64_t)slot_rows * Hp, HH = (int64_t)Hp * Hp;      // slot = activations of one layer

    // rows of this block: input of layer 2
    if (!TRAIN && rd_partial != nullptr) {
        // the many-row layer-1 GEMM left its SNP-group partial sums: the group sum + shift term + b1 + ELU that
        // l1_gemm_reduce_kernel would do in a launch of its own (33 MB read back by one kernel at 1000 and at 4096 rows)
        // happens here, spread over every row block of the stack launch.  Same association as that kernel - four
        // quarters of the groups, ((q0 + q1) + q2) + q3, then + (shift + b1) - so the activations are the same bits.
        const int gq = (rd_G + 3) / 4;
        for (int i = t; i < R * Hp; i += SF_THREADS) {
            const int r = i / Hp, n = i % Hp;
            float v = 0.f;
            if (r0 + r < n_b) {
                const float* src = rd_partial + (int64_t)(r0 + r) * Hp + n;
                float sq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g0 = q * gq, g1 = g0 + gq < rd_G ? g0 + gq : rd_G;
                    float z = 0.f;
                    for (int g = g0; g < g1; ++g) z += src[(int64_t)g * rd_MH];
                    sq[q] = z;
                }
                float c = rd_cvec8[n];
#pragma unroll
                for (int sl = 1; sl < 8; ++sl) c += rd_cvec8[sl * Hp + n];
                v = elu_f((((sq[0] + sq[1]) + sq[2]) + sq[3]) + (c + rd_b1[n]));
            }
            act[r][n] = v;
        }
    } else {
        for (int i = t; i < R * Hp; i += SF_THREADS) act[i / Hp][i % Hp] = a1_in[(int64_t)(r0 + i / Hp) * Hp + i % Hp];
    }
    __syncthreads();

    // out[r][n] = sum_k in[r][k] * Wcur[k][n]: partial over this thread's k-group, reduced over groups later.
    // The weight stream is software-pipelined: two register buffers of CH rows; while one is consumed
    // the other is in flight, and the first chunk of the NEXT layer's kernel is requested before this
    // layer's reduction so L2 latency never sits on the chain.  Chunk 0 of Wcur is already in bufA.
    constexpr int CH = (KPG / 2) < 16 ? (KPG / 2) : 16;
    constexpr int NCH = KPG / CH;
    static_assert(KPG >= 2 && NCH % 2 == 0, "k-group must split into an even number of chunks");
    f32x4 bufA[CH], bufB[CH];
    auto load_chunk = [&](const float* __restrict__ Wsrc, int c, f32x4 (&buf)[CH]) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(Wsrc) + (int64_t)(kq * KPG + c * CH) * C4 + c4;
#pragma unroll
        for (int j = 0; j < CH; ++j) buf[j] = wp[(int64_t)j * C4];
    };
    auto fma_chunk = [&](int c, const f32x4 (&buf)[CH], f32x4 (&acc)[R]) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float a = act[r][kq * KPG + c * CH + j];
                acc[r][0] = fmaf(a, buf[j][0], acc[r][0]);
                acc[r][1] = fmaf(a, buf[j][1], acc[r][1]);
                acc[r][2] = fmaf(a, buf[j][2], acc[r][2]);
                acc[r][3] = fmaf(a, buf[j][3], acc[r][3]);
            }
        }
    };
    // On entry chunks 0 and 1 of Wcur are in bufA / bufB (or in flight).  Each buffer is refilled with the
    // chunk two ahead the moment it has been consumed; past the end of this layer that is the NEXT pass's
    // kernel (Wnext is always a valid matrix: the last pass prefetches a dummy), so ~a whole layer
    // (256 KB per workgroup) is in flight across the reduction phase and no load is conditional.
    auto contract = [&](const float* __restrict__ Wcur, const float* __restrict__ Wnext) {
        f32x4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int c = 0; c < NCH; c += 2) {
            const bool wrap = c + 2 >= NCH;
            const float* Wn = wrap ? Wnext : Wcur;
            const int cn = wrap ? c + 2 - NCH : c + 2;
            fma_chunk(c, bufA, acc);
            load_chunk(Wn, cn, bufA);
            fma_chunk(c + 1, bufB, acc);
            load_chunk(Wn, cn + 1, bufB);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) *reinterpret_cast<f32x4*>(&part[kq][r][4 * c4]) = acc[r];
    };
    // weight matrix of pass p: forward layers 2..L use Wh[0..L-2]; backward L..2 use WhT[L-2..0]
    const int n_pass = TRAIN ? 2 * (L - 1) : (L - 1);
    auto wseq = [&](int p) -> const float* {
        if (p >= n_pass) return Wh;                      // dummy prefetch after the last pass
        return p < L - 1 ? Wh + (int64_t)p * HH : WhT + (int64_t)(2 * (L - 1) - 1 - p) * HH;
    };
    load_chunk(wseq(0), 0, bufA);
    load_chunk(wseq(0), 1, bufB);
    auto reduced = [&](int r, int n) {     // fixed-order sum over the k-groups
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < KG; ++g) s += part[g][r][n];
        return s;
    };

    // per-thread output slots of the epilogues: element i = t + SF_THREADS*o of the R x Hp block
    constexpr int NO = (R * Hp + SF_THREADS - 1) / SF_THREADS;
    // head operands requested now; consumed after the forward chain
    float h_wa0 = 0.f, h_wa1 = 0.f;
    if (t < Hp) { h_wa0 = wa[2 * t]; h_wa1 = wa[2 * t + 1]; }
    float h_ba0 = 0.f, h_ba1 = 0.f, h_w00 = 0.f, h_w01 = 0.f, h_w10 = 0.f, h_w11 = 0.f, h_bb0 = 0.f, h_bb1 = 0.f;
    float h_y0 = 0.f, h_y1 = 0.f;
    if (t < R) {
        h_ba0 = ba[0]; h_ba1 = ba[1]; h_w00 = wb[0]; h_w01 = wb[1]; h_w10 = wb[2]; h_w11 = wb[3];
        h_bb0 = bb[0]; h_bb1 = bb[1];
        if (Y != nullptr && r0 + t < n_b) { h_y0 = Y[(int64_t)rows[r0 + t] * 2]; h_y1 = Y[(int64_t)rows[r0 + t] * 2 + 1]; }
    }

    // ---------------- forward: layers 2..L
    for (int l = 2; l <= L; ++l) {
        const float* bias = bh + (int64_t)(l - 2) * Hp;
        const bool dr = TRAIN && mask != nullptr && l == n_pre;
        float e_bias[NO], e_keep[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) {          // epilogue operands requested before the contraction
            const int i = t + SF_THREADS * o;
            e_bias[o] = 0.f; e_keep[o] = 1.f;
            if (i < R * Hp) {
                e_bias[o] = bias[i % Hp];
                if (dr) e_keep[o] = mask[(int64_t)(r0 + i / Hp) * Hp + i % Hp] ? keep_scale : 0.f;
            }
        }
        contract(wseq(l - 2), wseq(l - 1));
        lds_barrier();
        float* aout = acts + (int64_t)(l - 1) * blk;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const int i = t + SF_THREADS * o;
            if (i < R * Hp) {
                const int r = i / Hp, n = i % Hp;
                const float a = elu_f(reduced(r, n) + e_bias[o]);
                const int64_t gi = (int64_t)(r0 + r) * Hp + n;
                if (TRAIN) aout[gi] = a;
                float nx = a;
                if (dr) {
                    nx = a * e_keep[o];
                    adrop[gi] = nx;
                }
                act[r][n] = nx;
            }
        }
        lds_barrier();
    }

    // ---------------- heads + loss (per row)
    {
        float p0[R], p1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { p0[r] = 0.f; p1[r] = 0.f; }
        if (t < Hp) {                   // Hp <= SF_THREADS: one k per thread
#pragma unroll
            for (int r = 0; r < R; ++r) { p0[r] = act[r][t] * h_wa0; p1[r] = act[r][t] * h_wa1; }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { p0[r] += __shfl_xor(p0[r], o); p1[r] += __shfl_xor(p1[r], o); }
            if ((t & 63) == 0) { part[0][r][2 * (t >> 6)] = p0[r]; part[0][r][2 * (t >> 6) + 1] = p1[r]; }
        }
        lds_barrier();
        if (t < R) {
            const int r = t, b = r0 + r;
            float y10 = h_ba0, y11 = h_ba1;
            for (int w = 0; w < SF_THREADS / 64; ++w) { y10 += part[0][r][2 * w]; y11 += part[0][r][2 * w + 1]; }
            const float w00 = h_w00, w01 = h_w01, w10 = h_w10, w11 = h_w11;
            const float y20 = y10 * w00 + y11 * w10 + h_bb0;
            const float y21 = y10 * w01 + y11 * w11 + h_bb1;
            float d = 0.f, g0 = 0.f, g1 = 0.f;
            const bool valid = b < n_b;
            if (valid && Y != nullptr) {
                const float e0 = y20 - h_y0, e1 = y21 - h_y1;
                d = sqrtf(fmaxf(e0 * e0 + e1 * e1, 0.f));
                if (d > 0.f) { g0 = e0 / d / (float)n_b; g1 = e1 / d / (float)n_b; }
            }
            if (TRAIN) {
                const float dy10 = g0 * w00 + g1 * w01, dy11 = g0 * w10 + g1 * w11;
                hs[r][0] = dy10; hs[r][1] = dy11;
                float* ho = head_out + 8 * b;       // [d, dy1_0, dy1_1, y1_0, y1_1, dy2_0, dy2_1, -]
                ho[0] = d; ho[1] = dy10; ho[2] = dy11; ho[3] = y10; ho[4] = y11; ho[5] = g0; ho[6] = g1; ho[7] = 0.f;
            } else if (valid) {
                yhat[2 * b] = y20; yhat[2 * b + 1] = y21;
                if (dist) dist[b] = d;
            }
        }
        lds_barrier();
    }
    if (!TRAIN) return;

    // ---------------- dz_L = (dy1 . Wa^T) * ELU'(a_L), then backward through layers L..2
    {
        float* dzo = dz + (int64_t)(L - 1) * blk;
        for (int i = t; i < R * Hp; i += SF_THREADS) {
            const int r = i / Hp, k = i % Hp;
            const float v = (hs[r][0] * wa[2 * k] + hs[r][1] * wa[2 * k + 1]) * elu_grad_from_act(act[r][k]);   // wa: L1/L2 hit
            dzo[(int64_t)(r0 + r) * Hp + k] = v;
            part[0][r][k] = v;     // staged; copied into act after the barrier (act is still being read)
        }
        lds_barrier();
        for (int i = t; i < R * Hp; i += SF_THREADS) act[i / Hp][i % Hp] = part[0][i / Hp][i % Hp];
        lds_barrier();
    }
    for (int l = L; l >= 2; --l) {
        const int p = (L - 1) + (L - l);             // pass index of this backward layer
        const float* aprev = acts + (int64_t)(l - 2) * blk;      // ELU output of layer l-1 (pre-dropout)
        const bool dr = mask != nullptr && l - 1 == n_pre;
        float e_g[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) {          // epilogue operands requested before the contraction
            const int i = t + SF_THREADS * o;
            e_g[o] = 0.f;
            if (i < R * Hp) {
                const int64_t gi = (int64_t)(r0 + i / Hp) * Hp + i % Hp;
                float keep = 1.f;
                if (dr) keep = mask[gi] ? keep_scale : 0.f;
                e_g[o] = keep * elu_grad_from_act(aprev[gi]);
            }
        }
        contract(wseq(p), wseq(p + 1));              // sum_n dz_l[r][n] * W_l[k][n] = dz_l . (W_l^T)[n][k]
        lds_barrier();
        float* dzo = dz + (int64_t)(l - 2) * blk;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const int i = t + SF_THREADS * o;
            if (i < R * Hp) {
                const int r = i / Hp, k = i % Hp;
                const float v = reduced(r, k) * e_g[o];
                dzo[(int64_t)(r0 + r) * Hp + k] = v;
                act[r][k] = v;
            }
        }
        lds_barrier();
    }
}

// Everything that reduces over the batch rows, for all hidden layers at once: see stack_tail.h (stack_dw_all_body).
template <int NHT, int RB>
__global__ __launch_bounds__(512) void stack_dw_all_kernel(loc_dw_tail_args ta, loc_gb_tail gb) {
    stack_dw_all_body<NHT, RB>((int)blockIdx.x, ta, gb);
}

// hidden kernels -> transposed copies (after init / import / best-weight reload)
__global__ void transpose_hidden_kernel(const float* __restrict__ Wh, float* __restrict__ WhT, int Hp, int nl) {
    __shared__ float tile[32][33];
    const int l = blockIdx.z, bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = Wh + (int64_t)l * Hp * Hp;
    float* dst = WhT + (int64_t)l * Hp * Hp;
    for (int j = ty; j < 32; j += 8) tile[j][tx] = src[(int64_t)(by + j) * Hp + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8) dst[(int64_t)(bx + j) * Hp + by + tx] = tile[tx][j];
    (void)nl;
}

extern "C" int loc_stack_fused_supported(int Hp) { return Hp == 64 || Hp == 128 || Hp == 256 || Hp == 512; }

#define SF_SWITCH(MACRO)                                                    \
    switch (Hp) {                                                           \
        case 64: MACRO(2); break;                                           \
        case 128: MACRO(4); break;                                          \
        case 256: MACRO(8); break;                                          \
        case 512: MACRO(16); break;                                         \
        default: loc_set_error("%s: fused stack needs width 64/128/256/512 after padding (got %d)", __func__, Hp); return -1; \
    }

extern "C" int loc_transpose_hidden(const float* Wh, float* WhT, int Hp, int n_hidden, void* stream) {
    if (n_hidden <= 0) return 0;
    hipLaunchKernelGGL(transpose_hidden_kernel, dim3(Hp / 32, Hp / 32, n_hidden), dim3(256), 0, (hipStream_t)stream, Wh,
                       WhT, Hp, n_hidden);
    LOC_CHECK_LAUNCH();
    return 0;
}

constexpr int SF_R = 2;   // batch rows per workgroup -> 16 workgroups per 32-row block
// rows per workgroup of the TRAINING launch (loc_tuning.stack_train_rows).  A worker is bound by its weight stream (4.7 MB of
// Wh + WhT from its XCD's L2, ~43 us at 110 GB/s per compute unit) whatever its row count, and the rows' arithmetic that does
// not overlap with the stream comes on top: measured at the metric's shape (bench.py, 32-row steps, 12 helpers) 4 rows per
// workgroup 189.9 us per step, 2 rows (rounds 1-4) 171.3, 1 row 162.4 (32 workers + helpers over two XCDs) - same bits.
// One row per workgroup is ahead at every shape tried (profiles/r05_stack_train_rows.log): widths 64 / 128 / 512 +4 %,
// 5,830 SNPs +10 %, 20,000 +9 %, 500,000 +1 %, --batch_size 64 +1.7 %, 128 +0.8 %.
static int sf_train_rows(const loc_tuning* tune, int n_b) {
    const int v = tune ? tune->stack_train_rows : 0;
    (void)n_b;
    return (v == 1 || v == 2 || v == 4) ? v : 1;
}

// L2 warm-up helper workgroups and XCD placement stride of the fused stack: speed hints (loc_tuning), defaults
// measured at width 256: helpers 0 -> 78.6 us, 4 -> 63, 8 -> 51.7, 12 -> 51.0, 32 -> 55
static int sf_helpers(const loc_tuning* tune) {
    if (!tune || tune->stack_helpers == 0) return 12;
    return tune->stack_helpers < 0 ? 0 : tune->stack_helpers;
}
static int sf_xcd_stride(const loc_tuning* tune) {
    const int v = tune ? tune->stack_xcd_stride : 0;
    return v == 1 || v == 2 || v == 4 || v == 8 ? v : 8;
}

extern "C" int loc_stack_forward_backward(const float* a1_in, const float* Wh, const float* WhT, const float* bh,
                                          const float* wa, const float* ba, const float* wb, const float* bb,
                                          const uint8_t* mask, float keep_scale, int Hp, int L, int n_pre, int n_b,
                                          int slot_rows, const int32_t* rows, const float* Y, float* acts,
                                          float* adrop, float* dz, float* head_out, const loc_tuning* tune,
                                          void* stream) {
    if (n_b < 1 || n_b > slot_rows || slot_rows % 32) {
        loc_set_error("loc_stack_forward_backward: n_b=%d, slot_rows=%d", n_b, slot_rows);
        return -1;
    }
    // every row of the row blocks in use is carried (rows >= n_b get a zero loss gradient), so the tail and the
    // layer-1 backward can contract whole 32-row blocks
    const int rpw = sf_train_rows(tune, n_b);
    const int nblk = (n_b + 31) / 32 * (32 / rpw);
    int xs = sf_xcd_stride(tune);
    int nh = xs > 1 ? sf_helpers(tune) : 0;
    while (xs > 1 && (nblk + nh + 8 / xs - 1) / (8 / xs) > 32) {     // more row groups than one XCD holds
        xs /= 2;
        nh = (nh + 8 / xs - 1) / (8 / xs) * (8 / xs);
    }
    if (xs == 1) nh = 0;
#define LAUNCH_TR(N, RR)                                                                                           \
    hipLaunchKernelGGL((stack_fused_kernel<N, RR, true>), dim3((nblk + nh) * xs), dim3(SF_THREADS), 0,             \
                       (hipStream_t)stream, a1_in, Wh, WhT, bh, wa, ba, wb, bb, mask, keep_scale, L, n_pre, n_b,  \
                       rows, Y, acts, adrop, dz, head_out, (float*)nullptr, (float*)nullptr, xs, nblk, slot_rows,         \
                       (const float*)nullptr, 0, (int64_t)0, (const float*)nullptr, (const float*)nullptr);
#define LAUNCH1(N) LAUNCH_TR(N, 1)
#define LAUNCH2(N) LAUNCH_TR(N, 2)
#define LAUNCH4(N) LAUNCH_TR(N, 4)
    if (rpw == 1) { SF_SWITCH(LAUNCH1) } else if (rpw == 4) { SF_SWITCH(LAUNCH4) } else { SF_SWITCH(LAUNCH2) }
#undef LAUNCH1
#undef LAUNCH2
#undef LAUNCH4
#undef LAUNCH_TR
    LOC_CHECK_LAUNCH();
    return 0;
}

static int sf_eval_launch(const float* a1, const float* rd_partial, int rd_G, int64_t rd_MH, const float* rd_cvec8,
                          const float* rd_b1, const float* Wh, const float* bh, const float* wa, const float* ba,
                          const float* wb, const float* bb, int Hp, int L, int n_b, const int32_t* rows, const float* Y,
                          float* yhat, float* dist, void* stream, int rows_form = 0) {
    // rows_form: 0 = by row count (width 256: 2, 4 or 8 rows per workgroup on the vector ALU while one round of workgroups
    // covers the rows, above 8 x compute units rows 16 or 32 rows per workgroup on the fp32 matrix pipe, stack_rows.hip: the
    // measured round times are there); 1 = always the 32-row matrix-pipe form (where supported), 2 = always the 16-row form;
    // -1 = always 2 rows per workgroup on the vector ALU, -2 = 4 rows, -3 = 8 rows (measurement / tests)
    if (rows_form >= 0 && loc_stack_rows_supported(Hp, L) && (rows_form > 0 || n_b >= loc_stack_rows_min_rows()))
        return sr_eval_launch(a1, rd_partial, rd_G, rd_MH, rd_cvec8, rd_b1, Wh, bh, wa, ba, wb, bb, L, n_b, rows, Y, yhat, dist,
                              rows_form == 1 ? 32 : rows_form == 2 ? 16 : 0, stream);
    const loc_tuning* tune = nullptr;
    int rpw = SF_R;
    if (Hp == 256) {
        const int cu = sr_compute_units();
        if (rows_form == -2 || (rows_form == 0 && n_b > 2 * cu && n_b <= 4 * cu)) rpw = 4;
        else if (rows_form == -3 || (rows_form == 0 && n_b > 4 * cu)) rpw = 8;
    }
    const int nblk = (n_b + rpw - 1) / rpw;
    // Workers + warm-up helpers of one launch should be co-resident (32 CUs per XCD): with more row groups
    // than one XCD holds, spread them over 2, 4 or all 8 XCDs (stride 4, 2, 1) instead of running in rounds.
    int xs = sf_xcd_stride(tune);
    int nh = xs > 1 ? sf_helpers(tune) : 0;
    while (xs > 1 && (nblk + nh + 8 / xs - 1) / (8 / xs) > 32) {
        xs /= 2;
        nh = (nh + 8 / xs - 1) / (8 / xs) * (8 / xs);      // the same number of helpers on every XCD in use
    }
    if (xs == 1) nh = 0;
#define LAUNCH_R(N, RR)                                                                                           \
    hipLaunchKernelGGL((stack_fused_kernel<N, RR, false>), dim3((nblk + nh) * xs), dim3(SF_THREADS), 0,           \
                       (hipStream_t)stream, a1, Wh, (const float*)nullptr, bh, wa, ba, wb, bb,                    \
                       (const uint8_t*)nullptr, 1.f, L, 0, n_b, rows, Y, (float*)nullptr, (float*)nullptr,        \
                       (float*)nullptr, (float*)nullptr, yhat, dist, xs, nblk, 32, rd_partial, rd_G, rd_MH, rd_cvec8,  \
                       rd_b1);
#define LAUNCH(N) LAUNCH_R(N, SF_R)
    if (rpw == 4) { LAUNCH_R(8, 4) }
    else if (rpw == 8) { LAUNCH_R(8, 8) }

    else { SF_SWITCH(LAUNCH) }
#undef LAUNCH
#undef LAUNCH_R
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_stack_forward_eval(const float* a1, const float* Wh, const float* bh, const float* wa,
                                      const float* ba, const float* wb, const float* bb, int Hp, int L, int n_b,
                                      const int32_t* rows, const float* Y, float* yhat, float* dist, void* stream) {
    return sf_eval_launch(a1, nullptr, 0, 0, nullptr, nullptr, Wh, bh, wa, ba, wb, bb, Hp, L, n_b, rows, Y, yhat, dist, stream);
}
extern "C" int loc_stack_forward_eval_form(const float* a1, const float* Wh, const float* bh, const float* wa,
                                           const float* ba, const float* wb, const float* bb, int Hp, int L, int n_b,
                                           const int32_t* rows, const float* Y, float* yhat, float* dist, int rows_form,
                                           void* stream) {
    return sf_eval_launch(a1, nullptr, 0, 0, nullptr, nullptr, Wh, bh, wa, ba, wb, bb, Hp, L, n_b, rows, Y, yhat, dist, stream,
                          rows_form);
}

extern "C" int loc_stack_forward_eval_partial(const float* partial, int groups, int64_t group_stride, const float* cvec8,
                                              const float* b1, const float* Wh, const float* bh, const float* wa,
                                              const float* ba, const float* wb, const float* bb, int Hp, int L, int n_b,
                                              const int32_t* rows, const float* Y, float* yhat, float* dist, int rows_form,
                                              void* stream) {
    if (!partial || groups < 1 || group_stride < (int64_t)n_b * Hp || !cvec8 || !b1) {
        loc_set_error("loc_stack_forward_eval_partial: groups=%d group_stride=%lld n_b=%d", groups, (long long)group_stride, n_b);
        return -1;
    }
    return sf_eval_launch(nullptr, partial, groups, group_stride, cvec8, b1, Wh, bh, wa, ba, wb, bb, Hp, L, n_b, rows, Y, yhat,
                          dist, stream, rows_form);
}

extern "C" int loc_stack_dw_adam_tail(int Hp, int L, int n_pre, int n_b, int slot_rows, int use_drop, const float* acts,
                                      const float* adrop, const float* dz, const float* head_out, float* params,
                                      float* adam_m, float* adam_v, float* WhT, int64_t off_wh, int64_t off_bh,
                                      int64_t off_wa, int64_t off_ba, int64_t off_wb, int64_t off_bb,
                                      float* loss_out, const float* alpha_tab, int alpha_tab_len, const float* lr,
                                      const int* t_base, int t_off, const loc_gb_tail* gb, void* stream) {
    if (n_b < 1 || n_b > LOC_BIG_BATCH_MAX || n_b > slot_rows) {
        loc_set_error("loc_stack_dw_adam: n_b=%d (limit %d), slot_rows=%d", n_b, LOC_BIG_BATCH_MAX, slot_rows);
        return -1;
    }
    const int nht = Hp / 32;
    loc_gb_tail g;
    if (gb) g = *gb; else { g = loc_gb_tail{}; g.K = 0; }
    const int grid = (L - 1) * nht * nht + 1 + (g.K > 0 ? (g.K + 511) / 512 : 0);
    loc_dw_tail_args ta;
    ta.L = L; ta.n_pre = n_pre; ta.n_b = n_b; ta.use_drop = use_drop;
    ta.acts = acts; ta.adrop = adrop; ta.dz = dz; ta.head_out = head_out;
    ta.P = params; ta.M = adam_m; ta.V = adam_v; ta.WhT = WhT;
    ta.off_wh = off_wh; ta.off_bh = off_bh; ta.off_wa = off_wa; ta.off_ba = off_ba; ta.off_wb = off_wb; ta.off_bb = off_bb;
    ta.loss_out = loss_out; ta.alpha_tab = alpha_tab; ta.alpha_tab_len = alpha_tab_len; ta.lr = lr; ta.t_base = t_base;
    ta.t_off = t_off; ta.slot_rows = slot_rows;
#define LAUNCH_RB(N, R) \
    hipLaunchKernelGGL((stack_dw_all_kernel<N, R>), dim3(grid), dim3(512), 0, (hipStream_t)stream, ta, g);
#define LAUNCH(N)                                                                                                 \
    switch ((n_b + 31) / 32) {                                                                                    \
        case 1: LAUNCH_RB(N, 1) break;                                                                            \
        case 2: LAUNCH_RB(N, 2) break;                                                                            \
        case 3: LAUNCH_RB(N, 3) break;                                                                            \
        case 4: LAUNCH_RB(N, 4) break;                                                                            \
        default: LAUNCH_RB(N, 0) break;   /* more than 128 rows: run-time block count */                          \
    }
    SF_SWITCH(LAUNCH)
#undef LAUNCH
#undef LAUNCH_RB
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_stack_dw_adam(int Hp, int L, int n_pre, int n_b, int use_drop, const float* acts,
                                 const float* adrop, const float* dz, const float* head_out, float* params,
                                 float* adam_m, float* adam_v, float* WhT, int64_t off_wh, int64_t off_bh,
                                 int64_t off_wa, int64_t off_ba, int64_t off_wb, int64_t off_bb, float* loss_out,
                                 const float* alpha_tab, int alpha_tab_len, const float* lr, const int* t_base,
                                 int t_off, void* stream) {
    return loc_stack_dw_adam_tail(Hp, L, n_pre, n_b, 32, use_drop, acts, adrop, dz, head_out, params, adam_m, adam_v, WhT,
                                  off_wh, off_bh, off_wa, off_ba, off_wb, off_bb, loss_out, alpha_tab, alpha_tab_len,
                                  lr, t_base, t_off, nullptr, stream);
}
