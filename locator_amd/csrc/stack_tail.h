// The row-reducing tail of a training step (dW / db + Adam of all hidden layers, heads, batch loss and -- optionally --
// the BatchNorm gamma / beta update) as a device function of the block index, so that it can run as its own launch
// (stack_fused.hip: stack_dw_all_kernel) or as trailing workgroups of the chained layer-1 launch (l1_chain.hip), where it
// fills the compute units that finish their k-tiles one iteration early.
#pragma once
#include "common.h"

struct loc_dw_tail_args {
    int L, n_pre, n_b, use_drop;
    const float *acts, *adrop, *dz, *head_out;
    float *P, *M, *V, *WhT;
    int64_t off_wh, off_bh, off_wa, off_ba, off_wb, off_bb;
    float* loss_out;
    const float* alpha_tab;
    int alpha_tab_len;
    const float* lr;
    const int* t_base;
    int t_off, slot_rows;
};

// l1_chain.hip: loc_l1_backward_adam_chain with the tail above as trailing workgroups of the same launch (tail may be NULL)
int l1_chain_launch(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, const int32_t* rows_next, int n_b_next,
                    const loc_dims* d, float* bn4, const float* bn_next_stats, const float* dz1, float* w1s, float* m1s,
                    float* v1s, float* gamma, float* beta, float* m_gamma, float* v_gamma, float* m_beta, float* v_beta,
                    float* b1, float* m_b1, float* v_b1, const float* alpha_tab, int alpha_tab_len, const float* lr,
                    const int* t_base, int t_off, int grid, float* partial, int64_t partial_floats, const loc_tuning* tune,
                    const loc_dw_tail_args* tail, int rb, void* stream);

// ---------------------------------------------------------------------------------------------
// Everything that reduces over the batch rows, for all hidden layers at once.
// Blocks [0, (L-1)*NHT^2): one 32x32 tile of W_l (l = 2..L): dW = in_l^T dz_l on the matrix core (wave 0),
//   Adam by all 512 threads, and the transposed copy W_l^T refreshed through LDS.  Tile row 0 also does db.
// Last block: heads (dWa, dba, dWb, dbb + Adam) and the batch loss.
// ---------------------------------------------------------------------------------------------
template <int NHT, int RB>
__device__ __forceinline__ void stack_dw_all_body(const int bid, const loc_dw_tail_args& ta, const loc_gb_tail& gb) {
    const int L = ta.L, n_pre = ta.n_pre, n_b = ta.n_b, use_drop = ta.use_drop, slot_rows = ta.slot_rows;
    const float* __restrict__ acts = ta.acts;
    const float* __restrict__ adrop = ta.adrop;
    const float* __restrict__ dz = ta.dz;
    const float* __restrict__ head_out = ta.head_out;
    float* __restrict__ P = ta.P;
    float* __restrict__ M = ta.M;
    float* __restrict__ V = ta.V;
    float* __restrict__ WhT = ta.WhT;
    const int64_t off_wh = ta.off_wh, off_bh = ta.off_bh, off_wa = ta.off_wa, off_ba = ta.off_ba, off_wb = ta.off_wb,
                  off_bb = ta.off_bb;
    float* __restrict__ loss_out = ta.loss_out;
    const float* __restrict__ alpha_tab = ta.alpha_tab;
    const int alpha_tab_len = ta.alpha_tab_len;
    const float* __restrict__ lr = ta.lr;
    const int* __restrict__ t_base = ta.t_base;
    const int t_off = ta.t_off;
    constexpr int Hp = NHT * 32;
    // Blocks past the tiles and the heads (only when gb.K > 0): BatchNorm gamma/beta Adam for 512 SNPs each,
    // from the partial sums the layer-1 backward left -- the step's two row-reducing tails share one launch.
    if (bid > (L - 1) * NHT * NHT) {
        const int k = (bid - (L - 1) * NHT * NHT - 1) * 512 + (int)threadIdx.x;
        if (k < gb.K)
            gamma_beta_adam_body(k, gb.Kp, gb.gbs, gb.gamma, gb.beta, gb.m_gamma, gb.v_gamma, gb.m_beta, gb.v_beta,
                                 adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off), gb.next_stats, gb.bn4);
        return;
    }
    // RB = 0: the number of 32-row blocks is a run-time value (--batch_size > 128): wave w takes blocks w, w + 8, ...
    constexpr int NP = RB == 0 ? 8 : (RB > 1 ? RB : 1);
    __shared__ float gt[32][33];
    __shared__ float gtp[NP][32][33];                   // per-row-block (RB = 0: per-wave) partial tiles
    __shared__ float sbp[NP][32];
    __shared__ float hsm[RB == 0 ? 1 : 32 * RB][8];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, jl = lane & 31, hi = lane >> 5;
    const int64_t blk = (int64_t)slot_rows * Hp, HH = (int64_t)Hp * Hp;
    const int nrb = RB == 0 ? (n_b + 31) / 32 : RB;    // 32-row blocks in use (1 unless --batch_size > 32)
    const float alpha = adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off);
    const int n_tiles = (L - 1) * NHT * NHT;

    if (bid < n_tiles) {
        const int li = bid / (NHT * NHT);          // 0-based hidden index: layer l = li + 2
        const int tile = bid % (NHT * NHT);
        const int kt = tile / NHT, nt = tile % NHT;
        const int l = li + 2;
        const float* in2 = (use_drop && l - 1 == n_pre) ? adrop : acts + (int64_t)(l - 2) * blk;
        const float* dz2 = dz + (int64_t)(l - 1) * blk;
        float* W2 = P + off_wh + li * HH;
        float* mW2 = M + off_wh + li * HH;
        float* vW2 = V + off_wh + li * HH;
        float* WT2 = WhT + li * HH;
        int64_t idx[2];
        float wv[2], mv[2], vv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = t + 512 * i;
            idx[i] = (int64_t)(kt * 32 + (e >> 5)) * Hp + nt * 32 + (e & 31);
            wv[i] = W2[idx[i]]; mv[i] = mW2[idx[i]]; vv[i] = vW2[idx[i]];
        }
        if constexpr (RB == 1) {
            if (w == 0) {
                float av[16], bv[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int b = 2 * s + hi;
                    av[s] = in2[b * Hp + kt * 32 + jl];
                    bv[s] = dz2[b * Hp + nt * 32 + jl];
                }
                f32x16 g = {0};
#pragma unroll
                for (int s = 0; s < 16; ++s) g = mfma32(av[s], bv[s], g);
#pragma unroll
                for (int r = 0; r < 16; ++r) gt[rowmap(r, hi)][jl] = g[r];
                if (kt == 0) {
                    float sb = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) sb += bv[i];
                    sb += __shfl_xor(sb, 32);
                    if (hi == 0) {
                        const int64_t o = off_bh + (int64_t)li * Hp + nt * 32 + jl;
                        float bw = P[o], bm = M[o], bvv = V[o];
                        adam_update(bw, bm, bvv, sb, alpha);
                        P[o] = bw; M[o] = bm; V[o] = bvv;
                    }
                }
            }
        } else if constexpr (RB == 0) {
            // run-time block count: wave w accumulates blocks w, w + 8, ... in order; the 8 per-wave tiles are then
            // added in a fixed order
            {
                f32x16 g = {0};
                float sb = 0.f;
                for (int rb = w; rb < nrb; rb += 8) {
                    float av[16], bv[16];
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const int b = 32 * rb + 2 * s + hi;
                        av[s] = in2[(int64_t)b * Hp + kt * 32 + jl];
                        bv[s] = dz2[(int64_t)b * Hp + nt * 32 + jl];
                    }
#pragma unroll
                    for (int s = 0; s < 16; ++s) g = mfma32(av[s], bv[s], g);
#pragma unroll
                    for (int i = 0; i < 16; ++i) sb += bv[i];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) gtp[w][rowmap(r, hi)][jl] = g[r];
                sb += __shfl_xor(sb, 32);
                if (hi == 0) sbp[w][jl] = sb;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = t + 512 * i;
                float a = gtp[0][e >> 5][e & 31];
#pragma unroll
                for (int rb = 1; rb < 8; ++rb) a += gtp[rb][e >> 5][e & 31];
                gt[e >> 5][e & 31] = a;
            }
            if (kt == 0 && t < 32) {
                float sb = sbp[0][t];
#pragma unroll
                for (int rb = 1; rb < 8; ++rb) sb += sbp[rb][t];
                const int64_t o = off_bh + (int64_t)li * Hp + nt * 32 + t;
                float bw = P[o], bm = M[o], bvv = V[o];
                adam_update(bw, bm, bvv, sb, alpha);
                P[o] = bw; M[o] = bm; V[o] = bvv;
            }
        } else {
            // one wave per row block (their load latencies overlap); the RB partial tiles are added in a fixed order
            if (w < RB) {
                float av[16], bv[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int b = 32 * w + 2 * s + hi;
                    av[s] = in2[b * Hp + kt * 32 + jl];
                    bv[s] = dz2[b * Hp + nt * 32 + jl];
                }
                f32x16 g = {0};
#pragma unroll
                for (int s = 0; s < 16; ++s) g = mfma32(av[s], bv[s], g);
#pragma unroll
                for (int r = 0; r < 16; ++r) gtp[w][rowmap(r, hi)][jl] = g[r];
                float sb = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) sb += bv[i];
                sb += __shfl_xor(sb, 32);
                if (hi == 0) sbp[w][jl] = sb;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = t + 512 * i;
                float a = gtp[0][e >> 5][e & 31];
#pragma unroll
                for (int rb = 1; rb < RB; ++rb) a += gtp[rb][e >> 5][e & 31];
                gt[e >> 5][e & 31] = a;
            }
            if (kt == 0 && t < 32) {
                float sb = sbp[0][t];
#pragma unroll
                for (int rb = 1; rb < RB; ++rb) sb += sbp[rb][t];
                const int64_t o = off_bh + (int64_t)li * Hp + nt * 32 + t;
                float bw = P[o], bm = M[o], bvv = V[o];
                adam_update(bw, bm, bvv, sb, alpha);
                P[o] = bw; M[o] = bm; V[o] = bvv;
            }
        }
        __syncthreads();
        float nw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = t + 512 * i;
            adam_update(wv[i], mv[i], vv[i], gt[e >> 5][e & 31], alpha);
            W2[idx[i]] = wv[i]; mW2[idx[i]] = mv[i]; vW2[idx[i]] = vv[i];
            nw[i] = wv[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int e = t + 512 * i; gt[e >> 5][e & 31] = nw[i]; }   // [k][n]
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {      // W^T[n][k], coalesced along k
            const int e = t + 512 * i, n = e >> 5, k = e & 31;
            WT2[(int64_t)(nt * 32 + n) * Hp + kt * 32 + k] = gt[k][n];
        }
        return;
    }
    // ---- head block
    const int nrow = 32 * nrb;
    if constexpr (RB != 0) {
        for (int i = t; i < 8 * nrow; i += 512) hsm[i >> 3][i & 7] = head_out[i];
        __syncthreads();
    }
    // head_out[b][0..7] = {per-sample loss, dy1[0..1], y1[0..1], dy2[0..1], -}: from LDS, or (run-time block count) from L2
    auto H = [&](int b, int c) -> float { return RB != 0 ? hsm[b][c] : head_out[(int64_t)b * 8 + c]; };
    const float* aL = acts + (int64_t)(L - 1) * blk;
    if (t == 0) {
        float s = 0.f;
        for (int b = 0; b < n_b; ++b) s += H(b, 0);
        loss_out[0] = s / (float)n_b;
    }
    if (t >= 64 && t < 72) {
        const int q = t - 64;
        float g = 0.f;
        int64_t off;
        if (q < 4) {            // dWb[i][j] = sum_b y1[b][i] dy2[b][j]
            const int i = q >> 1, j = q & 1;
            for (int b = 0; b < nrow; ++b) g += H(b, 3 + i) * H(b, 5 + j);
            off = off_wb + q;
        } else if (q < 6) {     // dbb[j] = sum_b dy2[b][j]
            for (int b = 0; b < nrow; ++b) g += H(b, 5 + (q - 4));
            off = off_bb + (q - 4);
        } else {                // dba[c] = sum_b dy1[b][c]
            for (int b = 0; b < nrow; ++b) g += H(b, 1 + (q - 6));
            off = off_ba + (q - 6);
        }
        float wv = P[off], mv = M[off], vv = V[off];
        adam_update(wv, mv, vv, g, alpha);
        P[off] = wv; M[off] = mv; V[off] = vv;
    }
    for (int k = t; k < Hp; k += 512) {      // dWa[k][c] = sum_b a_L[b][k] dy1[b][c]
        float g0 = 0.f, g1 = 0.f;
#pragma unroll 8
        for (int b = 0; b < nrow; ++b) {
            const float av = aL[(int64_t)b * Hp + k];
            g0 = fmaf(av, H(b, 1), g0);
            g1 = fmaf(av, H(b, 2), g1);
        }
        const int64_t o = off_wa + 2 * k;
        float w0 = P[o], m0 = M[o], v0 = V[o], w1 = P[o + 1], m1 = M[o + 1], v1 = V[o + 1];
        adam_update(w0, m0, v0, g0, alpha);
        adam_update(w1, m1, v1, g1, alpha);
        P[o] = w0; M[o] = m0; V[o] = v0; P[o + 1] = w1; M[o + 1] = m1; V[o + 1] = v1;
    }
}
