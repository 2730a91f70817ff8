// Hidden Dense(width, elu) layers, Dropout, the two Dense(2) heads and the Euclidean loss
// (reference: /root/reference/locator/locator.py:314-315, :319-325), forward and backward
// with fused Adam.  M = 32 rows per launch; fp32 MFMA 32x32x2.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// out[b][n] = ELU(sum_k in[b][k] W[k][n] + bias[n]).  Block (512 threads, 8 waves) per 32-unit
// output tile; the waves split the contraction in groups of 8 k and reduce through LDS.  Both
// operands come straight from global/L2 (32 KB activations, 256 KB weights): these launches are
// latency-bound, so the fewer dependent hops the better.
// ---------------------------------------------------------------------------------------------
#define DENSE_THREADS 512
#define DENSE_WAVES 8

template <int NHT>
__global__ __launch_bounds__(DENSE_THREADS) void dense_fwd_kernel(const float* __restrict__ in,
                                                                  const float* __restrict__ W,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ out,
                                                                  float* __restrict__ out_drop,
                                                                  const uint8_t* __restrict__ mask,
                                                                  float keep_scale) {
    constexpr int Hp = NHT * 32;
    constexpr int NG = 4 * NHT;                                    // groups of 8 k
    constexpr int NM = (NG + DENSE_WAVES - 1) / DENSE_WAVES;       // groups per wave
    __shared__ float red[DENSE_WAVES][16][64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, jl = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.x * 32;
    f32x4 av[NM];
    float bw[NM][4];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int g = w + DENSE_WAVES * m;
        if (g < NG) {
            const int k = 8 * g + 4 * hi;
            av[m] = *reinterpret_cast<const f32x4*>(in + jl * Hp + k);
#pragma unroll
            for (int c = 0; c < 4; ++c) bw[m][c] = W[(int64_t)(k + c) * Hp + n0 + jl];
        }
    }
    // epilogue operands requested now, so their latency hides under the loads/MFMAs above
    float e_bias[2], e_keep[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = t + DENSE_THREADS * i, r = p >> 6, ln = p & 63;
        const int b = rowmap(r, ln >> 5), n = n0 + (ln & 31);
        e_bias[i] = bias[n];
        e_keep[i] = mask ? (mask[b * Hp + n] ? keep_scale : 0.f) : 1.f;
    }
    f32x16 acc = {0};
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        if (w + DENSE_WAVES * m < NG) {
            acc = mfma32(av[m][0], bw[m][0], acc);
            acc = mfma32(av[m][1], bw[m][1], acc);
            acc = mfma32(av[m][2], bw[m][2], acc);
            acc = mfma32(av[m][3], bw[m][3], acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][r][lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = t + DENSE_THREADS * i, r = p >> 6, ln = p & 63;
        float z = ((red[0][r][ln] + red[1][r][ln]) + (red[2][r][ln] + red[3][r][ln])) +
                  ((red[4][r][ln] + red[5][r][ln]) + (red[6][r][ln] + red[7][r][ln]));
        const int b = rowmap(r, ln >> 5), n = n0 + (ln & 31);
        float a = elu_f(z + e_bias[i]);
        out[b * Hp + n] = a;
        if (mask) out_drop[b * Hp + n] = a * e_keep[i];
    }
}

// ---------------------------------------------------------------------------------------------
// One backward launch, two kinds of block:
//   DX blocks (blockIdx < n_dx): dz_prev[b][k] = (sum_n dz[b][n] W[k][n]) * drop * ELU'(a_prev[b][k])
//   DW waves  (the rest): for ANOTHER layer (the one above, whose dx is already done):
//       dW2[k][n] = sum_b in2[b][k] dz2[b][n], db2[n] = sum_b dz2[b][n], then Adam in place.
// ---------------------------------------------------------------------------------------------
template <int NHT>
__global__ __launch_bounds__(DENSE_THREADS) void dense_bwd_kernel(
    int n_dx, const float* __restrict__ dz, const float* __restrict__ W, const float* __restrict__ a_prev,
    const uint8_t* __restrict__ mask, float keep_scale, float* __restrict__ dz_prev, const float* __restrict__ in2,
    const float* __restrict__ dz2, float* __restrict__ W2, float* __restrict__ mW2, float* __restrict__ vW2,
    float* __restrict__ b2, float* __restrict__ mb2, float* __restrict__ vb2, const float* __restrict__ alpha_tab,
    int alpha_tab_len, const float* __restrict__ lr, const int* __restrict__ t_base, int t_off) {
    constexpr int Hp = NHT * 32;
    constexpr int NG = 4 * NHT;
    constexpr int NM = (NG + DENSE_WAVES - 1) / DENSE_WAVES;
    __shared__ float red[DENSE_WAVES][16][64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, jl = lane & 31, hi = lane >> 5;

    if ((int)blockIdx.x < n_dx) {
        // D[i = row b][j = k]: A[i][kk] = dz[b][n], B[kk][j] = W[k0 + j][n]; lanes hold 16-byte pieces of rows
        const int k0 = blockIdx.x * 32;
        f32x4 av[NM], bv[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int g = w + DENSE_WAVES * m;
            if (g < NG) {
                const int n = 8 * g + 4 * hi;
                av[m] = *reinterpret_cast<const f32x4*>(dz + jl * Hp + n);
                bv[m] = *reinterpret_cast<const f32x4*>(W + (int64_t)(k0 + jl) * Hp + n);
            }
        }
        float e_g[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = t + DENSE_THREADS * i, r = p >> 6, ln = p & 63;
            const int b = rowmap(r, ln >> 5), k = k0 + (ln & 31);
            float keep = mask ? (mask[b * Hp + k] ? keep_scale : 0.f) : 1.f;
            e_g[i] = keep * elu_grad_from_act(a_prev[b * Hp + k]);
        }
        f32x16 acc = {0};
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (w + DENSE_WAVES * m < NG) {
                acc = mfma32(av[m][0], bv[m][0], acc);
                acc = mfma32(av[m][1], bv[m][1], acc);
                acc = mfma32(av[m][2], bv[m][2], acc);
                acc = mfma32(av[m][3], bv[m][3], acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) red[w][r][lane] = acc[r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = t + DENSE_THREADS * i, r = p >> 6, ln = p & 63;
            float v = ((red[0][r][ln] + red[1][r][ln]) + (red[2][r][ln] + red[3][r][ln])) +
                      ((red[4][r][ln] + red[5][r][ln]) + (red[6][r][ln] + red[7][r][ln]));
            const int b = rowmap(r, ln >> 5), k = k0 + (ln & 31);
            dz_prev[b * Hp + k] = v * e_g[i];
        }
        return;
    }
    // ---- DW block: one 32x32 tile of W2.  Every thread requests its two (w, m, v) triples up front;
    // wave 0 forms the gradient tile on the matrix core and shares it through LDS; Adam + stores by all.
    const int tile = (int)blockIdx.x - n_dx;
    const int kt = tile / NHT, nt = tile % NHT;
    const float alpha = adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off);
    float (*gt)[33] = reinterpret_cast<float (*)[33]>(&red[0][0][0]);   // [32][33]
    int64_t idx[2];
    float wv[2], mv[2], vv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = t + DENSE_THREADS * i;
        idx[i] = (int64_t)(kt * 32 + (e >> 5)) * Hp + nt * 32 + (e & 31);
        wv[i] = W2[idx[i]]; mv[i] = mW2[idx[i]]; vv[i] = vW2[idx[i]];
    }
    if (w == 0) {
        float av[16], bv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            int b = 2 * s + hi;
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
                int n = nt * 32 + jl;
                float bw = b2[n], bm = mb2[n], bvv = vb2[n];
                adam_update(bw, bm, bvv, sb, alpha);
                b2[n] = bw; mb2[n] = bm; vb2[n] = bvv;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = t + DENSE_THREADS * i;
        adam_update(wv[i], mv[i], vv[i], gt[e >> 5][e & 31], alpha);
        W2[idx[i]] = wv[i]; mW2[idx[i]] = mv[i]; vW2[idx[i]] = vv[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Dense(2) -> Dense(2) -> euclidean_distance_loss, backward, Adam on the head, and
// dz_last[b][k] = dA[b][k] * ELU'(a[b][k]).  One block of 256 threads.
// Loss gradient at d == 0 is taken as 0 (Keras: NaN) — the one intentional deviation.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void head_train_kernel(const float* __restrict__ a, int Hp, int n_b,
                                                         const int32_t* __restrict__ rows,
                                                         const float* __restrict__ Y, float* __restrict__ wa,
                                                         float* __restrict__ ba, float* __restrict__ wb,
                                                         float* __restrict__ bb, float* __restrict__ m,
                                                         float* __restrict__ v, int64_t off_wa, int64_t off_ba,
                                                         int64_t off_wb, int64_t off_bb,
                                                         float* __restrict__ dz_last, float* __restrict__ loss_out,
                                                         const float* __restrict__ alpha_tab, int alpha_tab_len,
                                                         const float* __restrict__ lr,
                                                         const int* __restrict__ t_base, int t_off) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    float* al = hsm;                     // [32][Hp + 1]
    float* wal = al + 32 * (Hp + 1);     // [Hp][2]
    __shared__ float y1s[32][2], dy1s[32][2], dy2s[32][2], ds[32], ys[32][2];
    const int t = threadIdx.x, P = Hp + 1;
    const float alpha = adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off);
    // one round trip: activations, Wa and the targets into LDS
    for (int i = t; i < 32 * Hp; i += 512) al[(i / Hp) * P + (i % Hp)] = a[i];
    for (int i = t; i < 2 * Hp; i += 512) wal[i] = wa[i];
    if (t < 64) {
        int b = t >> 1;
        ys[b][t & 1] = b < n_b ? Y[(int64_t)rows[b] * 2 + (t & 1)] : 0.f;
    }
    const float ba0 = ba[0], ba1 = ba[1], w00 = wb[0], w01 = wb[1], w10 = wb[2], w11 = wb[3];
    const float bb0 = bb[0], bb1 = bb[1];
    float pm0 = 0.f, pv0 = 0.f, pm1 = 0.f, pv1 = 0.f;     // Adam moments of Wa[t], requested early
    if (t < Hp) {
        pm0 = m[off_wa + 2 * t]; pv0 = v[off_wa + 2 * t];
        pm1 = m[off_wa + 2 * t + 1]; pv1 = v[off_wa + 2 * t + 1];
    }
    __syncthreads();
    {   // y1[b][c] = sum_k a[b][k] Wa[k][c]: 16 threads per row
        const int b = t >> 4, part = t & 15;
        float s0 = 0.f, s1 = 0.f;
        for (int k = part; k < Hp; k += 16) {
            float av = al[b * P + k];
            s0 = fmaf(av, wal[2 * k], s0);
            s1 = fmaf(av, wal[2 * k + 1], s1);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
        if (part == 0) { y1s[b][0] = s0 + ba0; y1s[b][1] = s1 + ba1; }
    }
    __syncthreads();
    if (t < 32) {
        const int b = t;
        float y10 = y1s[b][0], y11 = y1s[b][1];
        float d = 0.f, g0 = 0.f, g1 = 0.f;
        if (b < n_b) {
            float y20 = y10 * w00 + y11 * w10 + bb0;
            float y21 = y10 * w01 + y11 * w11 + bb1;
            float e0 = y20 - ys[b][0], e1 = y21 - ys[b][1];
            d = sqrtf(fmaxf(e0 * e0 + e1 * e1, 0.f));
            if (d > 0.f) { g0 = e0 / d / (float)n_b; g1 = e1 / d / (float)n_b; }
        }
        ds[b] = d;
        dy2s[b][0] = g0; dy2s[b][1] = g1;
        dy1s[b][0] = g0 * w00 + g1 * w01;
        dy1s[b][1] = g0 * w10 + g1 * w11;
    }
    __syncthreads();
    if (t == 0) {
        float s = 0.f;
        for (int b = 0; b < n_b; ++b) s += ds[b];
        loss_out[0] = s / (float)n_b;
    }
    if (t >= 64 && t < 72) {   // a different wave from the loss sum
        const int q = t - 64;
        float g = 0.f;
        float* p; int64_t off;
        if (q < 4) {  // dWb[i][j] = sum_b y1[b][i] dy2[b][j]
            int i = q >> 1, j = q & 1;
            for (int b = 0; b < 32; ++b) g += y1s[b][i] * dy2s[b][j];
            p = wb + q; off = off_wb + q;
        } else if (q < 6) {
            int j = q - 4;
            for (int b = 0; b < 32; ++b) g += dy2s[b][j];
            p = bb + j; off = off_bb + j;
        } else {
            int c = q - 6;
            for (int b = 0; b < 32; ++b) g += dy1s[b][c];
            p = ba + c; off = off_ba + c;
        }
        float wv = *p, mv = m[off], vv = v[off];
        adam_update(wv, mv, vv, g, alpha);
        *p = wv; m[off] = mv; v[off] = vv;
    }
    // dz_last[b][k] = (dy1[b] . Wa[k]) * ELU'(a[b][k]): thread (b-half, k)
    for (int i = t; i < 32 * Hp; i += 512) {
        const int b = i / Hp, k = i - b * Hp;
        float av = al[b * P + k];
        dz_last[i] = (dy1s[b][0] * wal[2 * k] + dy1s[b][1] * wal[2 * k + 1]) * elu_grad_from_act(av);
    }
    // dWa[k][c] = sum_b a[b][k] dy1[b][c], Adam
    for (int k = t; k < Hp; k += 512) {
        float w0 = wal[2 * k], w1 = wal[2 * k + 1], g0 = 0.f, g1 = 0.f;
#pragma unroll 8
        for (int b = 0; b < 32; ++b) {
            float av = al[b * P + k];
            g0 = fmaf(av, dy1s[b][0], g0);
            g1 = fmaf(av, dy1s[b][1], g1);
        }
        float m0 = pm0, v0 = pv0, m1 = pm1, v1 = pv1;     // the early request covers k == t (every k when Hp <= 512)
        if (k != t) {                                       // widths above 512: second pass of the loop
            m0 = m[off_wa + 2 * k]; v0 = v[off_wa + 2 * k];
            m1 = m[off_wa + 2 * k + 1]; v1 = v[off_wa + 2 * k + 1];
        }
        adam_update(w0, m0, v0, g0, alpha);
        adam_update(w1, m1, v1, g1, alpha);
        wa[2 * k] = w0; wa[2 * k + 1] = w1;
        m[off_wa + 2 * k] = m0; v[off_wa + 2 * k] = v0;
        m[off_wa + 2 * k + 1] = m1; v[off_wa + 2 * k + 1] = v1;
    }
}

__global__ __launch_bounds__(256) void head_eval_kernel(const float* __restrict__ a, int Hp, int n_b,
                                                        const float* __restrict__ wa, const float* __restrict__ ba,
                                                        const float* __restrict__ wb, const float* __restrict__ bb,
                                                        float* __restrict__ yhat, const int32_t* __restrict__ rows,
                                                        const float* __restrict__ Y, float* __restrict__ dist) {
    const int t = threadIdx.x, b = t >> 3, part = t & 7;
    float s0 = 0.f, s1 = 0.f;
    for (int k = part; k < Hp; k += 8) {
        float av = a[b * Hp + k];
        s0 = fmaf(av, wa[2 * k], s0);
        s1 = fmaf(av, wa[2 * k + 1], s1);
    }
    s0 += __shfl_xor(s0, 1); s1 += __shfl_xor(s1, 1);
    s0 += __shfl_xor(s0, 2); s1 += __shfl_xor(s1, 2);
    s0 += __shfl_xor(s0, 4); s1 += __shfl_xor(s1, 4);
    if (part == 0 && b < n_b) {
        float y10 = s0 + ba[0], y11 = s1 + ba[1];
        float y20 = y10 * wb[0] + y11 * wb[2] + bb[0];
        float y21 = y10 * wb[1] + y11 * wb[3] + bb[1];
        yhat[2 * b] = y20;
        yhat[2 * b + 1] = y21;
        if (dist) {
            float e0 = y20 - Y[(int64_t)rows[b] * 2], e1 = y21 - Y[(int64_t)rows[b] * 2 + 1];
            dist[b] = sqrtf(fmaxf(e0 * e0 + e1 * e1, 0.f));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
template <typename F>
static int set_max_lds2(F* func, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(func),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        loc_set_error("hipFuncSetAttribute(%zu): %s", bytes, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

#define NHT_SWITCH(NHT_VALUE, MACRO)                                                        \
    switch (NHT_VALUE) {                                                                    \
        case 1: MACRO(1); break;   case 2: MACRO(2); break;   case 3: MACRO(3); break;      \
        case 4: MACRO(4); break;   case 5: MACRO(5); break;   case 6: MACRO(6); break;      \
        case 7: MACRO(7); break;   case 8: MACRO(8); break;   case 9: MACRO(9); break;      \
        case 10: MACRO(10); break; case 11: MACRO(11); break; case 12: MACRO(12); break;    \
        case 13: MACRO(13); break; case 14: MACRO(14); break; case 15: MACRO(15); break;    \
        case 16: MACRO(16); break; case 17: MACRO(17); break; case 18: MACRO(18); break;    \
        case 19: MACRO(19); break; case 20: MACRO(20); break; case 21: MACRO(21); break;    \
        case 22: MACRO(22); break; case 23: MACRO(23); break; case 24: MACRO(24); break;    \
        case 25: MACRO(25); break; case 26: MACRO(26); break; case 27: MACRO(27); break;    \
        case 28: MACRO(28); break; case 29: MACRO(29); break; case 30: MACRO(30); break;    \
        case 31: MACRO(31); break; case 32: MACRO(32); break;                               \
        default: loc_set_error("%s: width %d unsupported (Hp must be 32..1024)", __func__, 32 * (NHT_VALUE)); return -1; \
    }

extern "C" int loc_dense_forward(const float* in, const float* W, const float* b, int Hp, float* out,
                                 float* out_drop, const uint8_t* mask, float keep_scale, void* stream) {
    const int nht = Hp / 32;
#define LAUNCH(N)                                                                                           \
    hipLaunchKernelGGL(dense_fwd_kernel<N>, dim3(N), dim3(DENSE_THREADS), 0, (hipStream_t)stream, in, W, b, out, \
                       out_drop, mask, keep_scale);
    NHT_SWITCH(nht, LAUNCH)
#undef LAUNCH
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_dense_backward(const float* dz, const float* W, const float* a_prev, const uint8_t* mask,
                                  float keep_scale, float* dz_prev, const float* in2, const float* dz2, float* W2,
                                  float* mW2, float* vW2, float* b2, float* mb2, float* vb2, int Hp,
                                  const float* alpha_tab, int alpha_tab_len, const float* lr, const int* t_base,
                                  int t_off, void* stream) {
    const int nht = Hp / 32;
    const int n_dx = dz_prev ? nht : 0;
    const int n_dw = W2 ? nht * nht : 0;
    if (n_dx + n_dw == 0) return 0;
#define LAUNCH(N)                                                                                               \
    hipLaunchKernelGGL(dense_bwd_kernel<N>, dim3(n_dx + n_dw), dim3(DENSE_THREADS), 0, (hipStream_t)stream,     \
                       n_dx, dz, W, a_prev, mask, keep_scale, dz_prev, in2, dz2, W2, mW2, vW2, b2, mb2, vb2,    \
                       alpha_tab, alpha_tab_len, lr, t_base, t_off);
    NHT_SWITCH(nht, LAUNCH)
#undef LAUNCH
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_head_train(const float* a, int Hp, int n_b, const int32_t* rows, const float* Y, float* wa,
                              float* ba, float* wb, float* bb, float* m, float* v, int64_t off_wa, int64_t off_ba,
                              int64_t off_wb, int64_t off_bb, float* dz_last, float* loss_out,
                              const float* alpha_tab, int alpha_tab_len, const float* lr, const int* t_base,
                              int t_off, void* stream) {
    if (n_b < 1 || n_b > LOC_ROWS) { loc_set_error("loc_head_train: n_b=%d out of 1..32", n_b); return -1; }
    const size_t hlds = ((size_t)32 * (Hp + 1) + 2 * Hp) * sizeof(float);
    LOC_ENSURE_LDS(head_train_kernel, hlds);
    hipLaunchKernelGGL(head_train_kernel, dim3(1), dim3(512), hlds, (hipStream_t)stream, a, Hp, n_b, rows, Y, wa, ba,
                       wb, bb, m, v, off_wa, off_ba, off_wb, off_bb, dz_last, loss_out, alpha_tab, alpha_tab_len,
                       lr, t_base, t_off);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_head_eval(const float* a, int Hp, int n_b, const float* wa, const float* ba, const float* wb,
                             const float* bb, float* yhat, const int32_t* rows, const float* Y, float* dist,
                             void* stream) {
    if (n_b < 1 || n_b > LOC_ROWS) { loc_set_error("loc_head_eval: n_b=%d out of 1..32", n_b); return -1; }
    hipLaunchKernelGGL(head_eval_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, Hp, n_b, wa, ba, wb, bb,
                       yhat, rows, Y, dist);
    LOC_CHECK_LAUNCH();
    return 0;
}
