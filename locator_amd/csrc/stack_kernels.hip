// Hidden Dense(width, elu) layers, Dropout, the two Dense(2) heads and the Euclidean loss
// (reference: /root/reference/locator/locator.py:314-315, :319-325), forward and backward
// with fused Adam.  M = 32 rows per launch; fp32 MFMA 32x32x2.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// out[b][n] = ELU(sum_k in[b][k] W[k][n] + bias[n]); block per 32-unit output tile, the four
// waves split the contraction and reduce through LDS.
// ---------------------------------------------------------------------------------------------
template <int NHT>
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        float* __restrict__ out_drop,
                                                        const uint8_t* __restrict__ mask, float keep_scale) {
    constexpr int Hp = NHT * 32;
    constexpr int PI = Hp + 4;
    constexpr int NM = NHT;  // 8-wide k groups per wave: (Hp/4)/8
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* inl = smem;            // [32][PI]
    float* red = smem + 32 * PI;  // [4][16][64]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, jl = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.x * 32;

    for (int f = t; f < 32 * Hp / 4; f += 256) {
        int b = f / (Hp / 4), c4 = f % (Hp / 4);
        *reinterpret_cast<f32x4*>(inl + b * PI + 4 * c4) = reinterpret_cast<const f32x4*>(in)[f];
    }
    // B operand straight from global: W[k][n0 + jl], 128-byte rows per half-wave
    const int kb = w * (Hp / 4);
    float bw[NM][4];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bw[m][c] = W[(int64_t)(kb + 8 * m + 4 * hi + c) * Hp + n0 + jl];
    }
    __syncthreads();
    f32x16 acc = {0};
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        f32x4 a = *reinterpret_cast<const f32x4*>(inl + jl * PI + kb + 8 * m + 4 * hi);
        acc = mfma32(a[0], bw[m][0], acc);
        acc = mfma32(a[1], bw[m][1], acc);
        acc = mfma32(a[2], bw[m][2], acc);
        acc = mfma32(a[3], bw[m][3], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int p = t + 256 * i, r = p >> 6, ln = p & 63;
        float z = (red[(0 * 16 + r) * 64 + ln] + red[(1 * 16 + r) * 64 + ln]) +
                  (red[(2 * 16 + r) * 64 + ln] + red[(3 * 16 + r) * 64 + ln]);
        int b = rowmap(r, ln >> 5), n = n0 + (ln & 31);
        float a = elu_f(z + bias[n]);
        out[b * Hp + n] = a;
        if (mask) out_drop[b * Hp + n] = mask[b * Hp + n] ? a * keep_scale : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// One backward launch, two kinds of block:
//   DX blocks (blockIdx < n_dx): dz_prev[b][k] = (sum_n dz[b][n] W[k][n]) * drop * ELU'(a_prev[b][k])
//   DW waves  (the rest): for ANOTHER layer (the one above, whose dx is already done):
//       dW2[k][n] = sum_b in2[b][k] dz2[b][n], db2[n] = sum_b dz2[b][n], then Adam in place.
// ---------------------------------------------------------------------------------------------
template <int NHT>
__global__ __launch_bounds__(256) void dense_bwd_kernel(
    int n_dx, const float* __restrict__ dz, const float* __restrict__ W, const float* __restrict__ a_prev,
    const uint8_t* __restrict__ mask, float keep_scale, float* __restrict__ dz_prev, const float* __restrict__ in2,
    const float* __restrict__ dz2, float* __restrict__ W2, float* __restrict__ mW2, float* __restrict__ vW2,
    float* __restrict__ b2, float* __restrict__ mb2, float* __restrict__ vb2, const float* __restrict__ alpha_tab,
    int alpha_tab_len, const float* __restrict__ lr, const int* __restrict__ t_base, int t_off) {
    constexpr int Hp = NHT * 32;
    constexpr int PI = Hp + 4;
    constexpr int NM = NHT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, jl = lane & 31, hi = lane >> 5;

    if ((int)blockIdx.x < n_dx) {
        float* dzl = smem;             // [32][PI]  A operand, lanes <-> rows
        float* wl = smem + 32 * PI;    // [32][PI]  W rows k0..k0+31
        float* red = wl + 32 * PI;     // [4][16][64]
        const int k0 = blockIdx.x * 32;
        for (int f = t; f < 32 * Hp / 4; f += 256) {
            int b = f / (Hp / 4), c4 = f % (Hp / 4);
            *reinterpret_cast<f32x4*>(dzl + b * PI + 4 * c4) = reinterpret_cast<const f32x4*>(dz)[f];
            *reinterpret_cast<f32x4*>(wl + b * PI + 4 * c4) =
                reinterpret_cast<const f32x4*>(W + (int64_t)k0 * Hp)[f];
        }
        __syncthreads();
        const int nb = w * (Hp / 4);
        f32x16 acc = {0};
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            f32x4 a = *reinterpret_cast<const f32x4*>(dzl + jl * PI + nb + 8 * m + 4 * hi);
            f32x4 b = *reinterpret_cast<const f32x4*>(wl + jl * PI + nb + 8 * m + 4 * hi);
            acc = mfma32(a[0], b[0], acc);
            acc = mfma32(a[1], b[1], acc);
            acc = mfma32(a[2], b[2], acc);
            acc = mfma32(a[3], b[3], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = acc[r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int p = t + 256 * i, r = p >> 6, ln = p & 63;
            float v = (red[(0 * 16 + r) * 64 + ln] + red[(1 * 16 + r) * 64 + ln]) +
                      (red[(2 * 16 + r) * 64 + ln] + red[(3 * 16 + r) * 64 + ln]);
            int b = rowmap(r, ln >> 5), k = k0 + (ln & 31);
            if (mask) v = mask[b * Hp + k] ? v * keep_scale : 0.f;
            dz_prev[b * Hp + k] = v * elu_grad_from_act(a_prev[b * Hp + k]);
        }
        return;
    }
    // ---- DW wave: one 32x32 tile of W2 per wave
    const int tile = ((int)blockIdx.x - n_dx) * 4 + w;
    if (tile >= NHT * NHT) return;
    const int kt = tile / NHT, nt = tile % NHT;
    const float alpha = adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off);
    float av[16], bv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        int b = 2 * s + hi;
        av[s] = in2[b * Hp + kt * 32 + jl];
        bv[s] = dz2[b * Hp + nt * 32 + jl];
    }
    float wv[16], mv[16], vv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int64_t idx = (int64_t)(kt * 32 + rowmap(r, hi)) * Hp + nt * 32 + jl;
        wv[r] = W2[idx]; mv[r] = mW2[idx]; vv[r] = vW2[idx];
    }
    f32x16 g = {0};
#pragma unroll
    for (int s = 0; s < 16; ++s) g = mfma32(av[s], bv[s], g);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int64_t idx = (int64_t)(kt * 32 + rowmap(r, hi)) * Hp + nt * 32 + jl;
        adam_update(wv[r], mv[r], vv[r], g[r], alpha);
        W2[idx] = wv[r]; mW2[idx] = mv[r]; vW2[idx] = vv[r];
    }
    if (kt == 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += bv[i];
        s += __shfl_xor(s, 32);
        if (hi == 0) {
            int n = nt * 32 + jl;
            float bw = b2[n], bm = mb2[n], bvv = vb2[n];
            adam_update(bw, bm, bvv, s, alpha);
            b2[n] = bw; mb2[n] = bm; vb2[n] = bvv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dense(2) -> Dense(2) -> euclidean_distance_loss, backward, Adam on the head, and
// dz_last[b][k] = dA[b][k] * ELU'(a[b][k]).  One block of 256 threads.
// Loss gradient at d == 0 is taken as 0 (Keras: NaN) — the one intentional deviation.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_train_kernel(const float* __restrict__ a, int Hp, int n_b,
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
    __shared__ float y1s[32][2], dy1s[32][2], dy2s[32][2], ds[32];
    const int t = threadIdx.x;
    const float alpha = adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off);
    {
        const int b = t >> 3, part = t & 7;
        float s0 = 0.f, s1 = 0.f;
        for (int k = part; k < Hp; k += 8) {
            float av = a[b * Hp + k];
            s0 = fmaf(av, wa[2 * k], s0);
            s1 = fmaf(av, wa[2 * k + 1], s1);
        }
        s0 += __shfl_xor(s0, 1); s1 += __shfl_xor(s1, 1);
        s0 += __shfl_xor(s0, 2); s1 += __shfl_xor(s1, 2);
        s0 += __shfl_xor(s0, 4); s1 += __shfl_xor(s1, 4);
        if (part == 0) { y1s[b][0] = s0 + ba[0]; y1s[b][1] = s1 + ba[1]; }
    }
    __syncthreads();
    if (t < 32) {
        const int b = t;
        float y10 = y1s[b][0], y11 = y1s[b][1];
        float w00 = wb[0], w01 = wb[1], w10 = wb[2], w11 = wb[3];
        float d = 0.f, g0 = 0.f, g1 = 0.f;
        if (b < n_b) {
            float y20 = y10 * w00 + y11 * w10 + bb[0];
            float y21 = y10 * w01 + y11 * w11 + bb[1];
            float e0 = y20 - Y[(int64_t)rows[b] * 2], e1 = y21 - Y[(int64_t)rows[b] * 2 + 1];
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
    if (t < 8) {
        float g = 0.f;
        float* p; int64_t off;
        if (t < 4) {  // dWb[i][j] = sum_b y1[b][i] dy2[b][j]
            int i = t >> 1, j = t & 1;
            for (int b = 0; b < 32; ++b) g += y1s[b][i] * dy2s[b][j];
            p = wb + t; off = off_wb + t;
        } else if (t < 6) {
            int j = t - 4;
            for (int b = 0; b < 32; ++b) g += dy2s[b][j];
            p = bb + j; off = off_bb + j;
        } else {
            int c = t - 6;
            for (int b = 0; b < 32; ++b) g += dy1s[b][c];
            p = ba + c; off = off_ba + c;
        }
        float wv = *p, mv = m[off], vv = v[off];
        adam_update(wv, mv, vv, g, alpha);
        *p = wv; m[off] = mv; v[off] = vv;
    }
    for (int k = t; k < Hp; k += 256) {
        float w0 = wa[2 * k], w1 = wa[2 * k + 1], g0 = 0.f, g1 = 0.f;
        for (int b = 0; b < 32; ++b) {
            float av = a[b * Hp + k], d0 = dy1s[b][0], d1 = dy1s[b][1];
            g0 = fmaf(av, d0, g0);
            g1 = fmaf(av, d1, g1);
            dz_last[b * Hp + k] = (d0 * w0 + d1 * w1) * elu_grad_from_act(av);
        }
        float m0 = m[off_wa + 2 * k], v0 = v[off_wa + 2 * k];
        float m1 = m[off_wa + 2 * k + 1], v1 = v[off_wa + 2 * k + 1];
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
        case 16: MACRO(16); break;                                                          \
        default: loc_set_error("%s: width %d unsupported (Hp must be 32..512)", __func__, 32 * (NHT_VALUE)); return -1; \
    }

extern "C" int loc_dense_forward(const float* in, const float* W, const float* b, int Hp, float* out,
                                 float* out_drop, const uint8_t* mask, float keep_scale, void* stream) {
    const int nht = Hp / 32;
    const size_t lds = ((size_t)32 * (Hp + 4) + 4 * 16 * 64) * sizeof(float);
#define LAUNCH(N)                                                                                               \
    {                                                                                                           \
        static size_t lds_set = 0;                                                        \
        if (lds > lds_set) { int rc = set_max_lds2(dense_fwd_kernel<N>, lds); if (rc) return rc; lds_set = lds; }                                                                                      \
        hipLaunchKernelGGL(dense_fwd_kernel<N>, dim3(N), dim3(256), lds, (hipStream_t)stream, in, W, b, out,    \
                           out_drop, mask, keep_scale);                                                         \
    }
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
    const int n_dw = W2 ? (nht * nht + 3) / 4 : 0;
    if (n_dx + n_dw == 0) return 0;
    const size_t lds = n_dx ? ((size_t)2 * 32 * (Hp + 4) + 4 * 16 * 64) * sizeof(float) : 0;
#define LAUNCH(N)                                                                                                \
    {                                                                                                            \
        static size_t lds_set = 0;                                                         \
        if (lds > lds_set) { int rc = set_max_lds2(dense_bwd_kernel<N>, lds); if (rc) return rc; lds_set = lds; }                                                                                       \
        hipLaunchKernelGGL(dense_bwd_kernel<N>, dim3(n_dx + n_dw), dim3(256), lds, (hipStream_t)stream, n_dx,    \
                           dz, W, a_prev, mask, keep_scale, dz_prev, in2, dz2, W2, mW2, vW2, b2, mb2, vb2,       \
                           alpha_tab, alpha_tab_len, lr, t_base, t_off);                                         \
    }
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
    hipLaunchKernelGGL(head_train_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, Hp, n_b, rows, Y, wa, ba,
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
