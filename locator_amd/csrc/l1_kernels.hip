// Layer 1 of locator's network on gfx950: BatchNormalization on the genotype
// matrix + Dense(width, elu) (reference: /root/reference/locator/locator.py:318-320),
// forward and fused backward+Adam.  All arithmetic fp32; contractions on
// v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
#include "common.h"

#define KT 32  // SNPs per k-tile
#define LP 36  // LDS pitch (floats) of a [row][32 k] tile: 16-B aligned rows, conflict-free ds_read_b128

// ---------------------------------------------------------------------------------------------
// BN batch statistics (training) — SURVEY.md A.2.  One thread per SNP.
// out4 = [scale | shift | mean | rstd], scale = gamma*rstd, shift = beta - mean*scale.
// ---------------------------------------------------------------------------------------------
__global__ void bn_batch_stats_kernel(const uint8_t* __restrict__ X, int64_t pitch, const int32_t* __restrict__ rows,
                                      int n_b, int K, int Kp, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float* __restrict__ mov_mean,
                                      float* __restrict__ mov_var, float* __restrict__ out4) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Kp) return;
    float scale = 0.f, shift = 0.f, mean = 0.f, rstd = 0.f;
    if (k < K) {
        int s = 0, ss = 0;
        for (int b = 0; b < n_b; ++b) {
            int x = X[(int64_t)rows[b] * pitch + k];
            s += x;
            ss += x * x;
        }
        mean = (float)s / (float)n_b;
        // biased variance, exact integer numerator: (n*sum(x^2) - sum(x)^2) / n^2
        float var = (float)(n_b * ss - s * s) / (float)(n_b * n_b);
        rstd = 1.0f / sqrtf(var + BN_EPS);
        scale = gamma[k] * rstd;
        shift = beta[k] - mean * scale;
        mov_mean[k] = mov_mean[k] * BN_MOMENTUM + mean * (1.0f - BN_MOMENTUM);
        mov_var[k] = mov_var[k] * BN_MOMENTUM + var * (1.0f - BN_MOMENTUM);
    }
    out4[k] = scale;
    out4[Kp + k] = shift;
    out4[2 * (int64_t)Kp + k] = mean;
    out4[3 * (int64_t)Kp + k] = rstd;
}

__global__ void bn_infer_scale_shift_kernel(int K, int Kp, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, const float* __restrict__ mov_mean,
                                            const float* __restrict__ mov_var, float* __restrict__ out4) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Kp) return;
    float scale = 0.f, shift = 0.f, mean = 0.f, rstd = 0.f;
    if (k < K) {
        mean = mov_mean[k];
        rstd = 1.0f / sqrtf(mov_var[k] + BN_EPS);
        scale = gamma[k] * rstd;
        shift = beta[k] - mean * scale;
    }
    out4[k] = scale;
    out4[Kp + k] = shift;
    out4[2 * (int64_t)Kp + k] = mean;
    out4[3 * (int64_t)Kp + k] = rstd;
}

// ---------------------------------------------------------------------------------------------
// Layer-1 forward, split over SNP tiles.  Block = 512 threads (8 waves); block g walks k-tiles
// g, g+grid, ... and leaves partial[g][32][Hp].  Per k-tile: the 32-SNP x Hp weight tile (one
// contiguous NHT*4 KB run of W1S) and the 32 x 32 xhat tile are staged through LDS
// (double-buffered, register prefetch of the next tile), wave w owns unit tiles w, w+8.
// ---------------------------------------------------------------------------------------------
template <int NHT>
__global__ __launch_bounds__(512) void l1_fwd_partial_kernel(const uint8_t* __restrict__ X, int64_t pitch,
                                                             const int32_t* __restrict__ rows, int n_b, int Kp,
                                                             const float* __restrict__ ss4,
                                                             const float* __restrict__ w1s,
                                                             float* __restrict__ partial) {
    constexpr int Hp = NHT * 32;
    constexpr int WF4 = NHT * 256;               // float4 per weight tile
    constexpr int NLD = (WF4 + 511) / 512;       // float4 loads per thread per tile
    constexpr int BUF = (Hp + 32) * LP;          // floats per LDS buffer: W rows then xhat rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int jl = lane & 31, hi = lane >> 5;
    const int nkt = Kp / KT;
    const float* scale = ss4;
    const float* shift = ss4 + Kp;

    // x staging role: threads 0..255 -> row b = t>>3, SNP quad kq = t&7
    const int xb = t >> 3, xq = t & 7;
    const bool xrole = t < 256;
    const bool xvalid = xrole && xb < n_b;
    const int64_t xrow = xvalid ? (int64_t)rows[xb] * pitch : 0;

    f32x4 wreg[NLD];
    uint32_t xreg = 0;
    f32x4 screg = {0, 0, 0, 0}, shreg = {0, 0, 0, 0};

    auto load_regs = [&](int kt) {
        const f32x4* src = reinterpret_cast<const f32x4*>(w1s + (int64_t)kt * NHT * 1024);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int f = t + 512 * i;
            if (f < WF4) wreg[i] = src[f];
        }
        if (xvalid) {
            int k0 = kt * KT + 4 * xq;
            xreg = *reinterpret_cast<const uint32_t*>(X + xrow + k0);
            screg = *reinterpret_cast<const f32x4*>(scale + k0);
            shreg = *reinterpret_cast<const f32x4*>(shift + k0);
        }
    };
    auto store_lds = [&](float* buf) {
        float* Wl = buf;
        float* xh = buf + Hp * LP;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int f = t + 512 * i;
            if (f < WF4) {
                int ht = f >> 8, q = (f >> 6) & 3, ln = f & 63;
                int h = ht * 32 + 8 * q + 4 * (ln >> 5), k = ln & 31;
                Wl[(h + 0) * LP + k] = wreg[i][0];
                Wl[(h + 1) * LP + k] = wreg[i][1];
                Wl[(h + 2) * LP + k] = wreg[i][2];
                Wl[(h + 3) * LP + k] = wreg[i][3];
            }
        }
        if (xrole) {
            f32x4 v = {0, 0, 0, 0};
            if (xvalid) {
                v[0] = fmaf((float)(xreg & 255u), screg[0], shreg[0]);
                v[1] = fmaf((float)((xreg >> 8) & 255u), screg[1], shreg[1]);
                v[2] = fmaf((float)((xreg >> 16) & 255u), screg[2], shreg[2]);
                v[3] = fmaf((float)(xreg >> 24), screg[3], shreg[3]);
            }
            *reinterpret_cast<f32x4*>(xh + xb * LP + 4 * xq) = v;
        }
    };

    f32x16 acc0 = {0}, acc1 = {0};
    const bool own0 = w < NHT, own1 = (w + 8) < NHT;

    int kt = blockIdx.x;
    if (kt < nkt) {
        load_regs(kt);
        store_lds(smem);
    }
    __syncthreads();
    int cur = 0;
    for (; kt < nkt; kt += gridDim.x) {
        const int nxt = kt + gridDim.x;
        const bool has_next = nxt < nkt;
        if (has_next) load_regs(nxt);
        const float* buf = smem + cur * BUF;
        const float* Wl = buf;
        const float* xh = buf + Hp * LP;
        if (own0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 a = *reinterpret_cast<const f32x4*>(xh + jl * LP + 8 * m + 4 * hi);
                f32x4 b = *reinterpret_cast<const f32x4*>(Wl + (w * 32 + jl) * LP + 8 * m + 4 * hi);
                acc0 = mfma32(a[0], b[0], acc0);
                acc0 = mfma32(a[1], b[1], acc0);
                acc0 = mfma32(a[2], b[2], acc0);
                acc0 = mfma32(a[3], b[3], acc0);
                if (own1) {
                    f32x4 b2 = *reinterpret_cast<const f32x4*>(Wl + ((w + 8) * 32 + jl) * LP + 8 * m + 4 * hi);
                    acc1 = mfma32(a[0], b2[0], acc1);
                    acc1 = mfma32(a[1], b2[1], acc1);
                    acc1 = mfma32(a[2], b2[2], acc1);
                    acc1 = mfma32(a[3], b2[3], acc1);
                }
            }
        }
        if (has_next) store_lds(smem + (cur ^ 1) * BUF);
        __syncthreads();
        cur ^= 1;
    }
    // D[i = row b][j = unit]: lane holds unit jl, rows rowmap(r, hi)
    float* pout = partial + (int64_t)blockIdx.x * 32 * Hp;
    if (own0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pout[rowmap(r, hi) * Hp + w * 32 + jl] = acc0[r];
    }
    if (own1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pout[rowmap(r, hi) * Hp + (w + 8) * 32 + jl] = acc1[r];
    }
}

// a1[b][h] = ELU(sum_g partial[g][b][h] + b1[h]); optional Dropout on this layer's output.
__global__ void l1_reduce_kernel(const float* __restrict__ partial, int G, int Hp, const float* __restrict__ b1,
                                 float* __restrict__ a1, float* __restrict__ a1_drop,
                                 const uint8_t* __restrict__ mask, float keep_scale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = 32 * Hp;
    if (idx >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 4 <= G; g += 4) {
        s0 += partial[(int64_t)(g + 0) * n + idx];
        s1 += partial[(int64_t)(g + 1) * n + idx];
        s2 += partial[(int64_t)(g + 2) * n + idx];
        s3 += partial[(int64_t)(g + 3) * n + idx];
    }
    for (; g < G; ++g) s0 += partial[(int64_t)g * n + idx];
    float z = ((s0 + s1) + (s2 + s3)) + b1[idx % Hp];
    float a = elu_f(z);
    a1[idx] = a;
    if (mask) a1_drop[idx] = mask[idx] ? a * keep_scale : 0.f;
}

// ---------------------------------------------------------------------------------------------
// Layer-1 backward + Adam, one pass over W1/m/v (24 B per weight), nothing else materialised.
// Block = 256 threads (4 waves), walks k-tiles kt = blockIdx.x, += grid.  Wave w owns unit
// tiles w, w+4, ...; for each (k-tile, unit-tile) it streams the three 4 KB swizzled runs with
// 16-byte loads straight into the MFMA accumulator layout:
//     dW^T[h][k]  = sum_b dZ[b][h] xhat[b][k]      (A from LDS, B = xhat registers)
//     dxhat[b][k] += sum_h dZ[b][h] W[h][k]        (A from LDS, B = the weight registers)
// then Adam in registers and 16-byte stores.  dxhat is reduced over the block's waves in LDS
// to give dgamma/dbeta (BN on the input has trainable gamma/beta: locator.py:318).
// ---------------------------------------------------------------------------------------------
template <int NHT>
__global__ __launch_bounds__(256) void l1_bwd_adam_kernel(
    const uint8_t* __restrict__ X, int64_t pitch, const int32_t* __restrict__ rows, int n_b, int K, int Kp,
    const float* __restrict__ bn4, const float* __restrict__ dz1, float* __restrict__ w1s, float* __restrict__ m1s,
    float* __restrict__ v1s, float* __restrict__ gamma, float* __restrict__ beta, float* __restrict__ m_gamma,
    float* __restrict__ v_gamma, float* __restrict__ m_beta, float* __restrict__ v_beta, float* __restrict__ b1,
    float* __restrict__ m_b1, float* __restrict__ v_b1, const float* __restrict__ alpha_tab, int alpha_tab_len,
    const float* __restrict__ lr, const int* __restrict__ t_base, int t_off) {
    constexpr int Hp = NHT * 32;
    constexpr int PZ = Hp + 1;  // dZ pitch: lanes<->rows reads hit distinct banks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzl = smem;                    // [32][PZ]
    float* part = dzl + 32 * PZ;          // [4][16][64]
    float* red2 = part + 4 * 16 * 64;     // [4][2][64]
    int* rows_l = reinterpret_cast<int*>(red2 + 4 * 2 * 64);  // [32]

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int jl = lane & 31, hi = lane >> 5;
    const int nkt = Kp / KT;
    const float alpha = adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off);

    for (int i = t; i < 32 * Hp; i += 256) dzl[(i / Hp) * PZ + (i % Hp)] = dz1[i];
    if (t < 32) rows_l[t] = t < n_b ? rows[t] : 0;
    __syncthreads();

    // bias of layer 1: db1[h] = sum_b dZ[b][h]   (block 0 only)
    if (blockIdx.x == 0) {
        for (int ht = w; ht < NHT; ht += 4) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += dzl[rowmap(r, hi) * PZ + ht * 32 + jl];
            s += __shfl_xor(s, 32);
            if (hi == 0) {
                int h = ht * 32 + jl;
                float wv = b1[h], mv = m_b1[h], vv = v_b1[h];
                adam_update(wv, mv, vv, s, alpha);
                b1[h] = wv; m_b1[h] = mv; v_b1[h] = vv;
            }
        }
    }

    const float* sc_p = bn4;
    const float* sh_p = bn4 + Kp;
    const float* mu_p = bn4 + 2 * (int64_t)Kp;
    const float* rs_p = bn4 + 3 * (int64_t)Kp;

    for (int kt = blockIdx.x; kt < nkt; kt += gridDim.x) {
        const int k = kt * KT + jl;
        const float sc = sc_p[k], sh = sh_p[k], mu = mu_p[k], rs = rs_p[k];
        float xh[16], xn[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int b = rowmap(r, hi);
            if (b < n_b) {
                float xv = (float)X[(int64_t)rows_l[b] * pitch + k];
                xh[r] = fmaf(xv, sc, sh);
                xn[r] = (xv - mu) * rs;
            } else {
                xh[r] = 0.f;
                xn[r] = 0.f;
            }
        }
        f32x16 dx = {0};
        for (int ht = w; ht < NHT; ht += 4) {
            const int64_t base = ((int64_t)kt * NHT + ht) * 1024;
            f32x4* wp = reinterpret_cast<f32x4*>(w1s + base);
            f32x4* mp = reinterpret_cast<f32x4*>(m1s + base);
            f32x4* vp = reinterpret_cast<f32x4*>(v1s + base);
            f32x4 wq[4], mq[4], vq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wq[q] = wp[q * 64 + lane];
                mq[q] = mp[q * 64 + lane];
                vq[q] = vp[q * 64 + lane];
            }
            // dW^T tile: D[i = unit][j = SNP], contraction over batch rows b = rowmap(s, hi)
            f32x16 g = {0};
#pragma unroll
            for (int s = 0; s < 16; ++s) g = mfma32(dzl[rowmap(s, hi) * PZ + ht * 32 + jl], xh[s], g);
            // dxhat tile: D[i = row b][j = SNP], contraction over units h = ht*32 + rowmap(s, hi)
#pragma unroll
            for (int s = 0; s < 16; ++s)
                dx = mfma32(dzl[jl * PZ + ht * 32 + rowmap(s, hi)], wq[s >> 2][s & 3], dx);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float wv = wq[q][c], mv = mq[q][c], vv = vq[q][c];
                    adam_update(wv, mv, vv, g[q * 4 + c], alpha);
                    wq[q][c] = wv; mq[q][c] = mv; vq[q][c] = vv;
                }
                wp[q * 64 + lane] = wq[q];
                mp[q * 64 + lane] = mq[q];
                vp[q * 64 + lane] = vq[q];
            }
        }
        // reduce dxhat over the 4 waves, then dgamma = sum_b dxhat*xn, dbeta = sum_b dxhat
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(w * 16 + r) * 64 + lane] = dx[r];
        __syncthreads();
        float pg = 0.f, pb = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((r >> 2) == w) {  // wave-uniform
                float d = (part[(0 * 16 + r) * 64 + lane] + part[(1 * 16 + r) * 64 + lane]) +
                          (part[(2 * 16 + r) * 64 + lane] + part[(3 * 16 + r) * 64 + lane]);
                pg = fmaf(d, xn[r], pg);
                pb += d;
            }
        }
        red2[(w * 2 + 0) * 64 + lane] = pg;
        red2[(w * 2 + 1) * 64 + lane] = pb;
        __syncthreads();
        if (w == 0) {
            float dg = (red2[(0 * 2 + 0) * 64 + lane] + red2[(1 * 2 + 0) * 64 + lane]) +
                       (red2[(2 * 2 + 0) * 64 + lane] + red2[(3 * 2 + 0) * 64 + lane]);
            float db = (red2[(0 * 2 + 1) * 64 + lane] + red2[(1 * 2 + 1) * 64 + lane]) +
                       (red2[(2 * 2 + 1) * 64 + lane] + red2[(3 * 2 + 1) * 64 + lane]);
            dg += __shfl_xor(dg, 32);
            db += __shfl_xor(db, 32);
            if (hi == 0 && k < K) {
                float wv = gamma[k], mv = m_gamma[k], vv = v_gamma[k];
                adam_update(wv, mv, vv, dg, alpha);
                gamma[k] = wv; m_gamma[k] = mv; v_gamma[k] = vv;
                wv = beta[k]; mv = m_beta[k]; vv = v_beta[k];
                adam_update(wv, mv, vv, db, alpha);
                beta[k] = wv; m_beta[k] = mv; v_beta[k] = vv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
template <typename F>
static int set_max_lds(F* func, size_t bytes) {
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

extern "C" int loc_bn_batch_stats(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, int K, int Kp,
                                  const float* gamma, const float* beta, float* mov_mean, float* mov_var,
                                  float* out4, void* stream) {
    if (n_b < 1 || n_b > LOC_ROWS) { loc_set_error("loc_bn_batch_stats: n_b=%d out of 1..32", n_b); return -1; }
    hipLaunchKernelGGL(bn_batch_stats_kernel, dim3((Kp + 255) / 256), dim3(256), 0, (hipStream_t)stream, X, x_pitch,
                       rows, n_b, K, Kp, gamma, beta, mov_mean, mov_var, out4);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_bn_infer_scale_shift(int K, int Kp, const float* gamma, const float* beta, const float* mov_mean,
                                        const float* mov_var, float* out4, void* stream) {
    hipLaunchKernelGGL(bn_infer_scale_shift_kernel, dim3((Kp + 255) / 256), dim3(256), 0, (hipStream_t)stream, K, Kp,
                       gamma, beta, mov_mean, mov_var, out4);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_l1_forward(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, const loc_dims* d,
                              const float* scale_shift, const float* w1s, const float* b1, float* partial, int grid,
                              float* a1, float* a1_drop, const uint8_t* mask, float keep_scale, void* stream) {
    if (n_b < 1 || n_b > LOC_ROWS) { loc_set_error("loc_l1_forward: n_b=%d out of 1..32", n_b); return -1; }
    const int nkt = d->Kp / KT, nht = d->Hp / 32;
    if (grid < 1) grid = 1;
    if (grid > nkt) grid = nkt;
    if (grid > LOC_MAX_FWD_GRID) grid = LOC_MAX_FWD_GRID;
    const size_t lds = 2 * (size_t)(d->Hp + 32) * LP * sizeof(float);
#define LAUNCH_FWD(N)                                                                                      \
    {                                                                                                      \
        static size_t lds_set = 0;                                               \
        if (lds > lds_set) { int rc = set_max_lds(l1_fwd_partial_kernel<N>, lds); if (rc) return rc; lds_set = lds; }                                                                                 \
        hipLaunchKernelGGL(l1_fwd_partial_kernel<N>, dim3(grid), dim3(512), lds, (hipStream_t)stream, X,   \
                           x_pitch, rows, n_b, d->Kp, scale_shift, w1s, partial);                          \
    }
    NHT_SWITCH(nht, LAUNCH_FWD)
#undef LAUNCH_FWD
    LOC_CHECK_LAUNCH();
    hipLaunchKernelGGL(l1_reduce_kernel, dim3((32 * d->Hp + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial,
                       grid, d->Hp, b1, a1, a1_drop, mask, keep_scale);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_l1_backward_adam(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b,
                                    const loc_dims* d, const float* bn4, const float* dz1, float* w1s, float* m1s,
                                    float* v1s, float* gamma, float* beta, float* m_gamma, float* v_gamma,
                                    float* m_beta, float* v_beta, float* b1, float* m_b1, float* v_b1,
                                    const float* alpha_tab, int alpha_tab_len, const float* lr, const int* t_base,
                                    int t_off, int grid, void* stream) {
    if (n_b < 1 || n_b > LOC_ROWS) { loc_set_error("loc_l1_backward_adam: n_b=%d out of 1..32", n_b); return -1; }
    const int nkt = d->Kp / KT, nht = d->Hp / 32;
    if (grid < 1) grid = 1;
    if (grid > nkt) grid = nkt;
    const size_t lds = ((size_t)32 * (d->Hp + 1) + 4 * 16 * 64 + 4 * 2 * 64 + 32) * sizeof(float);
#define LAUNCH_BWD(N)                                                                                          \
    {                                                                                                          \
        static size_t lds_set = 0;                                                      \
        if (lds > lds_set) { int rc = set_max_lds(l1_bwd_adam_kernel<N>, lds); if (rc) return rc; lds_set = lds; }                                                                                     \
        hipLaunchKernelGGL(l1_bwd_adam_kernel<N>, dim3(grid), dim3(256), lds, (hipStream_t)stream, X, x_pitch, \
                           rows, n_b, d->K, d->Kp, bn4, dz1, w1s, m1s, v1s, gamma, beta, m_gamma, v_gamma,     \
                           m_beta, v_beta, b1, m_b1, v_b1, alpha_tab, alpha_tab_len, lr, t_base, t_off);       \
    }
    NHT_SWITCH(nht, LAUNCH_BWD)
#undef LAUNCH_BWD
    LOC_CHECK_LAUNCH();
    return 0;
}
