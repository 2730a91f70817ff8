// Layer 1 of locator's network on gfx950: BatchNormalization on the genotype
// matrix + Dense(width, elu) (reference: /root/reference/locator/locator.py:318-320),
// forward and fused backward+Adam.  All arithmetic fp32; contractions on
// v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#define KT 32  // SNPs per k-tile
#define LP 36  // LDS pitch (floats) of a [row][32 k] tile: 16-B aligned rows, conflict-free ds_read_b128

// ---------------------------------------------------------------------------------------------
// BN batch statistics (training) — SURVEY.md A.2.  One thread per SNP.
// out4 = [scale | shift | mean | rstd], scale = gamma*rstd, shift = beta - mean*scale.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void bn_batch_stats_kernel(
    const uint8_t* __restrict__ X, int64_t pitch, const int32_t* __restrict__ rows, int n_b, int K, int Kp,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mov_mean,
    float* __restrict__ mov_var, float* __restrict__ out4) {
    // one thread per 4 consecutive SNPs: a wave reads 256 contiguous bytes of each batch row
    const int k0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (k0 >= Kp) return;
    int s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
#pragma unroll 8
    for (int b = 0; b < n_b; ++b) {
        uint32_t v = *reinterpret_cast<const uint32_t*>(X + (int64_t)rows[b] * pitch + k0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int x = (v >> (8 * c)) & 255;
            s[c] += x;
            ss[c] += x * x;
        }
    }
    f32x4 o_sc = {0, 0, 0, 0}, o_sh = {0, 0, 0, 0}, o_mu = {0, 0, 0, 0}, o_rs = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = k0 + c;
        if (k < K) {
            float mean = (float)s[c] / (float)n_b;
            // biased variance, exact integer numerator: (n*sum(x^2) - sum(x)^2) / n^2
            float var = (float)(n_b * ss[c] - s[c] * s[c]) / (float)(n_b * n_b);
            float rstd = 1.0f / sqrtf(var + BN_EPS);
            float scale = gamma[k] * rstd;
            o_sc[c] = scale;
            o_sh[c] = beta[k] - mean * scale;
            o_mu[c] = mean;
            o_rs[c] = rstd;
            mov_mean[k] = mov_mean[k] * BN_MOMENTUM + mean * (1.0f - BN_MOMENTUM);
            mov_var[k] = mov_var[k] * BN_MOMENTUM + var * (1.0f - BN_MOMENTUM);
        }
    }
    *reinterpret_cast<f32x4*>(out4 + k0) = o_sc;
    *reinterpret_cast<f32x4*>(out4 + Kp + k0) = o_sh;
    *reinterpret_cast<f32x4*>(out4 + 2 * (int64_t)Kp + k0) = o_mu;
    *reinterpret_cast<f32x4*>(out4 + 3 * (int64_t)Kp + k0) = o_rs;
}

// ---------------------------------------------------------------------------------------------
// BN batch statistics for EVERY minibatch of an epoch in one launch: they depend only on the genotype
// matrix and the epoch's permutation, not on any weight.  Grid (SNP quads, steps).
// stats_ep[step] = [mean | biased var] (2*Kp floats).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void bn_epoch_stats_kernel(const uint8_t* __restrict__ X, int64_t pitch,
                                                             const int32_t* __restrict__ rows_all, int batch,
                                                             int n_last, int n_steps, int K, int Kp,
                                                             float* __restrict__ stats_ep) {
    const int k0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int step = blockIdx.y;
    if (k0 >= Kp) return;
    const int n_b = step == n_steps - 1 ? n_last : batch;
    const int32_t* rows = rows_all + (int64_t)step * batch;
    int s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
#pragma unroll 8
    for (int b = 0; b < n_b; ++b) {
        uint32_t v = *reinterpret_cast<const uint32_t*>(X + (int64_t)rows[b] * pitch + k0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int x = (v >> (8 * c)) & 255;
            s[c] += x;
            ss[c] += x * x;
        }
    }
    f32x4 mu = {0, 0, 0, 0}, var = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (k0 + c < K) {
            mu[c] = (float)s[c] / (float)n_b;
            // exact integer numerator; 64-bit, because n_b * ss reaches 2^31 from n_b = 182 rows of 255s on
            var[c] = (float)((int64_t)n_b * ss[c] - (int64_t)s[c] * s[c]) / ((float)n_b * (float)n_b);
        }
    }
    float* o = stats_ep + (int64_t)step * 2 * Kp;
    *reinterpret_cast<f32x4*>(o + k0) = mu;
    *reinterpret_cast<f32x4*>(o + Kp + k0) = var;
}

// Moving-statistics recurrence over the epoch's steps (same order and arithmetic as the per-step
// update) and bn4 = [scale|shift|mean|rstd] for step 0 from the current gamma/beta.
__global__ void bn_epoch_finish_kernel(int K, int Kp, int n_steps, const float* __restrict__ stats_ep,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ mov_mean, float* __restrict__ mov_var,
                                       float* __restrict__ bn4) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Kp) return;
    float scale = 0.f, shift = 0.f, mean0 = 0.f, rstd0 = 0.f;
    if (k < K) {
        float mm = mov_mean[k], mv = mov_var[k];
        for (int j = 0; j < n_steps; ++j) {
            const float mu = stats_ep[(int64_t)j * 2 * Kp + k], var = stats_ep[(int64_t)j * 2 * Kp + Kp + k];
            mm = mm * BN_MOMENTUM + mu * (1.0f - BN_MOMENTUM);
            mv = mv * BN_MOMENTUM + var * (1.0f - BN_MOMENTUM);
        }
        mov_mean[k] = mm;
        mov_var[k] = mv;
        mean0 = stats_ep[k];
        rstd0 = 1.0f / sqrtf(stats_ep[Kp + k] + BN_EPS);
        scale = gamma[k] * rstd0;
        shift = beta[k] - mean0 * scale;
    }
    if (bn4 == nullptr) return;            // moving statistics only (step 0's scale / shift came with a chained forward)
    bn4[k] = scale;
    bn4[Kp + k] = shift;
    bn4[2 * (int64_t)Kp + k] = mean0;
    bn4[3 * (int64_t)Kp + k] = rstd0;
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
                                                             float* __restrict__ partial,
                                                             const uint8_t* __restrict__ in_mask, float in_ks) {
    // in_mask != NULL: Dropout sits directly on the BatchNorm output (--nlayers 1, locator.py:319-323): keep flags
    // [32][Kp], xhat -> xhat * mask * in_ks before the contraction
    constexpr int Hp = NHT * 32;
    constexpr int WF4 = NHT * 256;               // float4 per weight tile
    constexpr int NLD = (WF4 + 511) / 512;       // float4 loads per thread per tile
    constexpr int BUF = (Hp + 32) * LP;          // floats per LDS buffer: W rows then xhat rows
    constexpr int NBUF = NHT > 16 ? 1 : 2;       // widths above 512: the tile no longer fits twice (2 x 152 KB): one buffer
    constexpr int NTW = (NHT + 7) / 8;           // unit tiles per wave: w, w + 8, ...
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
    uint32_t xreg = 0, mreg = 0x01010101u;
    f32x4 screg = {0, 0, 0, 0}, shreg = {0, 0, 0, 0};

    auto load_regs = [&](int kt) {
        const f32x4* src = reinterpret_cast<const f32x4*>(w1s + (int64_t)kt * NHT * 1024);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int f = t + 512 * i;
            if (f < WF4) wreg[i] = src[f];   // default cache policy on purpose, see l1_bwd_adam_kernel
        }
        if (xvalid) {
            int k0 = kt * KT + 4 * xq;
            xreg = *reinterpret_cast<const uint32_t*>(X + xrow + k0);
            screg = *reinterpret_cast<const f32x4*>(scale + k0);
            shreg = *reinterpret_cast<const f32x4*>(shift + k0);
            if (in_mask) mreg = *reinterpret_cast<const uint32_t*>(in_mask + (int64_t)xb * Kp + k0);
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
                if (in_mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ((mreg >> (8 * e)) & 255u) ? v[e] * in_ks : 0.f;
                }
            }
            *reinterpret_cast<f32x4*>(xh + xb * LP + 4 * xq) = v;
        }
    };

    f32x16 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = f32x16{0};

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
        if (w < NHT) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 a = *reinterpret_cast<const f32x4*>(xh + jl * LP + 8 * m + 4 * hi);
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    if (w + 8 * j < NHT) {
                        f32x4 b = *reinterpret_cast<const f32x4*>(Wl + ((w + 8 * j) * 32 + jl) * LP + 8 * m + 4 * hi);
                        acc[j] = mfma32(a[0], b[0], acc[j]);
                        acc[j] = mfma32(a[1], b[1], acc[j]);
                        acc[j] = mfma32(a[2], b[2], acc[j]);
                        acc[j] = mfma32(a[3], b[3], acc[j]);
                    }
                }
            }
        }
        if (NBUF == 1) {
            __syncthreads();                     // everyone is done with the only buffer before it is refilled
            if (has_next) store_lds(smem);
            __syncthreads();
        } else {
            if (has_next) store_lds(smem + (cur ^ 1) * BUF);
            __syncthreads();
            cur ^= 1;
        }
    }
    // D[i = row b][j = unit]: lane holds unit jl, rows rowmap(r, hi)
    float* pout = partial + (int64_t)blockIdx.x * 32 * Hp;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        if (w + 8 * j < NHT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pout[rowmap(r, hi) * Hp + (w + 8 * j) * 32 + jl] = acc[j][r];
        }
    }
}

// a1[b][h] = ELU(sum_g partial[g][b][h] + b1[h]); optional Dropout on this layer's output.
// Block = 256 threads = 16 consecutive float4 outputs x 16 groups of partials; a thread has its 16-byte loads of 16
// groups in flight at once (the partials sit in other XCDs' L2 or in the Infinity Cache: one round trip per 16 groups
// instead of one per two), the 16 group sums are then added through LDS in a fixed order.
__global__ __launch_bounds__(256) void l1_reduce_kernel(const float* __restrict__ partial, int G, int rows_p,
                                                        int Hp, const float* __restrict__ b1,
                                                        float* __restrict__ a1,
                                                        float* __restrict__ a1_drop,
                                                        const uint8_t* __restrict__ mask, float keep_scale) {
    __shared__ f32x4 red[16][16];
    const int o = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int64_t idx = ((int64_t)blockIdx.x * 16 + o) * 4;
    const int64_t n = (int64_t)rows_p * Hp;
    f32x4 s = {0, 0, 0, 0};
    int g = q;
    for (; g + 15 * 16 < G; g += 256) {
        f32x4 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = *reinterpret_cast<const f32x4*>(partial + (int64_t)(g + 16 * e) * n + idx);
#pragma unroll
        for (int e = 0; e < 16; ++e) s = s + v[e];
    }
    for (; g < G; g += 16) s = s + *reinterpret_cast<const f32x4*>(partial + (int64_t)g * n + idx);
    red[q][o] = s;
    __syncthreads();
    if (q == 0) {
        f32x4 z = red[0][o];
#pragma unroll
        for (int j = 1; j < 16; ++j) z = z + red[j][o];
        const f32x4 bias = *reinterpret_cast<const f32x4*>(b1 + idx % Hp);
        f32x4 a;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = elu_f(z[e] + bias[e]);
        *reinterpret_cast<f32x4*>(a1 + idx) = a;
        if (mask) {
            const uint32_t m4 = *reinterpret_cast<const uint32_t*>(mask + idx);
            f32x4 d;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = ((m4 >> (8 * e)) & 255u) ? a[e] * keep_scale : 0.f;
            *reinterpret_cast<f32x4*>(a1_drop + idx) = d;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Layer-1 backward + Adam, one pass over W1/m/v (24 B per weight), nothing else materialised.
//
// Work unit = one (k-tile, unit-tile) pair = 1024 weights = three contiguous 4 KB runs (W, m, v) in
// the swizzled layout.  The units, ordered k-tile major, are cut into one contiguous range per WAVE;
// a wave streams its range with 16-byte loads straight into the MFMA accumulator layout, requesting
// the next unit's 12 KB into a second register set before it works on the current one (round 3 found that the
// compiler's wait counts, merged conservatively around the loop, make the first use of the CURRENT set wait for most
// of that request, so little overlaps inside a wave; the kernel is HBM-bound through its 8 waves per CU regardless --
// l1_chain.hip, which needed the overlap, uses untracked loads and a hand-counted wait instead):
//     dW^T[h][k]  = sum_b dZ[b][h] xhat[b][k]      (A from LDS, B = xhat registers)
//     dxhat[b][k] += sum_h dZ[b][h] W[h][k]        (A from LDS, B = the weight registers)
// then Adam in registers and 16-byte stores.  Cache policy (template mask NTM: 1 = m,v loads, 8 = m,v
// stores, 2 = W loads, 4 = W stores non-temporal): the Adam moments are touched exactly once per step, so
// they stream non-temporally and stop evicting W1 (102 MB of the 256 MB Infinity Cache), which the next
// forward and this kernel both re-read.  Measured (samples/s, 2 runs each): NTM 0 153.4k, 1 154.7k,
// 8 156.1k, 9 157.9k, 13 (default) 158.0k; non-temporal W loads in the FORWARD cost 5-7 %
// (backward 101 -> 110-114 us), so W loads keep the default policy everywhere.
// There is no barrier and no cross-wave traffic in the
// loop.  dxhat is only needed for BN's trainable gamma/beta (locator.py:318): each wave folds its
// share into (sum_b dxhat*xn, sum_b dxhat) per SNP and leaves it in `gbs`; a k-tile is touched by at
// most two waves (ranges are >= NHT units), slot 0 = the wave that did unit-tile 0, slot 1 = the
// other.  l1_gamma_beta_adam_kernel then adds the two slots in a fixed order and applies Adam.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_update_fast(float& w, float& m, float& v, float g, float alpha) {
    m = m + (g - m) * ADAM_C1;
    v = v + (g * g - v) * ADAM_C2;
    // v_sqrt_f32 / v_rcp_f32: 1 ulp each, i.e. < 4e-7 relative on an update that is <= lr
    w = w - (m * alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + ADAM_EPS);
}

template <int NHT, int NTM = 13, bool INDROP = false>
__global__ __launch_bounds__(256, 2) void l1_bwd_adam_kernel(
    const uint8_t* __restrict__ X, int64_t pitch, const int32_t* __restrict__ rows, int n_b, int K, int Kp,
    const float* __restrict__ bn4, const float* __restrict__ dz1, float* __restrict__ w1s, float* __restrict__ m1s,
    float* __restrict__ v1s, float* __restrict__ gbs, float* __restrict__ b1, float* __restrict__ m_b1,
    float* __restrict__ v_b1, const float* __restrict__ alpha_tab, int alpha_tab_len,
    const float* __restrict__ lr, const int* __restrict__ t_base, int t_off, int n_active,
    const uint8_t* __restrict__ in_mask, float in_ks) {
    // INDROP: Dropout directly on the BatchNorm output (--nlayers 1): keep flags in_mask [32][Kp].  The layer's input is
    // xhat * mk (mk = mask * in_ks), so dW uses xhat * mk and the gradient reaching BatchNorm is dxhat * mk.
    constexpr int Hp = NHT * 32;
    constexpr int PZ = Hp + 1;  // dZ pitch: lanes<->rows reads hit distinct banks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzl = smem;                                        // [32][PZ]
    int* rows_l = reinterpret_cast<int*>(dzl + 32 * PZ);      // [32]

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

    const int gw = blockIdx.x * 4 + w;
    if (gw >= n_active) return;
    const int64_t U = (int64_t)nkt * NHT;
    const int u0 = (int)((int64_t)gw * U / n_active), u1 = (int)((int64_t)(gw + 1) * U / n_active);

    const float* sc_p = bn4;
    const float* sh_p = bn4 + Kp;
    const float* mu_p = bn4 + 2 * (int64_t)Kp;
    const float* rs_p = bn4 + 3 * (int64_t)Kp;

    float xh[16], xn[16];
    float mk[INDROP ? 16 : 1];
    f32x16 dx = {0};
    int cur_kt = -1, first_ht = 0;

    auto flush = [&](int kt, int last_ht) {
        float pg = 0.f, pb = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pg = fmaf(dx[r], xn[r], pg);            // INDROP: xn already carries mk
            if (INDROP) pb = fmaf(dx[r], mk[r], pb);
            else pb += dx[r];
        }
        pg += __shfl_xor(pg, 32);
        pb += __shfl_xor(pb, 32);
        if (hi == 0) {
            float* g0 = gbs + (int64_t)kt * 128;   // [slot][2][32]
            if (first_ht == 0) {
                g0[jl] = pg; g0[32 + jl] = pb;
                if (last_ht == NHT - 1) { g0[64 + jl] = 0.f; g0[96 + jl] = 0.f; }
            } else {
                g0[64 + jl] = pg; g0[96 + jl] = pb;
            }
        }
    };
    auto load_unit = [&](int u, f32x4 (&wq)[4], f32x4 (&mq)[4], f32x4 (&vq)[4]) {
        const int64_t base = (int64_t)u * 1024;
        const f32x4* wp = reinterpret_cast<const f32x4*>(w1s + base);
        const f32x4* mp = reinterpret_cast<const f32x4*>(m1s + base);
        const f32x4* vp = reinterpret_cast<const f32x4*>(v1s + base);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wq[q] = (NTM & 2) ? __builtin_nontemporal_load(wp + q * 64 + lane) : wp[q * 64 + lane];
            mq[q] = (NTM & 1) ? __builtin_nontemporal_load(mp + q * 64 + lane) : mp[q * 64 + lane];
            vq[q] = (NTM & 1) ? __builtin_nontemporal_load(vp + q * 64 + lane) : vp[q * 64 + lane];
        }
    };
    auto step = [&](int u, f32x4 (&wq)[4], f32x4 (&mq)[4], f32x4 (&vq)[4], f32x4 (&wn)[4], f32x4 (&mn)[4],
                    f32x4 (&vn)[4]) {
        const int kt = u / NHT, ht = u - kt * NHT;
        if (kt != cur_kt) {   // wave-uniform
            if (cur_kt >= 0) flush(cur_kt, NHT - 1);
            const int k = kt * KT + jl;
            const float sc = sc_p[k], sh = sh_p[k], mu = mu_p[k], rs = rs_p[k];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int b = rowmap(r, hi);
                float xv = (float)X[(int64_t)rows_l[b] * pitch + k];
                bool ok = b < n_b;
                xh[r] = ok ? fmaf(xv, sc, sh) : 0.f;
                xn[r] = ok ? (xv - mu) * rs : 0.f;
                if (INDROP) {
                    mk[r] = (ok && in_mask[(int64_t)b * Kp + k]) ? in_ks : 0.f;
                    xh[r] *= mk[r];
                    xn[r] *= mk[r];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[r] = 0.f;
            cur_kt = kt;
            first_ht = ht;
        }
        if (u + 1 < u1) load_unit(u + 1, wn, mn, vn);
        // dW^T tile: D[i = unit][j = SNP], contraction over batch rows b = rowmap(s, hi)
        f32x16 g = {0};
#pragma unroll
        for (int s = 0; s < 16; ++s) g = mfma32(dzl[rowmap(s, hi) * PZ + ht * 32 + jl], xh[s], g);
        // dxhat tile: D[i = row b][j = SNP], contraction over units h = ht*32 + rowmap(s, hi)
#pragma unroll
        for (int s = 0; s < 16; ++s) dx = mfma32(dzl[jl * PZ + ht * 32 + rowmap(s, hi)], wq[s >> 2][s & 3], dx);
        const int64_t base = (int64_t)u * 1024;
        f32x4* wp = reinterpret_cast<f32x4*>(w1s + base);
        f32x4* mp = reinterpret_cast<f32x4*>(m1s + base);
        f32x4* vp = reinterpret_cast<f32x4*>(v1s + base);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float wv = wq[q][c], mv = mq[q][c], vv = vq[q][c];
                adam_update_fast(wv, mv, vv, g[q * 4 + c], alpha);
                wq[q][c] = wv; mq[q][c] = mv; vq[q][c] = vv;
            }
            if (NTM & 4) __builtin_nontemporal_store(wq[q], wp + q * 64 + lane); else wp[q * 64 + lane] = wq[q];
            if (NTM & 8) __builtin_nontemporal_store(mq[q], mp + q * 64 + lane); else mp[q * 64 + lane] = mq[q];
            if (NTM & 8) __builtin_nontemporal_store(vq[q], vp + q * 64 + lane); else vp[q * 64 + lane] = vq[q];
        }
    };

    if (u0 < u1) {
        f32x4 wA[4], mA[4], vA[4], wB[4], mB[4], vB[4];
        load_unit(u0, wA, mA, vA);
        for (int u = u0; u < u1; u += 2) {
            step(u, wA, mA, vA, wB, mB, vB);
            if (u + 1 < u1) step(u + 1, wB, mB, vB, wA, mA, vA);
        }
        flush(cur_kt, (u1 - 1) % NHT);
    }
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>), so that register arrays
// indexed by the row block never degrade to scratch memory when the unroller gives up
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// ---------------------------------------------------------------------------------------------
// Layer-1 backward + Adam for MORE than 32 rows (--batch_size 33..128: RB = 2..4 row blocks of 32).
//
// With RB row blocks the fp32-MFMA formulation above needs 32*RB MFMAs of 64 cycles per 12 KB of weight
// stream and turns matrix-bound at RB >= 3.  This kernel removes almost all of that work, exactly:
//   * xhat = s_k (x - c_k) + (t_k + s_k c_k) with x a uint8 genotype and c_k = rint(batch mean), so
//         dW^T[h][k] = s_k S[h][k] + (t_k + s_k c_k) dzsum[h],  S[h][k] = sum_b dz[b][h] (x[b][k] - c_k),
//         dzsum[h] = sum_b dz[b][h].
//     x - c is a small integer, exact in bf16, and dz splits exactly into three bf16 pieces (truncation, 8+8+8
//     significand bits), so S is three v_mfma_f32_32x32x16_bf16 per 16 rows with exact products and fp32
//     accumulation; centring on c keeps both terms at the size of the result (no cancellation);
//   * BatchNorm's gamma/beta gradients need sum_b dxhat*xn and sum_b dxhat with dxhat = dz W^T; pushing the sum
//     over b inside gives   sum_h W[h][k] * rstd_k (S[h][k] - (mu_k - c_k) dzsum[h])   and   sum_h W[h][k] dzsum[h]
//     -- per-lane reductions over the accumulator the wave already holds (lane = SNP k), so dxhat is never formed
//     and its MFMAs disappear.
// Per unit: 6*RB bf16 MFMAs of 32 cycles instead of 32*RB fp32 MFMAs of 64.  dz lives in LDS transposed,
// [unit][row] fp32 (33 KB per row block at width 256), so RB = 2 keeps two 4-wave workgroups per CU and
// RB = 3, 4 run one 8-wave workgroup.  Same work partition, cache policy, gamma/beta hand-off (gbs slots) and
// Adam arithmetic as l1_bwd_adam_kernel.
// ---------------------------------------------------------------------------------------------
template <int NHT, int NTM, int RB>
__global__ __launch_bounds__(RB > 2 ? 512 : 256, RB > 2 ? 1 : 2) void l1_bwd_adam_rows_kernel(
    const uint8_t* __restrict__ X, int64_t pitch, const int32_t* __restrict__ rows, int n_b, int K, int Kp,
    const float* __restrict__ bn4, const float* __restrict__ dz1, float* __restrict__ w1s, float* __restrict__ m1s,
    float* __restrict__ v1s, float* __restrict__ gbs, float* __restrict__ b1, float* __restrict__ m_b1,
    float* __restrict__ v_b1, const float* __restrict__ alpha_tab, int alpha_tab_len,
    const float* __restrict__ lr, const int* __restrict__ t_base, int t_off, int n_active) {
    constexpr int Hp = NHT * 32;
    constexpr int NR = 32 * RB;                 // rows the workgroup is built for
    constexpr int PT = NR + 4;                  // pitch of dzT rows (floats): 16-B aligned, banks shifted by 4 per unit
    constexpr int NQ = 2 * RB;                  // 16-row MFMA blocks
    constexpr int NT = RB > 2 ? 512 : 256, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzT = smem;                                              // [Hp][PT]   dz transposed: [unit][row]
    float* dzsum = dzT + Hp * PT;                                   // [Hp]
    int* rows_l = reinterpret_cast<int*>(dzsum + Hp);               // [NR]

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int jl = lane & 31, hi = lane >> 5;
    const int nkt = Kp / KT;
    // the launcher picks RB = ceil(n_b / 32), so all NR rows are in use; dz rows n_b..NR-1 are zero
    const float alpha = adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off);

    // dz1 is [row][unit]; the MFMA A operand wants [unit][row].  Lanes run along the rows (conflict-free LDS
    // stores; the 16-byte global reads of different rows all hit L2, dz1 is 32*RB KB)
    for (int i = t; i < NR * (Hp / 4); i += NT) {
        const int r = i % NR, hq = i / NR;
        const f32x4 v = *reinterpret_cast<const f32x4*>(dz1 + (int64_t)r * Hp + 4 * hq);
        dzT[(4 * hq + 0) * PT + r] = v[0];
        dzT[(4 * hq + 1) * PT + r] = v[1];
        dzT[(4 * hq + 2) * PT + r] = v[2];
        dzT[(4 * hq + 3) * PT + r] = v[3];
    }
    if (t < NR) rows_l[t] = t < n_b ? rows[t] : 0;
    __syncthreads();
    for (int h = t; h < Hp; h += NT) {          // 16-byte reads: one unit per lane, PT = 4 mod 64 -> conflict-free
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int r = 0; r < NR; r += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(dzT + h * PT + r);
            s4[0] += v[0]; s4[1] += v[1]; s4[2] += v[2]; s4[3] += v[3];
        }
        dzsum[h] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    __syncthreads();

    // bias of layer 1: db1[h] = sum_b dZ[b][h]   (block 0 only)
    if (blockIdx.x == 0) {
        for (int h = t; h < Hp; h += NT) {
            float wv = b1[h], mv = m_b1[h], vv = v_b1[h];
            adam_update(wv, mv, vv, dzsum[h], alpha);
            b1[h] = wv; m_b1[h] = mv; v_b1[h] = vv;
        }
    }

    const int gw = blockIdx.x * NW + w;
    if (gw >= n_active) return;
    const int64_t U = (int64_t)nkt * NHT;
    const int u0 = (int)((int64_t)gw * U / n_active), u1 = (int)((int64_t)(gw + 1) * U / n_active);

    const float* sc_p = bn4;
    const float* sh_p = bn4 + Kp;
    const float* mu_p = bn4 + 2 * (int64_t)Kp;
    const float* rs_p = bn4 + 3 * (int64_t)Kp;

    // genotypes of the current k-tile for this lane's SNP as the MFMA B operand: block q holds rows
    // 16q + 8hi + 0..7 as eight bf16 values
    u32x4 xq[NQ];
    float k_sc = 0.f, k_sh = 0.f, k_mu = 0.f, k_rs = 0.f, k_c = 0.f, pg = 0.f, pb = 0.f;
    int cur_kt = -1, first_ht = 0;

    auto flush = [&](int kt, int last_ht) __attribute__((always_inline)) {
        float g_ = pg * k_rs, b_ = pb;
        g_ += __shfl_xor(g_, 32);
        b_ += __shfl_xor(b_, 32);
        if (hi == 0) {
            float* g0 = gbs + (int64_t)kt * 128;   // [slot][2][32]
            if (first_ht == 0) {
                g0[jl] = g_; g0[32 + jl] = b_;
                if (last_ht == NHT - 1) { g0[64 + jl] = 0.f; g0[96 + jl] = 0.f; }
            } else {
                g0[64 + jl] = g_; g0[96 + jl] = b_;
            }
        }
    };
    auto load_unit = [&](int u, f32x4 (&wq)[4], f32x4 (&mq)[4], f32x4 (&vq)[4]) __attribute__((always_inline)) {
        const int64_t base = (int64_t)u * 1024;
        const f32x4* wp = reinterpret_cast<const f32x4*>(w1s + base);
        const f32x4* mp = reinterpret_cast<const f32x4*>(m1s + base);
        const f32x4* vp = reinterpret_cast<const f32x4*>(v1s + base);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wq[q] = (NTM & 2) ? __builtin_nontemporal_load(wp + q * 64 + lane) : wp[q * 64 + lane];
            mq[q] = (NTM & 1) ? __builtin_nontemporal_load(mp + q * 64 + lane) : mp[q * 64 + lane];
            vq[q] = (NTM & 1) ? __builtin_nontemporal_load(vp + q * 64 + lane) : vp[q * 64 + lane];
        }
    };
    auto step = [&](int u, f32x4 (&wq)[4], f32x4 (&mq)[4], f32x4 (&vq)[4], f32x4 (&wn)[4], f32x4 (&mn)[4],
                    f32x4 (&vn)[4]) __attribute__((always_inline)) {
        const int kt = u / NHT, ht = u - kt * NHT;
        if (kt != cur_kt) {   // wave-uniform
            if (cur_kt >= 0) flush(cur_kt, NHT - 1);
            const int k = kt * KT + jl;
            // Genotypes are centred on the integer nearest the batch mean, c_k = rint(mu_k): x - c_k is still an
            // exact bf16 integer, and xhat = s (x - c) + (t + s c) leaves two terms of the size of the result
            // instead of two large ones that cancel (t = beta - mu s):  k_sh, k_mu hold t + s c and mu - c.
            {
                const float sc = sc_p[k], sh = sh_p[k], mu = mu_p[k];
                const float c = rintf(mu);
                k_sc = sc; k_sh = fmaf(sc, c, sh); k_mu = mu - c; k_rs = rs_p[k];
                k_c = c;
            }
            static_for<NQ>([&](auto QI) __attribute__((always_inline)) {
                constexpr int q = decltype(QI)::value;
                float xv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)      // small integers: exact in bf16
                    xv[i] = (float)X[(int64_t)rows_l[16 * q + 8 * hi + i] * pitch + k] - k_c;
                xq[q][0] = pack_top16(fbits(xv[0]), fbits(xv[1]));
                xq[q][1] = pack_top16(fbits(xv[2]), fbits(xv[3]));
                xq[q][2] = pack_top16(fbits(xv[4]), fbits(xv[5]));
                xq[q][3] = pack_top16(fbits(xv[6]), fbits(xv[7]));
            });
            pg = 0.f; pb = 0.f;
            cur_kt = kt;
            first_ht = ht;
        }
        if (u + 1 < u1) load_unit(u + 1, wn, mn, vn);
        // S[h][k] = sum_b dz[b][h] x[b][k]: D[i = unit][j = SNP], 16 rows per MFMA, three exact bf16 pieces of dz
        f32x16 S = {0};
        const float* arow = dzT + (ht * 32 + jl) * PT + 8 * hi;
        static_for<NQ>([&](auto QI) __attribute__((always_inline)) {
            constexpr int q = decltype(QI)::value;
            {
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(arow + 16 * q);
                const f32x4 d1 = *reinterpret_cast<const f32x4*>(arow + 16 * q + 4);
                float v[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
                const bf16x8 b = __builtin_bit_cast(bf16x8, xq[q]);
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    u32x4 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t lo = fbits(v[2 * e]), hh = fbits(v[2 * e + 1]);
                        a[e] = pack_top16(lo, hh);
                        if (p < 2) {
                            v[2 * e] -= bitsf(lo & 0xFFFF0000u);
                            v[2 * e + 1] -= bitsf(hh & 0xFFFF0000u);
                        }
                    }
                    S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b, S, 0, 0, 0);
                }
            }
        });
        const int64_t base = (int64_t)u * 1024;
        f32x4* wp = reinterpret_cast<f32x4*>(w1s + base);
        f32x4* mp = reinterpret_cast<f32x4*>(m1s + base);
        f32x4* vp = reinterpret_cast<f32x4*>(v1s + base);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // accumulator rows 4q..4q+3 are units ht*32 + 8q + 4hi + 0..3: their dz sums are one 16-byte read
            const f32x4 ds4 = *reinterpret_cast<const f32x4*>(dzsum + ht * 32 + 8 * q + 4 * hi);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float s_ = S[q * 4 + c], ds = ds4[c];
                float wv = wq[q][c], mv = mq[q][c], vv = vq[q][c];
                pg = fmaf(wv, fmaf(-k_mu, ds, s_), pg);          // sum_h W (S - mu dzsum); rstd applied at the flush
                pb = fmaf(wv, ds, pb);
                const float g = fmaf(k_sc, s_, k_sh * ds);
                adam_update_fast(wv, mv, vv, g, alpha);
                wq[q][c] = wv; mq[q][c] = mv; vq[q][c] = vv;
            }
            if (NTM & 4) __builtin_nontemporal_store(wq[q], wp + q * 64 + lane); else wp[q * 64 + lane] = wq[q];
            if (NTM & 8) __builtin_nontemporal_store(mq[q], mp + q * 64 + lane); else mp[q * 64 + lane] = mq[q];
            if (NTM & 8) __builtin_nontemporal_store(vq[q], vp + q * 64 + lane); else vp[q * 64 + lane] = vq[q];
        }
    };

    if (u0 < u1) {
        f32x4 wA[4], mA[4], vA[4], wB[4], mB[4], vB[4];
        load_unit(u0, wA, mA, vA);
        for (int u = u0; u < u1; u += 2) {
            step(u, wA, mA, vA, wB, mB, vB);
            if (u + 1 < u1) step(u + 1, wB, mB, vB, wA, mA, vA);
        }
        flush(cur_kt, (u1 - 1) % NHT);
    }
}

// ---------------------------------------------------------------------------------------------
// Layer-1 backward + Adam for MORE than 128 rows (--batch_size 129..LOC_BIG_BATCH_MAX, accepted by the reference,
// locator.py:122): the dz image of l1_bwd_adam_rows_kernel no longer fits the LDS, so this variant keeps its algebra
// (S[h][k] = sum_b dz[b][h] (x[b][k] - c_k), dW and the gamma/beta sums derived from S and dzsum, dxhat never formed)
// but streams dz from L2 row block by row block: per unit and 32-row block 16 fp32 MFMAs (v_mfma_f32_32x32x2_f32:
// exact fp32 products) whose A operand is a coalesced 128-byte read of dz and whose B operand is x - c for the
// block's rows.  It is matrix-bound (2 x n_b MFMA cycles per 12 KB of weight stream), not HBM-bound - a correct path
// for a setting far from the default, not a tuned one.  dzsum comes from dz_colsum_kernel.  Same work partition, cache
// policy, gamma/beta hand-off (gbs slots) and Adam arithmetic as the other two kernels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dz_colsum_kernel(const float* __restrict__ dz1, int n_rows, int Hp,
                                                        float* __restrict__ dzsum) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= Hp) return;
    float s = 0.f;
    for (int b = 0; b < n_rows; ++b) s += dz1[(int64_t)b * Hp + h];      // fixed order
    dzsum[h] = s;
}

template <int NHT, int NTM>
__global__ __launch_bounds__(256, 2) void l1_bwd_adam_big_kernel(
    const uint8_t* __restrict__ X, int64_t pitch, const int32_t* __restrict__ rows, int n_b, int K, int Kp,
    const float* __restrict__ bn4, const float* __restrict__ dz1, const float* __restrict__ dzsum_g,
    float* __restrict__ w1s, float* __restrict__ m1s, float* __restrict__ v1s, float* __restrict__ gbs,
    float* __restrict__ b1, float* __restrict__ m_b1, float* __restrict__ v_b1, const float* __restrict__ alpha_tab,
    int alpha_tab_len, const float* __restrict__ lr, const int* __restrict__ t_base, int t_off, int n_active) {
    constexpr int Hp = NHT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzsum = smem;                                            // [Hp]
    int* rows_l = reinterpret_cast<int*>(dzsum + Hp);               // [32 * nrb]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int jl = lane & 31, hi = lane >> 5;
    const int nkt = Kp / KT;
    const int nrb = (n_b + 31) / 32;                                // dz rows n_b..32*nrb-1 are zero (stack kernel)
    const float alpha = adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off);
    for (int h = t; h < Hp; h += 256) dzsum[h] = dzsum_g[h];
    for (int i = t; i < 32 * nrb; i += 256) rows_l[i] = i < n_b ? rows[i] : rows[0];
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int h = t; h < Hp; h += 256) {
            float wv = b1[h], mv = m_b1[h], vv = v_b1[h];
            adam_update(wv, mv, vv, dzsum[h], alpha);
            b1[h] = wv; m_b1[h] = mv; v_b1[h] = vv;
        }
    }
    const int gw = blockIdx.x * 4 + w;
    if (gw >= n_active) return;
    const int64_t U = (int64_t)nkt * NHT;
    const int u0 = (int)((int64_t)gw * U / n_active), u1 = (int)((int64_t)(gw + 1) * U / n_active);
    const float* sc_p = bn4;
    const float* sh_p = bn4 + Kp;
    const float* mu_p = bn4 + 2 * (int64_t)Kp;
    const float* rs_p = bn4 + 3 * (int64_t)Kp;
    float k_sc = 0.f, k_sh = 0.f, k_mu = 0.f, k_rs = 0.f, k_c = 0.f, pg = 0.f, pb = 0.f;
    int cur_kt = -1, first_ht = 0;

    auto flush = [&](int kt, int last_ht) __attribute__((always_inline)) {
        float g_ = pg * k_rs, b_ = pb;
        g_ += __shfl_xor(g_, 32);
        b_ += __shfl_xor(b_, 32);
        if (hi == 0) {
            float* g0 = gbs + (int64_t)kt * 128;   // [slot][2][32]
            if (first_ht == 0) {
                g0[jl] = g_; g0[32 + jl] = b_;
                if (last_ht == NHT - 1) { g0[64 + jl] = 0.f; g0[96 + jl] = 0.f; }
            } else {
                g0[64 + jl] = g_; g0[96 + jl] = b_;
            }
        }
    };
    for (int u = u0; u < u1; ++u) {
        const int kt = u / NHT, ht = u - kt * NHT;
        if (kt != cur_kt) {   // wave-uniform
            if (cur_kt >= 0) flush(cur_kt, NHT - 1);
            const int k = kt * KT + jl;
            const float sc = sc_p[k], sh = sh_p[k], mu = mu_p[k];
            const float c = rintf(mu);
            k_sc = sc; k_sh = fmaf(sc, c, sh); k_mu = mu - c; k_rs = rs_p[k];
            k_c = c;
            pg = 0.f; pb = 0.f;
            cur_kt = kt;
            first_ht = ht;
        }
        const int64_t base = (int64_t)u * 1024;
        f32x4 wq[4], mq[4], vq[4];
        {
            const f32x4* wp = reinterpret_cast<const f32x4*>(w1s + base);
            const f32x4* mp = reinterpret_cast<const f32x4*>(m1s + base);
            const f32x4* vp = reinterpret_cast<const f32x4*>(v1s + base);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wq[q] = wp[q * 64 + lane];
                mq[q] = (NTM & 1) ? __builtin_nontemporal_load(mp + q * 64 + lane) : mp[q * 64 + lane];
                vq[q] = (NTM & 1) ? __builtin_nontemporal_load(vp + q * 64 + lane) : vp[q * 64 + lane];
            }
        }
        // S[h][k] = sum_b dz[b][h] (x[b][k] - c_k): D[i = unit][j = SNP], two rows per MFMA, row blocks in order
        f32x16 S = {0};
        const int k = kt * KT + jl;
        for (int rb = 0; rb < nrb; ++rb) {
            float av[16], bv[16];
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) {
                const int b = rb * 32 + rowmap(s_, hi);
                av[s_] = dz1[(int64_t)b * Hp + ht * 32 + jl];
                bv[s_] = (float)X[(int64_t)rows_l[b] * pitch + k] - k_c;
            }
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) S = mfma32(av[s_], bv[s_], S);
        }
        f32x4* wp = reinterpret_cast<f32x4*>(w1s + base);
        f32x4* mp = reinterpret_cast<f32x4*>(m1s + base);
        f32x4* vp = reinterpret_cast<f32x4*>(v1s + base);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 ds4 = *reinterpret_cast<const f32x4*>(dzsum + ht * 32 + 8 * q + 4 * hi);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float s_ = S[q * 4 + c], ds = ds4[c];
                float wv = wq[q][c], mv = mq[q][c], vv = vq[q][c];
                pg = fmaf(wv, fmaf(-k_mu, ds, s_), pg);          // sum_h W (S - mu dzsum); rstd applied at the flush
                pb = fmaf(wv, ds, pb);
                const float g = fmaf(k_sc, s_, k_sh * ds);
                adam_update_fast(wv, mv, vv, g, alpha);
                wq[q][c] = wv; mq[q][c] = mv; vq[q][c] = vv;
            }
            if (NTM & 4) __builtin_nontemporal_store(wq[q], wp + q * 64 + lane); else wp[q * 64 + lane] = wq[q];
            if (NTM & 8) __builtin_nontemporal_store(mq[q], mp + q * 64 + lane); else mp[q * 64 + lane] = mq[q];
            if (NTM & 8) __builtin_nontemporal_store(vq[q], vp + q * 64 + lane); else vp[q * 64 + lane] = vq[q];
        }
    }
    if (u0 < u1) flush(cur_kt, (u1 - 1) % NHT);
}

// gamma/beta Adam from the per-wave partial sums left by l1_bwd_adam_kernel (fixed order: slot 0 + slot 1).
__global__ void l1_gamma_beta_adam_kernel(int K, const float* __restrict__ gbs, float* __restrict__ gamma,
                                          float* __restrict__ beta, float* __restrict__ m_gamma,
                                          float* __restrict__ v_gamma, float* __restrict__ m_beta,
                                          float* __restrict__ v_beta, const float* __restrict__ alpha_tab,
                                          int alpha_tab_len, const float* __restrict__ lr,
                                          const int* __restrict__ t_base, int t_off, int Kp,
                                          const float* __restrict__ next_stats, float* __restrict__ bn4) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float alpha = adam_alpha(alpha_tab, alpha_tab_len, lr, t_base, t_off);
    gamma_beta_adam_body(k, Kp, gbs, gamma, beta, m_gamma, v_gamma, m_beta, v_beta, alpha, next_stats, bn4);
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
        case 16: MACRO(16); break; case 17: MACRO(17); break; case 18: MACRO(18); break;    \
        case 19: MACRO(19); break; case 20: MACRO(20); break; case 21: MACRO(21); break;    \
        case 22: MACRO(22); break; case 23: MACRO(23); break; case 24: MACRO(24); break;    \
        case 25: MACRO(25); break; case 26: MACRO(26); break; case 27: MACRO(27); break;    \
        case 28: MACRO(28); break; case 29: MACRO(29); break; case 30: MACRO(30); break;    \
        case 31: MACRO(31); break; case 32: MACRO(32); break;                               \
        default: loc_set_error("%s: width %d unsupported (Hp must be 32..1024)", __func__, 32 * (NHT_VALUE)); return -1; \
    }

extern "C" int loc_bn_batch_stats(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, int K, int Kp,
                                  const float* gamma, const float* beta, float* mov_mean, float* mov_var,
                                  float* out4, void* stream) {
    if (n_b < 1 || n_b > LOC_ROWS) { loc_set_error("loc_bn_batch_stats: n_b=%d out of 1..32", n_b); return -1; }
    hipLaunchKernelGGL(bn_batch_stats_kernel, dim3((Kp / 4 + 127) / 128), dim3(128), 0, (hipStream_t)stream, X,
                       x_pitch, rows, n_b, K, Kp, gamma, beta, mov_mean, mov_var, out4);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_bn_epoch_stats(const uint8_t* X, int64_t x_pitch, const int32_t* rows_all, int batch, int n_last,
                                  int n_steps, int K, int Kp, const float* gamma, const float* beta,
                                  float* mov_mean, float* mov_var, float* stats_ep, float* bn4, void* stream) {
    if (batch < 1 || batch > LOC_BIG_BATCH_MAX || n_last < 1 || n_last > batch || n_steps < 1) {
        loc_set_error("loc_bn_epoch_stats: bad batch=%d n_last=%d n_steps=%d", batch, n_last, n_steps);
        return -1;
    }
    hipLaunchKernelGGL(bn_epoch_stats_kernel, dim3((Kp / 4 + 127) / 128, n_steps), dim3(128), 0, (hipStream_t)stream,
                       X, x_pitch, rows_all, batch, n_last, n_steps, K, Kp, stats_ep);
    LOC_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_epoch_finish_kernel, dim3((Kp + 255) / 256), dim3(256), 0, (hipStream_t)stream, K, Kp,
                       n_steps, stats_ep, gamma, beta, mov_mean, mov_var, bn4);
    LOC_CHECK_LAUNCH();
    return 0;
}

// The two halves on their own (round 4: an epoch whose first layer-1 forward was chained into the previous epoch's last
// step computes the NEXT epoch's batch statistics early and applies its own moving-statistics updates without touching
// the scale / shift the chained kernel left): stats only, then the n_steps moving updates (+ step 0's bn4 unless NULL).
extern "C" int loc_bn_epoch_stats_only(const uint8_t* X, int64_t x_pitch, const int32_t* rows_all, int batch, int n_last,
                                       int n_steps, int K, int Kp, float* stats_ep, void* stream) {
    if (batch < 1 || batch > LOC_BIG_BATCH_MAX || n_last < 1 || n_last > batch || n_steps < 1) {
        loc_set_error("loc_bn_epoch_stats_only: bad batch=%d n_last=%d n_steps=%d", batch, n_last, n_steps);
        return -1;
    }
    hipLaunchKernelGGL(bn_epoch_stats_kernel, dim3((Kp / 4 + 127) / 128, n_steps), dim3(128), 0, (hipStream_t)stream,
                       X, x_pitch, rows_all, batch, n_last, n_steps, K, Kp, stats_ep);
    LOC_CHECK_LAUNCH();
    return 0;
}
extern "C" int loc_bn_epoch_finish(int n_steps, int K, int Kp, const float* gamma, const float* beta, float* mov_mean,
                                   float* mov_var, const float* stats_ep, float* bn4, void* stream) {
    if (n_steps < 1) { loc_set_error("loc_bn_epoch_finish: n_steps=%d", n_steps); return -1; }
    hipLaunchKernelGGL(bn_epoch_finish_kernel, dim3((Kp + 255) / 256), dim3(256), 0, (hipStream_t)stream, K, Kp,
                       n_steps, stats_ep, gamma, beta, mov_mean, mov_var, bn4);
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

static int l1_forward_impl(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, const loc_dims* d,
                           const float* scale_shift, const float* w1s, const float* b1, float* partial, int grid,
                           float* a1, float* a1_drop, const uint8_t* mask, float keep_scale, const uint8_t* in_mask,
                           float in_ks, void* stream) {
    if (n_b < 1 || n_b > LOC_ROWS) { loc_set_error("loc_l1_forward: n_b=%d out of 1..32", n_b); return -1; }
    const int nkt = d->Kp / KT, nht = d->Hp / 32;
    if (grid < 1) grid = 1;
    if (grid > nkt) grid = nkt;
    if (grid > LOC_MAX_FWD_GRID) grid = LOC_MAX_FWD_GRID;
    const size_t lds = (nht > 16 ? 1 : 2) * (size_t)(d->Hp + 32) * LP * sizeof(float);
#define LAUNCH_FWD(N)                                                                                      \
    {                                                                                                      \
        LOC_ENSURE_LDS((l1_fwd_partial_kernel<N>), lds);                                                   \
        hipLaunchKernelGGL(l1_fwd_partial_kernel<N>, dim3(grid), dim3(512), lds, (hipStream_t)stream, X,   \
                           x_pitch, rows, n_b, d->Kp, scale_shift, w1s, partial, in_mask, in_ks);          \
    }
    NHT_SWITCH(nht, LAUNCH_FWD)
#undef LAUNCH_FWD
    LOC_CHECK_LAUNCH();
    hipLaunchKernelGGL(l1_reduce_kernel, dim3(32 * d->Hp / 64), dim3(256), 0, (hipStream_t)stream, partial, grid,
                       32, d->Hp, b1, a1, a1_drop, mask, keep_scale);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_l1_forward(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, const loc_dims* d,
                              const float* scale_shift, const float* w1s, const float* b1, float* partial, int grid,
                              float* a1, float* a1_drop, const uint8_t* mask, float keep_scale, void* stream) {
    return l1_forward_impl(X, x_pitch, rows, n_b, d, scale_shift, w1s, b1, partial, grid, a1, a1_drop, mask, keep_scale,
                           nullptr, 1.f, stream);
}
extern "C" int loc_l1_forward_in_dropout(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b,
                                         const loc_dims* d, const float* scale_shift, const float* w1s, const float* b1,
                                         float* partial, int grid, float* a1, const uint8_t* in_mask, float keep_scale,
                                         void* stream) {
    return l1_forward_impl(X, x_pitch, rows, n_b, d, scale_shift, w1s, b1, partial, grid, a1, nullptr, nullptr, 1.f,
                           in_mask, keep_scale, stream);
}

// reduction alone, with the optional Dropout on layer 1's output: the chained training step (l1_chain.hip) leaves
// the partial sums of the NEXT minibatch, whose step then starts here
int loc_l1_reduce_launch_drop(const float* partial, int G, int rows_p, int Hp, const float* b1, float* a1, float* a1_drop,
                              const uint8_t* mask, float keep_scale, void* stream) {
    hipLaunchKernelGGL(l1_reduce_kernel, dim3(rows_p * Hp / 64), dim3(256), 0, (hipStream_t)stream, partial, G,
                       rows_p, Hp, b1, a1, a1_drop, mask, keep_scale);
    LOC_CHECK_LAUNCH();
    return 0;
}

// reduction of the large-M forward (l1_rows.hip): rows_p rows, no dropout
int loc_l1_reduce_launch(const float* partial, int G, int rows_p, int Hp, const float* b1, float* a1, void* stream) {
    hipLaunchKernelGGL(l1_reduce_kernel, dim3(rows_p * Hp / 64), dim3(256), 0, (hipStream_t)stream, partial, G,
                       rows_p, Hp, b1, a1, (float*)nullptr, (const uint8_t*)nullptr, 1.f);
    LOC_CHECK_LAUNCH();
    return 0;
}

// main kernel only (W1 / b1); the gamma/beta update that consumes gb_scratch is launched by the caller
static int l1_backward_main_impl(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b,
                                 const loc_dims* d, const float* bn4, const float* dz1, float* w1s,
                                 float* m1s, float* v1s, float* b1, float* m_b1, float* v_b1,
                                 float* gb_scratch, const float* alpha_tab, int alpha_tab_len,
                                 const float* lr, const int* t_base, int t_off, int grid,
                                 const loc_tuning* tune, const uint8_t* in_mask, float in_ks, void* stream) {
    if (in_mask && n_b > LOC_ROWS) {
        loc_set_error("loc_l1_backward_adam: dropout on the BatchNorm output (--nlayers 1) needs --batch_size <= 32");
        return -1;
    }
    if (n_b < 1 || n_b > LOC_BIG_BATCH_MAX) {
        loc_set_error("loc_l1_backward_adam: n_b=%d out of 1..%d", n_b, LOC_BIG_BATCH_MAX);
        return -1;
    }
    const int nkt = d->Kp / KT, nht = d->Hp / 32;
    if (grid < 1) grid = 1;
    // one contiguous range of >= NHT units per active wave, so a k-tile is split over at most two waves
    int n_active = grid * 4;
    if (n_active > nkt) n_active = nkt;
    grid = (n_active + 3) / 4;
    if (n_b > LOC_MAX_BATCH) {
        // more than 128 rows: row blocks streamed from L2 (l1_bwd_adam_big_kernel); dzsum goes through the first Hp
        // floats of gb_scratch's tail, which is free until the kernel's own flushes (they write [0, 4*Kp))
        if (in_mask) { loc_set_error("loc_l1_backward_adam: --nlayers 1 with dropout needs --batch_size <= 32"); return -1; }
        float* dzsum = gb_scratch + 4 * (int64_t)d->Kp;
        hipLaunchKernelGGL(dz_colsum_kernel, dim3((d->Hp + 255) / 256), dim3(256), 0, (hipStream_t)stream, dz1,
                           (n_b + 31) / 32 * 32, d->Hp, dzsum);
        LOC_CHECK_LAUNCH();
        const size_t lds_big = ((size_t)d->Hp + (size_t)(n_b + 31) / 32 * 32) * sizeof(float);
#define LAUNCH_BWD_BIG(N)                                                                                      \
    hipLaunchKernelGGL((l1_bwd_adam_big_kernel<N, 13>), dim3(grid), dim3(256), lds_big, (hipStream_t)stream, X, x_pitch, \
                       rows, n_b, d->K, d->Kp, bn4, dz1, dzsum, w1s, m1s, v1s, gb_scratch, b1, m_b1, v_b1,     \
                       alpha_tab, alpha_tab_len, lr, t_base, t_off, n_active);
        switch (nht) {
            case 2: LAUNCH_BWD_BIG(2) break;
            case 4: LAUNCH_BWD_BIG(4) break;
            case 8: LAUNCH_BWD_BIG(8) break;
            default: loc_set_error("loc_l1_backward_adam: more than 32 rows need width 64/128/256 after padding (got %d)", d->Hp); return -1;
        }
#undef LAUNCH_BWD_BIG
        LOC_CHECK_LAUNCH();
        return 0;
    }
    const int rb = (n_b + LOC_ROWS - 1) / LOC_ROWS;
    const size_t lds = rb == 1 ? ((size_t)32 * (d->Hp + 1) + 32) * sizeof(float)
                               : ((size_t)d->Hp * (32 * rb + 4) + d->Hp + 32 * rb) * sizeof(float);
    // tune->l1b_rows = 1: the bf16x3 row-block kernel also for <= 32 rows (measurement / parity switch)
    if (rb == 1 && tune && tune->l1b_rows == 1 && nht == 8 && !in_mask) {
        const size_t lds = ((size_t)d->Hp * 36 + d->Hp + 32) * sizeof(float);
        LOC_ENSURE_LDS((l1_bwd_adam_rows_kernel<8, 13, 1>), lds);
        hipLaunchKernelGGL((l1_bwd_adam_rows_kernel<8, 13, 1>), dim3(grid), dim3(256), lds, (hipStream_t)stream, X,
                           x_pitch, rows, n_b, d->K, d->Kp, bn4, dz1, w1s, m1s, v1s, gb_scratch, b1, m_b1, v_b1,
                           alpha_tab, alpha_tab_len, lr, t_base, t_off, n_active);
        LOC_CHECK_LAUNCH();
        return 0;
    }
    if (rb > 1) {
        // more than 32 rows: RB row blocks per weight tile (widths of the fused hidden stack up to 256 only).
        // RB = 2 keeps two workgroups per CU (66 KB of dz each at width 256); RB = 3, 4 run one per CU.
#define LAUNCH_BWD_RB(N, R)                                                                                    \
    {                                                                                                          \
        LOC_ENSURE_LDS((l1_bwd_adam_rows_kernel<N, 13, R>), lds);                                              \
        hipLaunchKernelGGL((l1_bwd_adam_rows_kernel<N, 13, R>), dim3(R > 2 ? (n_active + 7) / 8 : grid),         \
                           dim3(R > 2 ? 512 : 256), lds, (hipStream_t)stream, X, x_pitch,                      \
                           rows, n_b, d->K, d->Kp, bn4, dz1, w1s, m1s, v1s, gb_scratch, b1, m_b1, v_b1,        \
                           alpha_tab, alpha_tab_len, lr, t_base, t_off, n_active);                             \
    }
#define LAUNCH_BWD_N(N)                                                                                        \
    switch (rb) {                                                                                              \
        case 2: LAUNCH_BWD_RB(N, 2) break;                                                                     \
        case 3: LAUNCH_BWD_RB(N, 3) break;                                                                     \
        default: LAUNCH_BWD_RB(N, 4) break;                                                                    \
    }
        switch (nht) {
            case 2: LAUNCH_BWD_N(2) break;
            case 4: LAUNCH_BWD_N(4) break;
            case 8: LAUNCH_BWD_N(8) break;
            default: loc_set_error("loc_l1_backward_adam: more than 32 rows need width 64/128/256 after padding (got %d)", d->Hp); return -1;
        }
#undef LAUNCH_BWD_N
#undef LAUNCH_BWD_RB
        LOC_CHECK_LAUNCH();
        return 0;
    }
#define LAUNCH_BWD(N)                                                                                          \
    {                                                                                                          \
        LOC_ENSURE_LDS((l1_bwd_adam_kernel<N>), lds);                                                          \
        hipLaunchKernelGGL(l1_bwd_adam_kernel<N>, dim3(grid), dim3(256), lds, (hipStream_t)stream, X, x_pitch, \
                           rows, n_b, d->K, d->Kp, bn4, dz1, w1s, m1s, v1s, gb_scratch, b1, m_b1, v_b1,        \
                           alpha_tab, alpha_tab_len, lr, t_base, t_off, n_active, (const uint8_t*)nullptr, 1.f); \
    }
#define LAUNCH_BWD_DROP(N)                                                                                     \
    {                                                                                                          \
        LOC_ENSURE_LDS((l1_bwd_adam_kernel<N, 13, true>), lds);                                                \
        hipLaunchKernelGGL((l1_bwd_adam_kernel<N, 13, true>), dim3(grid), dim3(256), lds, (hipStream_t)stream, X, \
                           x_pitch, rows, n_b, d->K, d->Kp, bn4, dz1, w1s, m1s, v1s, gb_scratch, b1, m_b1, v_b1, \
                           alpha_tab, alpha_tab_len, lr, t_base, t_off, n_active, in_mask, in_ks);             \
    }
    if (in_mask) {
        NHT_SWITCH(nht, LAUNCH_BWD_DROP)
        LOC_CHECK_LAUNCH();
        return 0;
    }
    // tune->l1b_nt_mask = 9 | 13 | 15 (or -1 for "nothing non-temporal") overrides the cache-policy mask NTM for
    // width 256 (measurement switch); 0 = the kernel's default
    const int ntm = !tune || tune->l1b_nt_mask == 0 ? -1 : (tune->l1b_nt_mask < 0 ? 0 : tune->l1b_nt_mask);
#define LAUNCH_BWD_NT(M)                                                                                       \
    {                                                                                                          \
        LOC_ENSURE_LDS((l1_bwd_adam_kernel<8, M>), lds);                                                       \
        hipLaunchKernelGGL((l1_bwd_adam_kernel<8, M>), dim3(grid), dim3(256), lds, (hipStream_t)stream, X, x_pitch, \
                           rows, n_b, d->K, d->Kp, bn4, dz1, w1s, m1s, v1s, gb_scratch, b1, m_b1, v_b1,        \
                           alpha_tab, alpha_tab_len, lr, t_base, t_off, n_active, (const uint8_t*)nullptr, 1.f); \
    }
    if (nht == 8 && ntm >= 0) {
        switch (ntm) {
            case 0: LAUNCH_BWD_NT(0) break; case 9: LAUNCH_BWD_NT(9) break; case 15: LAUNCH_BWD_NT(15) break;
            default: LAUNCH_BWD_NT(13) break;
        }
    } else {
        NHT_SWITCH(nht, LAUNCH_BWD)
    }
#undef LAUNCH_BWD
#undef LAUNCH_BWD_DROP
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_l1_backward_adam_main(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b,
                                         const loc_dims* d, const float* bn4, const float* dz1, float* w1s,
                                         float* m1s, float* v1s, float* b1, float* m_b1, float* v_b1,
                                         float* gb_scratch, const float* alpha_tab, int alpha_tab_len,
                                         const float* lr, const int* t_base, int t_off, int grid,
                                         const loc_tuning* tune, void* stream) {
    return l1_backward_main_impl(X, x_pitch, rows, n_b, d, bn4, dz1, w1s, m1s, v1s, b1, m_b1, v_b1, gb_scratch, alpha_tab,
                                 alpha_tab_len, lr, t_base, t_off, grid, tune, nullptr, 1.f, stream);
}

static int l1_backward_impl(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b,
                            const loc_dims* d, const float* bn4, const float* dz1, float* w1s, float* m1s,
                            float* v1s, float* gamma, float* beta, float* m_gamma, float* v_gamma,
                            float* m_beta, float* v_beta, float* b1, float* m_b1, float* v_b1,
                            float* gb_scratch, const float* alpha_tab, int alpha_tab_len, const float* lr,
                            const int* t_base, int t_off, int grid, const float* bn_next_stats,
                            float* bn4_out, void* ev_after_main, const loc_tuning* tune, const uint8_t* in_mask,
                            float in_ks, void* stream) {
    int rc = l1_backward_main_impl(X, x_pitch, rows, n_b, d, bn4, dz1, w1s, m1s, v1s, b1, m_b1, v_b1, gb_scratch,
                                   alpha_tab, alpha_tab_len, lr, t_base, t_off, grid, tune, in_mask, in_ks, stream);
    if (rc) return rc;
    if (ev_after_main) (void)hipEventRecord((hipEvent_t)ev_after_main, (hipStream_t)stream);
    hipLaunchKernelGGL(l1_gamma_beta_adam_kernel, dim3((d->K + 255) / 256), dim3(256), 0, (hipStream_t)stream, d->K,
                       gb_scratch, gamma, beta, m_gamma, v_gamma, m_beta, v_beta, alpha_tab, alpha_tab_len, lr,
                       t_base, t_off, d->Kp, bn_next_stats, bn4_out);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_l1_backward_adam_in_dropout(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b,
                                               const loc_dims* d, const float* bn4, const float* dz1, float* w1s,
                                               float* m1s, float* v1s, float* gamma, float* beta, float* m_gamma,
                                               float* v_gamma, float* m_beta, float* v_beta, float* b1, float* m_b1,
                                               float* v_b1, float* gb_scratch, const float* alpha_tab,
                                               int alpha_tab_len, const float* lr, const int* t_base, int t_off,
                                               int grid, const float* bn_next_stats, float* bn4_out,
                                               const loc_tuning* tune, const uint8_t* in_mask, float keep_scale,
                                               void* stream) {
    return l1_backward_impl(X, x_pitch, rows, n_b, d, bn4, dz1, w1s, m1s, v1s, gamma, beta, m_gamma, v_gamma, m_beta,
                            v_beta, b1, m_b1, v_b1, gb_scratch, alpha_tab, alpha_tab_len, lr, t_base, t_off, grid,
                            bn_next_stats, bn4_out, nullptr, tune, in_mask, keep_scale, stream);
}

extern "C" int loc_l1_backward_adam(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b,
                                    const loc_dims* d, const float* bn4, const float* dz1, float* w1s, float* m1s,
                                    float* v1s, float* gamma, float* beta, float* m_gamma, float* v_gamma,
                                    float* m_beta, float* v_beta, float* b1, float* m_b1, float* v_b1,
                                    float* gb_scratch, const float* alpha_tab, int alpha_tab_len, const float* lr,
                                    const int* t_base, int t_off, int grid, const float* bn_next_stats,
                                    float* bn4_out, void* ev_after_main, const loc_tuning* tune, void* stream) {
    return l1_backward_impl(X, x_pitch, rows, n_b, d, bn4, dz1, w1s, m1s, v1s, gamma, beta, m_gamma, v_gamma, m_beta,
                            v_beta, b1, m_b1, v_b1, gb_scratch, alpha_tab, alpha_tab_len, lr, t_base, t_off, grid,
                            bn_next_stats, bn4_out, ev_after_main, tune, nullptr, 1.f, stream);
}
