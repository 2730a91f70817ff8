// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA f32 32x32x2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/locator_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t fbits(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __builtin_bit_cast(float, u); }
// dword = { top16(lo) in bits 0..15, top16(hi) in bits 16..31 }: two bf16 values from two fp32 by truncation
__device__ __forceinline__ uint32_t pack_top16(uint32_t lo, uint32_t hi) {
    return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}

#define BN_EPS 1e-3f
#define BN_MOMENTUM 0.99f
#define ADAM_C1 0.1f     /* (float)(1 - 0.9)   */
#define ADAM_C2 0.001f   /* (float)(1 - 0.999) */
#define ADAM_EPS 1e-7f

// Row of accumulator register r in lane-half hi for v_mfma_f32_32x32x2_f32:
// D[i][j]: j = lane & 31, i = rowmap(r, lane >> 5).
__device__ __forceinline__ int rowmap(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float elu_f(float z) { return z > 0.f ? z : expm1f(z); }
// ELU'(z) from the activation value a = ELU(z):  1 if z > 0 else e^z = a + 1
__device__ __forceinline__ float elu_grad_from_act(float a) { return a > 0.f ? 1.f : a + 1.f; }

// Keras Adam: eps outside the root, bias correction folded into alpha.
__device__ __forceinline__ void adam_update(float& w, float& m, float& v, float g, float alpha) {
    m = m + (g - m) * ADAM_C1;
    v = v + (g * g - v) * ADAM_C2;
    w = w - (m * alpha) / (sqrtf(v) + ADAM_EPS);
}

__device__ __forceinline__ float adam_alpha(const float* alpha_tab, int alpha_tab_len, const float* lr,
                                            const int* t_base, int t_off) {
    int t = t_base[0] + t_off;
    if (t >= alpha_tab_len) t = alpha_tab_len - 1;
    return lr[0] * alpha_tab[t];
}

// gamma/beta Adam for SNP k from the per-wave partial sums left by l1_bwd_adam_kernel (fixed order: slot 0 +
// slot 1), and -- if next_stats is given -- the NEXT minibatch's [scale|shift|mean|rstd] from its precomputed
// batch statistics and the just-updated gamma/beta.  Shared by l1_gamma_beta_adam_kernel and the tail blocks of
// stack_dw_all_kernel.
__device__ __forceinline__ void gamma_beta_adam_body(int k, int Kp, const float* __restrict__ gbs,
                                                     float* __restrict__ gamma, float* __restrict__ beta,
                                                     float* __restrict__ m_gamma, float* __restrict__ v_gamma,
                                                     float* __restrict__ m_beta, float* __restrict__ v_beta,
                                                     float alpha, const float* __restrict__ next_stats,
                                                     float* __restrict__ bn4) {
    const float* g0 = gbs + (int64_t)(k >> 5) * 128 + (k & 31);
    const float dg = g0[0] + g0[64], db = g0[32] + g0[96];
    float wv = gamma[k], mv = m_gamma[k], vv = v_gamma[k];
    adam_update(wv, mv, vv, dg, alpha);
    const float g = wv;
    gamma[k] = wv; m_gamma[k] = mv; v_gamma[k] = vv;
    wv = beta[k]; mv = m_beta[k]; vv = v_beta[k];
    adam_update(wv, mv, vv, db, alpha);
    beta[k] = wv; m_beta[k] = mv; v_beta[k] = vv;
    if (next_stats) {
        const float mu = next_stats[k], rstd = 1.0f / sqrtf(next_stats[Kp + k] + BN_EPS);
        const float sc = g * rstd;
        bn4[k] = sc;
        bn4[Kp + k] = wv - mu * sc;
        bn4[2 * (int64_t)Kp + k] = mu;
        bn4[3 * (int64_t)Kp + k] = rstd;
    }
}

// ---- Philox4x32-10 counter RNG -------------------------------------------------
struct philox4 {
    uint32_t v[4];
};
__host__ __device__ inline philox4 philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    philox4 r;
    r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
    return r;
}

// W1S index of element (h, k): see include/locator_hip.h
__host__ __device__ inline int64_t w1s_index(int h, int k, int nht) {
    int kt = k >> 5, kl = k & 31, ht = h >> 5, hl = h & 31;
    int q = hl >> 3, hi = (hl >> 2) & 1, c = hl & 3;
    return ((int64_t)(kt * nht + ht) * 4 + q) * 256 + (hi * 32 + kl) * 4 + c;
}

// host-side error plumbing (api.hip)
void loc_set_error(const char* fmt, ...);
// l1_gemm.hip: the shift-term and SNP-group reductions shared by the bf16 and int8 large-M GEMMs
int gm_launch_cvec(const float* cpart, int nkt64, float* cvec8, void* stream);
int gm_launch_reduce(const float* partial, int G, int64_t MH, const float* cvec8, const float* b1, float* a1,
                     void* stream);

// stack_rows.hip: the hidden stack + heads for many rows on the fp32 matrix pipe (32 or 16 rows per workgroup: tile_rows, 0 = by row count)
int sr_eval_launch(const float* a1, const float* rd_partial, int rd_G, int64_t rd_MH, const float* rd_cvec8, const float* rd_b1,
                   const float* Wh, const float* bh, const float* wa, const float* ba, const float* wb, const float* bb, int L,
                   int n_b, const int32_t* rows, const float* Y, float* yhat, float* dist, int tile_rows, void* stream);
extern "C" int loc_stack_rows_min_rows(void);
int sr_compute_units();   // compute units of the current device (cached per device; 256 when there is none)
extern "C" int loc_stack_rows_supported(int Hp, int L);

// l1_kernels.hip: the layer-1 reduction alone (partial sums of G groups -> a1, optional Dropout on a1)
int loc_l1_reduce_launch_drop(const float* partial, int G, int rows_p, int Hp, const float* b1, float* a1, float* a1_drop,
                              const uint8_t* mask, float keep_scale, void* stream);

// Raises a kernel's dynamic-LDS limit.  The limit is an attribute of the function PER DEVICE, so the "largest value
// set so far" is remembered per (call site = kernel instantiation, device); the call is idempotent, which makes the
// unsynchronised cache benign.  FUNC may hold template commas: wrap it in parentheses.
#define LOC_MAX_DEVICES 64
#define LOC_ENSURE_LDS(FUNC, BYTES)                                                                          \
    do {                                                                                                     \
        static size_t set__[LOC_MAX_DEVICES] = {};                                                           \
        int dev__ = 0;                                                                                       \
        (void)hipGetDevice(&dev__);                                                                          \
        dev__ = dev__ < 0 ? 0 : dev__ % LOC_MAX_DEVICES;                                                     \
        if ((size_t)(BYTES) > set__[dev__]) {                                                                \
            hipError_t e__ = hipFuncSetAttribute(reinterpret_cast<const void*>(FUNC),                        \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES));  \
            if (e__ != hipSuccess) {                                                                         \
                loc_set_error("hipFuncSetAttribute(%zu): %s", (size_t)(BYTES), hipGetErrorString(e__));     \
                return (int)e__;                                                                             \
            }                                                                                                \
            set__[dev__] = (size_t)(BYTES);                                                                  \
        }                                                                                                    \
    } while (0)
#define LOC_GRID_Y_MAX 32768 /* HIP limits grid.y to 65535: kernels launched with one y-block per row stride over it */
#define LOC_CHECK_LAUNCH()                                              \
    do {                                                                \
        hipError_t e__ = hipGetLastError();                             \
        if (e__ != hipSuccess) {                                        \
            loc_set_error("%s: %s", __func__, hipGetErrorString(e__));  \
            return (int)e__;                                            \
        }                                                               \
    } while (0)
