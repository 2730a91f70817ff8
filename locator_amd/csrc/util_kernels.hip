// Utility kernels: Glorot init, dropout keep-masks, bootstrap column gather, W1 layout conversion.
#include "common.h"

__device__ __forceinline__ float u01(uint32_t r) { return ((float)r + 0.5f) * 2.3283064365386963e-10f; }

// Logical matrix R x C (Keras kernel, in x out), value(r, c) keyed by the counter r*C + c so the
// same seed gives the same logical matrix whatever the storage layout.  Storage: swizzled W1S
// (r = SNP k, c = unit h) or row-major [Rp][Cp]; padding is written as zero.
__global__ void init_glorot_kernel(float* __restrict__ dst, int R, int C, int Rp, int Cp, int swizzled, float limit,
                                   uint64_t seed, uint64_t stream_id) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Rp * Cp) return;
    int r = (int)(i / Cp), c = (int)(i % Cp);
    float val = 0.f;
    if (r < R && c < C) {
        philox4 p = philox4x32_10((uint64_t)r * (uint64_t)C + (uint64_t)c, stream_id, seed);
        val = (2.f * u01(p.v[0]) - 1.f) * limit;
    }
    int64_t o = swizzled ? w1s_index(c, r, Cp / 32) : i;
    dst[o] = val;
}

__global__ void init_uniform_kernel(float* __restrict__ dst, int64_t n, float limit, uint64_t seed,
                                    uint64_t stream_id) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * j >= n) return;
    philox4 p = philox4x32_10((uint64_t)j, stream_id, seed);
    for (int c = 0; c < 4; ++c)
        if (4 * j + c < n) dst[4 * j + c] = (2.f * u01(p.v[c]) - 1.f) * limit;
}

__global__ void dropout_mask_kernel(uint8_t* __restrict__ mask, int64_t n, uint32_t thresh, uint64_t seed,
                                    uint64_t offset4) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * j >= n) return;
    philox4 p = philox4x32_10(offset4 + (uint64_t)j, 0x6d61736bULL /* "mask" */, seed);
    for (int c = 0; c < 4; ++c)
        if (4 * j + c < n) mask[4 * j + c] = p.v[c] >= thresh ? 1 : 0;
}

__global__ void gather_columns_kernel(const uint8_t* __restrict__ src, int64_t src_pitch,
                                      const int32_t* __restrict__ site_order, int K, uint8_t* __restrict__ dst,
                                      int64_t dst_pitch, int n_rows) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= K) return;
    const int so = site_order[j];
    for (int r = blockIdx.y; r < n_rows; r += gridDim.y)
        dst[(int64_t)r * dst_pitch + j] = src[(int64_t)r * src_pitch + so];
}

__global__ void w1_swizzle_kernel(const float* __restrict__ w_kh, int K, int H, float* __restrict__ w1s, int Kp,
                                  int Hp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Kp * Hp) return;
    int k = (int)(i / Hp), h = (int)(i % Hp);
    w1s[w1s_index(h, k, Hp / 32)] = (k < K && h < H) ? w_kh[(int64_t)k * H + h] : 0.f;
}

__global__ void w1_unswizzle_kernel(const float* __restrict__ w1s, int Hp, float* __restrict__ w_kh, int K, int H) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)K * H) return;
    int k = (int)(i / H), h = (int)(i % H);
    w_kh[i] = w1s[w1s_index(h, k, Hp / 32)];
}

extern "C" int loc_init_glorot(float* dst, int R, int C, int Rp, int Cp, int swizzled, uint64_t seed,
                               uint64_t stream_id, void* stream) {
    if (swizzled && (Rp % 32 || Cp % 32)) { loc_set_error("loc_init_glorot: swizzled needs Rp,Cp %% 32 == 0"); return -1; }
    double limit = sqrt(6.0 / ((double)R + (double)C));
    int64_t n = (int64_t)Rp * Cp;
    hipLaunchKernelGGL(init_glorot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst,
                       R, C, Rp, Cp, swizzled, (float)limit, seed, stream_id);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_init_uniform(float* dst, int64_t n, float limit, uint64_t seed, uint64_t stream_id,
                                void* stream) {
    if (n <= 0) return 0;
    int64_t nt = (n + 3) / 4;
    hipLaunchKernelGGL(init_uniform_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dst, n, limit, seed, stream_id);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_dropout_mask_fill(uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset,
                                     void* stream) {
    if (n <= 0) return 0;
    if (offset % 4) { loc_set_error("loc_dropout_mask_fill: offset must be a multiple of 4"); return -1; }
    if (!(p >= 0.f && p < 1.f)) { loc_set_error("loc_dropout_mask_fill: p=%f out of [0,1)", p); return -1; }
    uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    int64_t nt = (n + 3) / 4;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       mask, n, thresh, seed, offset / 4);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_gather_columns(const uint8_t* src, int64_t src_pitch, const int32_t* site_order, int K,
                                  uint8_t* dst, int64_t dst_pitch, int n_rows, void* stream) {
    if (n_rows <= 0 || K <= 0) return 0;
    hipLaunchKernelGGL(gather_columns_kernel, dim3((K + 255) / 256, n_rows < LOC_GRID_Y_MAX ? n_rows : LOC_GRID_Y_MAX), dim3(256),
                       0, (hipStream_t)stream, src, src_pitch, site_order, K, dst, dst_pitch, n_rows);
    LOC_CHECK_LAUNCH();
    return 0;
}

// largest genotype byte: 16 bytes per thread and step, wave maximum by shuffles, one atomicMax per wave
__global__ __launch_bounds__(256) void genotype_max_kernel(const uint8_t* __restrict__ X, int64_t pitch, int K,
                                                           int n_rows, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    const int K16 = K & ~15;
    for (int r = blockIdx.y; r < n_rows; r += gridDim.y) {        // grid.y is capped (LOC_GRID_Y_MAX): rows stride over it
        const uint8_t* row = X + (int64_t)r * pitch;
        for (int k = (blockIdx.x * 256 + threadIdx.x) * 16; k < K16; k += gridDim.x * 256 * 16) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(row + k);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t a = v[d];
                m = max(m, max(max(a & 255u, (a >> 8) & 255u), max((a >> 16) & 255u, a >> 24)));
            }
        }
        if (blockIdx.x == 0)
            for (int k = K16 + threadIdx.x; k < K; k += 256) m = max(m, (uint32_t)row[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    // one atomic per wave at most - and none once the answer is in: a diploid matrix reaches its maximum of 2 within the first
    // few waves, and 131,072 waves of a 16,384 x 100,000 matrix queueing on one address took 6 ms for a 0.3 ms pass
    if ((threadIdx.x & 63) == 0 && m > *reinterpret_cast<volatile uint32_t*>(out)) atomicMax(out, m);
}

extern "C" int loc_genotype_max(const uint8_t* X, int64_t x_pitch, int n_rows, int K, uint32_t* out, void* stream) {
    if (n_rows <= 0 || K <= 0) return 0;
    if (((uintptr_t)X & 15) || x_pitch % 16) { loc_set_error("loc_genotype_max: needs a 16-byte aligned X and row pitch"); return -1; }
    const int gx = K >= 65536 ? 8 : 1;
    hipLaunchKernelGGL(genotype_max_kernel, dim3(gx, n_rows < LOC_GRID_Y_MAX ? n_rows : LOC_GRID_Y_MAX), dim3(256), 0,
                       (hipStream_t)stream, X, x_pitch, K, n_rows, out);
    LOC_CHECK_LAUNCH();
    return 0;
}

// 2-bit packing of a genotype matrix whose values are 0..3 (loc_genotype_max): four SNPs per byte, SNP 4 j + i in bits
// 2 i + 1 : 2 i of byte j.  One thread packs 16 SNPs into 4 bytes.
__global__ __launch_bounds__(256) void pack_genotypes_2bit_kernel(const uint8_t* __restrict__ X, int64_t pitch, int Kp,
                                                                  uint8_t* __restrict__ X2, int64_t pitch2, int n_rows) {
    const int k16 = blockIdx.x * 256 + threadIdx.x;
    if (k16 * 16 >= Kp) return;
    for (int r = blockIdx.y; r < n_rows; r += gridDim.y) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(X + (int64_t)r * pitch + 16 * k16);
        uint32_t out = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t a = v[q];
            const uint32_t b = (a & 3u) | (((a >> 8) & 3u) << 2) | (((a >> 16) & 3u) << 4) | (((a >> 24) & 3u) << 6);
            out |= b << (8 * q);
        }
        *reinterpret_cast<uint32_t*>(X2 + (int64_t)r * pitch2 + 4 * k16) = out;
    }
}

extern "C" int loc_pack_genotypes_2bit(const uint8_t* X, int64_t x_pitch, int n_rows, int Kp, uint8_t* X2, int64_t x2_pitch,
                                       void* stream) {
    if (n_rows <= 0) return 0;
    if (((uintptr_t)X & 15) || x_pitch % 16 || Kp % 16 || ((uintptr_t)X2 & 3) || x2_pitch % 4 || x2_pitch < Kp / 4) {
        loc_set_error("loc_pack_genotypes_2bit: needs a 16-byte aligned X / row pitch, Kp %% 16 == 0 and a 4-byte aligned X2 with "
                      "row pitch >= Kp / 4");
        return -1;
    }
    hipLaunchKernelGGL(pack_genotypes_2bit_kernel, dim3((Kp / 16 + 255) / 256, n_rows < LOC_GRID_Y_MAX ? n_rows : LOC_GRID_Y_MAX),
                       dim3(256), 0, (hipStream_t)stream, X, x_pitch, Kp, X2, x2_pitch, n_rows);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_w1_swizzle(const float* w_kh, int K, int H, float* w1s, int Kp, int Hp, void* stream) {
    int64_t n = (int64_t)Kp * Hp;
    hipLaunchKernelGGL(w1_swizzle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_kh,
                       K, H, w1s, Kp, Hp);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_w1_unswizzle(const float* w1s, int Kp, int Hp, float* w_kh, int K, int H, void* stream) {
    int64_t n = (int64_t)K * H;
    (void)Kp;
    hipLaunchKernelGGL(w1_unswizzle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w1s,
                       Hp, w_kh, K, H);
    LOC_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// The three Keras callbacks of a fit on the device (locator.py:330-362), so that model.fit's epoch loop
// (locator.py:367-376) needs no host round trip between epochs: nothing that decides the NEXT epoch's work depends on
// the host - permutations and dropout masks are callback-independent, and ModelCheckpoint / EarlyStopping /
// ReduceLROnPlateau are three comparisons of one scalar.
// ---------------------------------------------------------------------------------------------------------
// One workgroup.  loss = sum_j loss_j n_j / n_train (what Keras logs: the sample-weighted mean of the per-step losses),
// val_loss = mean of the validation distances; both summed in double, in index order.  Then, in the callback list's order
// (locator.py:362): checkpoint (strict <), early stopping (wait += 1 first; fires on wait >= patience and epoch > 0),
// LR plateau (strict <, else wait += 1 and on wait >= patience: lr <- max(lr * factor, 0) in fp32, wait <- 0).  The LR
// logged for an epoch is the one it trained with.  Once early stopping has fired the state is frozen: later epochs that
// the host had already enqueued change neither the best weights nor the LR nor the history rows the host keeps.
__global__ __launch_bounds__(256) void epoch_callbacks_kernel(const float* __restrict__ stats, int steps, int batch, int n_last,
                                                              int n_val, loc_cb_state* __restrict__ st, float* __restrict__ lr,
                                                              double* __restrict__ hist, int hist_cap) {
    __shared__ float buf[1024];
    const int t = threadIdx.x;
    const int n_train = (steps - 1) * batch + n_last;
    double loss = 0.0, val = 0.0;
    // stage through LDS in chunks of 1024 values; thread 0 adds them up in index order (a few hundred values per epoch)
    for (int base = 0; base < steps; base += 1024) {
        const int m = steps - base < 1024 ? steps - base : 1024;
        __syncthreads();
        for (int i = t; i < m; i += 256) buf[i] = stats[base + i];
        __syncthreads();
        if (t == 0)
            for (int i = 0; i < m; ++i) loss += (double)buf[i] * (double)(base + i == steps - 1 ? n_last : batch);
    }
    for (int base = 0; base < n_val; base += 1024) {
        const int m = n_val - base < 1024 ? n_val - base : 1024;
        __syncthreads();
        for (int i = t; i < m; i += 256) buf[i] = stats[steps + base + i];
        __syncthreads();
        if (t == 0)
            for (int i = 0; i < m; ++i) val += (double)buf[i];
    }
    if (t != 0) return;
    loss /= (double)n_train;
    val /= (double)n_val;
    const int epoch = st->epoch;
    st->epoch = epoch + 1;
    st->save_now = 0;
    if (st->stopped) return;
    const float lr_logged = st->lr;
    int flags = 0;
    if (val < st->ck_best) { st->ck_best = val; st->save_now = 1; st->best_epoch = epoch; flags |= 1; }
    st->es_wait += 1;
    if (val < st->es_best) { st->es_best = val; st->es_wait = 0; }
    if (st->es_wait >= st->patience && epoch > 0) { st->stopped = 1; st->stop_epoch = epoch; flags |= 2; }
    if (val < st->rl_best) {
        st->rl_best = val;
        st->rl_wait = 0;
    } else {
        st->rl_wait += 1;
        if (st->rl_wait >= st->lr_patience) {
            const float nl = fmaxf(st->lr * st->lr_factor, 0.0f);
            st->lr = nl;
            *lr = nl;
            st->rl_wait = 0;
            flags |= 4;
        }
    }
    if (epoch < hist_cap) {
        double* row = hist + 4 * (int64_t)epoch;
        row[0] = loss; row[1] = val; row[2] = (double)lr_logged; row[3] = (double)flags;
    }
}

// ModelCheckpoint's save as a predicated device copy: every workgroup reads the flag the callback kernel left.
__global__ __launch_bounds__(256) void snapshot_if_kernel(const loc_cb_state* __restrict__ st, const f32x4* __restrict__ src,
                                                          f32x4* __restrict__ dst, int64_t n4) {
    if (!st->save_now) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

extern "C" int loc_epoch_callbacks(const float* stats, int steps, int batch, int n_last, int n_val, loc_cb_state* state,
                                   float* lr, double* hist, int hist_cap, void* stream) {
    if (steps < 1 || batch < 1 || n_last < 1 || n_last > batch || n_val < 1 || !state || !lr || !hist) {
        loc_set_error("loc_epoch_callbacks: steps=%d batch=%d n_last=%d n_val=%d (all >= 1, n_last <= batch)", steps, batch,
                      n_last, n_val);
        return -1;
    }
    hipLaunchKernelGGL(epoch_callbacks_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, steps, batch, n_last, n_val,
                       state, lr, hist, hist_cap);
    LOC_CHECK_LAUNCH();
    return 0;
}

extern "C" int loc_snapshot_if(const loc_cb_state* state, const float* params, float* best, int64_t n, void* stream) {
    if (n % 4 || ((uintptr_t)params & 15) || ((uintptr_t)best & 15)) {
        loc_set_error("loc_snapshot_if: needs 16-byte aligned buffers and a multiple of 4 floats");
        return -1;
    }
    hipLaunchKernelGGL(snapshot_if_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, state,
                       reinterpret_cast<const f32x4*>(params), reinterpret_cast<f32x4*>(best), n / 4);
    LOC_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Replicate summarisation (the step right after --windows / --bootstrap): per sample, over its n replicate predictions,
//   the geographic centroid = mean of the predictions                       (/root/reference/locator_py/plot_locator.py:39-44)
//   the prediction with the highest Gaussian kernel density, bandwidth h, the density evaluated at the predictions
//   themselves: sklearn KernelDensity(kernel='gaussian').fit(P).score_samples(P) and the FIRST index of its maximum
//                                                                            (plot_locator.py:26-37, scripts/plot_locator.R:57-113)
// score_i = log sum_j exp(-|p_i - p_j|^2 / (2 h^2)) + const: the j = i term is exp(0), so the log-sum-exp needs no shift.
// One workgroup per sample, float64 throughout (map units): the points sit in LDS (up to KDE_LDS_POINTS; beyond that they are
// read through L1), thread t scores points t, t + 256, ... by a j-ordered sum - a fixed order, so the result does not depend on
// the launch - and the workgroup reduces (score, index) with "larger score, then smaller index".  A sample with a non-finite
// coordinate has no density estimate (sklearn raises; the reference then reports the mean): peak index -1, peak = centroid.
constexpr int KDE_LDS_POINTS = 4096;          // 64 KB of (x, y) doubles

__global__ __launch_bounds__(256) void kde_peak_kernel(const double* __restrict__ xy, const int64_t* __restrict__ offsets,
                                                       double inv_2h2, int32_t* __restrict__ peak_index,
                                                       double* __restrict__ out) {
    __shared__ double pts[2 * KDE_LDS_POINTS];
    __shared__ double r_score[256];
    __shared__ int r_idx[256];
    __shared__ double r_sx[256], r_sy[256];
    __shared__ int r_bad[256];
    const int s = blockIdx.x, t = threadIdx.x;
    const int64_t o0 = offsets[s];
    const int n = (int)(offsets[s + 1] - o0);
    const double* p = xy + 2 * o0;
    const bool in_lds = n <= KDE_LDS_POINTS;
    double sx = 0.0, sy = 0.0;
    int bad = 0;
    for (int i = t; i < n; i += 256) {
        const double x = p[2 * i], y = p[2 * i + 1];
        if (in_lds) { pts[2 * i] = x; pts[2 * i + 1] = y; }
        sx += x; sy += y;
        bad |= !(isfinite(x) && isfinite(y));
    }
    r_sx[t] = sx; r_sy[t] = sy; r_bad[t] = bad;
    __syncthreads();
    const double* q = in_lds ? pts : p;
    double best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int i = t; i < n; i += 256) {
        const double xi = q[2 * i], yi = q[2 * i + 1];
        double acc = 0.0;
        for (int j = 0; j < n; ++j) {
            const double dx = xi - q[2 * j], dy = yi - q[2 * j + 1];
            acc += exp(-(dx * dx + dy * dy) * inv_2h2);
        }
        const double sc = log(acc);
        if (sc > best) { best = sc; best_i = i; }          // ascending i: a tie keeps the earlier point
    }
    r_score[t] = best; r_idx[t] = best_i;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) {
            r_sx[t] += r_sx[t + w]; r_sy[t] += r_sy[t + w]; r_bad[t] |= r_bad[t + w];
            const double a = r_score[t], b = r_score[t + w];
            if (b > a || (b == a && r_idx[t + w] < r_idx[t])) { r_score[t] = b; r_idx[t] = r_idx[t + w]; }
        }
        __syncthreads();
    }
    if (t == 0) {
        const double gx = n ? r_sx[0] / (double)n : NAN, gy = n ? r_sy[0] / (double)n : NAN;
        const bool ok = n > 0 && !r_bad[0] && r_idx[0] != 0x7fffffff;
        const int pi = ok ? r_idx[0] : -1;
        peak_index[s] = pi;
        out[4 * s + 0] = ok ? p[2 * pi] : gx;
        out[4 * s + 1] = ok ? p[2 * pi + 1] : gy;
        out[4 * s + 2] = gx;
        out[4 * s + 3] = gy;
    }
}

extern "C" int loc_kde_peak_batch(const double* xy, const int64_t* offsets, int n_samples, double bandwidth,
                                  int32_t* peak_index, double* out, void* stream) {
    if (n_samples < 0 || !(bandwidth > 0.0) || (n_samples > 0 && (!xy || !offsets || !peak_index || !out))) {
        loc_set_error("loc_kde_peak_batch: n_samples=%d bandwidth=%g (needs n_samples >= 0, bandwidth > 0, non-null buffers)",
                      n_samples, bandwidth);
        return -1;
    }
    if (n_samples == 0) return 0;
    hipLaunchKernelGGL(kde_peak_kernel, dim3(n_samples), dim3(256), 0, (hipStream_t)stream, xy, offsets,
                       1.0 / (2.0 * bandwidth * bandwidth), peak_index, out);
    LOC_CHECK_LAUNCH();
    return 0;
}
