/*
 * locator_hip.h — C ABI of liblocator_hip.so (gfx950 / MI355X).
 *
 * The reference (kr-colab/locator) has no FFI: its hot path is four Python
 * functions that call Keras (load_network / load_callbacks / train_network /
 * predict_locs, /root/reference/locator/locator.py:311-470).  This header is
 * the boundary a maintainer binds instead of `tf.keras`: every entry point
 * names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ or torch types.
 *   - every pointer is a DEVICE pointer unless the name starts with `h_`.
 *   - the caller owns every buffer; nothing here allocates or frees device memory.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no
 *     implicit synchronisation.  The library reads no environment variables and
 *     keeps no settings: every speed hint is an explicit argument (loc_tuning).
 *     The only process state is loc_last_error's thread-local message and, per
 *     kernel and device, the largest dynamic-LDS limit already requested with
 *     hipFuncSetAttribute (an idempotent cache).  Safe from several host threads
 *     / processes on different streams or devices; safe to capture into a hipGraph.
 *   - return value: 0 = ok, otherwise a hipError_t (or -1 for a bad argument);
 *     loc_last_error() returns a thread-local message.
 *
 * Data layout
 *   X      genotypes, uint8, sample-major [n_samples][x_pitch], x_pitch = Kp.
 *   W1S    layer-1 kernel, fp32, tile-swizzled: element (h,k) lives at
 *            ((kt*nht + ht)*4 + q)*256 + (hi*32 + kl)*4 + c
 *          kt=k>>5 kl=k&31 ht=h>>5 q=(h&31)>>3 hi=((h&31)>>2)&1 c=h&3, nht=Hp/32;
 *          i.e. each (32 SNP x 32 unit) tile is stored in the accumulator
 *          layout of v_mfma_f32_32x32x2_f32, so a wave streams it with
 *          fully-coalesced 16-byte loads.  Adam moments use the same layout.
 *   hidden kernels   fp32 row-major [Hp][Hp] (in x out, the Keras orientation).
 *   Kp = K rounded up to 32, Hp = width rounded up to 32; padding is zero and
 *   stays zero under training.
 */
#ifndef LOCATOR_HIP_H
#define LOCATOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOC_ROWS 32 /* rows per row block = MFMA M                                              */
#define LOC_MAX_WIDTH 1024 /* --width limit.  Up to 512 (after padding to 64/128/256/512) the hidden stack is one fused
                              launch; other widths, and everything above 512, take the per-layer kernels           */
#define LOC_MAX_BATCH 128 /* largest --batch_size of the row-block kernels: four 32-row blocks per step */
#define LOC_BIG_BATCH_MAX 4096 /* --batch_size limit: above LOC_MAX_BATCH the step streams its row blocks from L2
                                  (l1_bwd_adam_big_kernel, run-time block counts in the tail) - correct, not tuned */
#define LOC_BATCH_SLOT 128 /* rows per activation slot of the training scratch when batch > 32   */
#define LOC_ROWS_TILE 128      /* rows per workgroup tile of the large-M layer-1 forward          */
#define LOC_PREDICT_CHUNK 16384 /* rows per large-M launch inside loc_predict (more rows per launch = fewer SNP groups
                                   = less partial-sum traffic, prologue and epilogue amortised: the int8 GEMM on the
                                   batched --jacknife shape goes 0.50 -> 0.55 of the bf16 peak from 4096 to 16384 rows) */
#define LOC_ROWS_BLOCKS 256    /* 128-row tiles the large-M scratch is sized for (one workgroup per CU)   */

typedef struct loc_dims {
    int K;     /* SNPs after filtering                        (traingen.shape[1], locator.py:318) */
    int Kp;    /* K rounded up to a multiple of 32            */
    int H;     /* --width                                     (locator.py:320)                    */
    int Hp;    /* H rounded up to a multiple of 32 (<= LOC_MAX_WIDTH) */
    int L;     /* --nlayers: number of Dense+ELU layers, >= 1 (locator.py:319-323)                */
    int n_pre; /* floor(L/2): ELU layers before Dropout       (locator.py:319); 0 for --nlayers 1:
                  Dropout then acts on the BatchNorm output and its keep mask is [rows][Kp]        */
} loc_dims;

/* Offsets (in floats) of every tensor inside the flat parameter buffer.
 * [0, n_trainable) is trainable (Adam m/v buffers have that length);
 * mov_mean / mov_var follow, so one copy of n_total floats is the
 * ModelCheckpoint snapshot (locator.py:332-348: all weights incl. BN moving stats). */
typedef struct loc_layout {
    int64_t w1;       /* Hp*Kp, swizzled                 */
    int64_t gamma;    /* Kp                              */
    int64_t beta;     /* Kp                              */
    int64_t b1;       /* Hp                              */
    int64_t wh;       /* (L-1) * Hp*Hp                   */
    int64_t bh;       /* (L-1) * Hp                      */
    int64_t wa;       /* Hp*2   Dense(2) #1 (locator.py:324) */
    int64_t ba;       /* 2                               */
    int64_t wb;       /* 4      Dense(2) #2 (locator.py:325) */
    int64_t bb;       /* 2                               */
    int64_t n_trainable;
    int64_t mov_mean; /* Kp */
    int64_t mov_var;  /* Kp */
    int64_t n_total;
} loc_layout;

/* Speed hints and measurement switches.  All-zero = defaults.  None of them changes a result beyond the summation
 * order of a documented exact mode (tests/test_gpu_edge.py: bit-identical fits for every placement value). */
typedef struct loc_tuning {
    int stack_helpers;    /* L2 warm-up helper workgroups of the fused hidden stack: 0 = default (12), -1 = none   */
    int stack_xcd_stride; /* 1, 2, 4 or 8: every n-th workgroup of the fused stack works, so the workers share
                             8/n XCDs (observed block -> XCD = b % 8); 0 = default (8: one XCD)                     */
    int l1b_nt_mask;      /* which Adam streams of the layer-1 backward are non-temporal (width 256): 9, 13, 15,
                             -1 = none; 0 = default (13: moments and the W1 write-back)                             */
    int l1b_rows;         /* 1: route <= 32-row steps of width 256 through the bf16x3 row-block backward           */
    int rows_rt;          /* 8: 256-row tiles in loc_l1_forward_rows (width 256, even number of 128-row tiles)     */
    int gemm_i8_unit_tiles; /* int8 GEMM: 32-unit tiles per wave: 1 = eight waves per workgroup (two per SIMD, 12 digit
                             fragments in flight each), 2 = four waves (one per SIMD, 512 registers, 32 in flight);
                             0 = default                                                                            */
    int stack_rows;       /* hidden stack of a many-row predict: 0 = default: from loc_stack_rows_min_rows() (8 x compute units + 1)
                             rows per chunk the fp32 matrix pipe - 16 or 32 rows per workgroup, whichever the measured time
                             model puts ahead on this device (16-row tiles up to 4096 rows and at 8193..12288 on 256 units) -
                             and 2, 4 or 8 rows per workgroup on the vector ALU below (the fewest that keep the rows in one
                             round of workgroups); 1 = the 32-row matrix-pipe form for every chunk; 2 = the 16-row form for
                             every chunk; -1 / -2 / -3 = always 2 / 4 / 8 rows per workgroup on the vector ALU
                             (measurement switches)                                                                       */
    int gemm_reduce;      /* many-row predicts on the int8 pipe: 0 = default: the SNP-group sum + shift + b1 + ELU of the layer-1
                             GEMM is its own launch (l1_gemm_reduce_kernel); 1 = it happens in the input stage of the
                             hidden-stack launch instead (loc_l1_forward_gemm_i8_partial + loc_stack_forward_eval_partial: same
                             bits, no reduction launch, no a1 round trip - built in round 4 and measured SLOWER: the few
                             workgroups of the stack launch read the 33 MB of partial sums with less parallelism than the
                             dedicated kernel, +17 us at 1000 rows and +30 us at 4096 against 8 us; kept as a switch)        */
    int chain_tail;       /* chained steps (loc_train_step_chain): 0 = default: the step's hidden-layer / head Adam tail runs
                             as trailing workgroups of the chained layer-1 launch (they fill the compute units that finish
                             their k-tiles an iteration early); -1 = its own launch after it (measurement switch)          */
    int stack_train_rows; /* batch rows per workgroup of the fused hidden stack of a TRAINING step: 0 = default: 1 (32 workers +
                             helpers over two XCDs for a 32-row step: measured 162 against 171 us per step at the metric's
                             shape - a worker is bound by its 4.7 MB weight stream, halving its rows' arithmetic shortens
                             what does not overlap with it); 1, 2 (rounds 1-4) or 4 = that many (measurement switch)         */
} loc_tuning;

/* Everything a training / inference step needs.  All device pointers. */
typedef struct loc_net {
    loc_dims d;
    float* params;           /* n_total floats                                        */
    float* adam_m;           /* n_trainable floats                                    */
    float* adam_v;           /* n_trainable floats                                    */
    const float* alpha_tab;  /* alpha_tab[t] = sqrt(1-b2^t)/(1-b1^t), t = 0..len-1     */
    int alpha_tab_len;
    const float* lr;         /* 1 float: current learning rate (ReduceLROnPlateau)    */
    const int* t_base;       /* 1 int: Adam steps completed before this graph/epoch   */
    const uint8_t* X;        /* [n_samples][x_pitch]                                  */
    int64_t x_pitch;
    const float* Y;          /* [n_samples][2] z-scored targets (locator.py:284-292)  */
    float drop_p;            /* --dropout_prop                                        */
    float* wht;              /* (L-1)*Hp*Hp: transposed hidden kernels (derived state, kept in sync by
                                loc_stack_dw_adam; refresh with loc_transpose_hidden after loading weights) */
    /* workspace, sized by loc_workspace_floats() */
    float* ws;
    float* ws_predict;       /* optional second workspace of the same size for loc_predict / loc_predict_scan (NULL = ws).  A fit
                                that chains the last step of an epoch into the first layer-1 forward of the next leaves that
                                forward's partial sums and scale / shift in ws across the epoch's validation sweep, which
                                therefore must work elsewhere */
    int l1_fwd_grid;         /* workgroups of the layer-1 forward (<= LOC_MAX_FWD_GRID) */
    int l1_bwd_grid;         /* workgroups of the layer-1 backward                      */
    int slot_rows;           /* rows per activation slot of the training scratch: 0 or 32 (--batch_size <= 32),
                                64 or LOC_BATCH_SLOT when --batch_size is 33..64 / 65..LOC_MAX_BATCH (the same 128-row scratch
                                layout; 64 lets steps of two 32-row blocks be chained), or --batch_size rounded up to a
                                multiple of 128 above that (the workspace then comes from loc_workspace_floats_batch) */
    int predict_pieces;      /* bf16 pieces per weight in the large-M inference forward: 3 = exact fp32
                                products (default when 0), 2 = ~2^-17, 1 = plain bf16 weights; -1 forces the
                                32-row fp32-MFMA kernel for every block of rows                          */
    void* l1_image;          /* optional: loc_l1_image_bytes(d, pieces) / loc_l1_image_i8_bytes(d, digits) bytes of
                                scratch.  When set, loc_predict over at least LOC_GEMM_MIN_ROWS(pieces) /
                                LOC_GEMM_I8_MIN_ROWS(digits) rows converts W1 once per call (loc_l1_image_build /
                                loc_l1_image_i8_build) and runs every row chunk through loc_l1_forward_gemm /
                                loc_l1_forward_gemm_i8; NULL keeps loc_l1_forward_rows */
    int64_t l1_image_bytes;
    int x_max;               /* largest genotype value in X (loc_genotype_max), 0 = not known.  The int8 GEMM needs
                                1 <= x_max <= 127; otherwise many-row predicts take the bf16 pieces above          */
    int predict_digits;      /* int8 digit planes per weight in the many-row inference forward: 3 = 24-bit fixed point
                                against each unit's largest weight (exact to fp32 accumulation; default when 0),
                                2 = 16-bit fixed point (fast), -1 = never (bf16 pieces only)                        */
    int l1_image_ready;      /* what l1_image already holds for the CURRENT parameters: 0 = nothing (loc_predict builds what it
                                needs), otherwise a value loc_predict_image_mode() returned for an earlier loc_predict call
                                since which neither the parameters nor the BatchNorm statistics changed - the conversion is
                                then skipped (predict_locs predicts twice with the same weights, locator.py:414, :441) */
    const uint8_t* X2;       /* optional: X packed 2 bits per genotype (loc_pack_genotypes_2bit; needs x_max <= 3), same rows.
                                When set, row chunks of at least LOC_GEMM_I8_PACKED_MIN_ROWS rows of an int8 many-row predict
                                read it instead of X (a quarter of the genotype traffic, bit-identical activations)        */
    int64_t x2_pitch;
    int l1_scan_ready;       /* != 0: the header of l1_image already holds loc_l1_quant_scan's per-unit maxima for the CURRENT
                                parameters (loc_predict_scan ran since they last changed): loc_predict's int8 image build skips
                                that pass over W1 */
    loc_tuning tune;
} loc_net;
/* Dynamic-range guard of the int8 digit image (loc_l1_quant_scan).  R_h = max_k |w'| / (1.2533 mean_k |w'|) for unit h, w' =
 * BatchNorm scale x first-layer weight: the largest weight over the rms of a Gaussian bulk with the unit's mean magnitude
 * (the mean, not the rms, so that a single huge weight cannot hide by inflating the yardstick).
 *   two digit planes (16-bit fixed point against the unit's largest weight) while  median_h R_h <= LOC_GUARD_FAST_MEDIAN and
 *   max_h R_h <= LOC_GUARD_FAST_MAX  - a typical weight then keeps >= 9 bits (6 at the worst unit);
 *   three planes (24 bits) while  max_h R_h <= LOC_GUARD_EXACT_MAX  - >= 14 bits for a typical weight, the "no worse than an
 *   fp32 accumulation" regime;
 *   otherwise the weights go through the bf16 x 3 pieces, exact for any fp32 weight.
 * Measured deviations of the predictions from the float64 forward per regime: tests/test_gpu_trained_predict.py, DESIGN.md. */
#define LOC_GUARD_FAST_MEDIAN 64.0f
#define LOC_GUARD_FAST_MAX 512.0f
#define LOC_GUARD_EXACT_MAX 512.0f
/* Rows from which the int8 image + GEMM beats the in-loop-conversion bf16x3 kernel including its once-per-call max pass
 * and conversion (K = 100,000, profiles/r03_gemm_bench.json). */
#define LOC_GEMM_I8_MIN_ROWS(digits) 512
/* rows per launch from which the int8 GEMM reads the 2-bit packed genotypes WHEN THE CALLER SUPPLIES THEM (loc_net.X2).  Round 6,
 * medians of interleaved graph replays (profiles/r06_gemm_packed_crossover.jsonl): rows streaming from HBM 2 % faster at 2048
 * rows, 5-6 % at 3072..8192, 10-12 % at 12,288..16,384; rows repeating a cache-resident matrix 3-6 % SLOWER at every count.
 * Building X2 costs 21 us per 100 MB, so nothing packs on its own (LocatorNet.auto_pack / --predict_packed are opt-in) */
#define LOC_GEMM_I8_PACKED_MIN_ROWS 3072
/* Rows from which image + GEMM beats the in-loop-conversion kernel INCLUDING the once-per-call conversion, measured at
 * K = 100,000 (profiles/r02_bench_default.json: in-loop 0.20 / 0.12 us per row at 3 / 1 pieces; image 51 / 30 us plus
 * 60 + 0.095 / 20 + 0.036 us per row). */
#define LOC_GEMM_MIN_ROWS(pieces) ((pieces) >= 3 ? 1152 : (pieces) == 2 ? 768 : 640)

#define LOC_MAX_FWD_GRID 512

/* BatchNorm gamma/beta update riding in the tail launch of a step (loc_stack_dw_adam_tail): the partial sums the
 * layer-1 backward left in gbs, the six K-vectors, and optionally the next minibatch's batch statistics
 * ([mean|var], from loc_bn_epoch_stats) from which the next step's [scale|shift|mean|rstd] goes to bn4. */
typedef struct loc_gb_tail {
    int K, Kp;
    const float* gbs;
    float *gamma, *beta, *m_gamma, *v_gamma, *m_beta, *v_beta;
    const float* next_stats;
    float* bn4;
} loc_gb_tail;

const char* loc_last_error(void);
int loc_version(void);

/* ---- layout helpers (host only) ---- */
int loc_make_dims(int K, int H, int L, loc_dims* out);
int loc_param_layout(const loc_dims* d, loc_layout* out);
int64_t loc_w1s_index(int h, int k, int Hp);
int64_t loc_workspace_floats(const loc_dims* d);
/* the same for --batch_size up to `batch` rows (> LOC_MAX_BATCH: one activation slot holds ceil(batch / 128) * 128 rows) */
int64_t loc_workspace_floats_batch(const loc_dims* d, int batch);

/* ---- utility kernels ---- */
/* Keras glorot_uniform init of one Dense kernel (locator.py:319-325 [K]): logical R x C (in x out),
 * U(+-sqrt(6/(R+C))) from Philox4x32-10 keyed by (seed, stream_id, r*C+c); stored row-major
 * [Rp][Cp] or (swizzled=1) as W1S with R = SNPs, C = units.  Padding is written as zero. */
int loc_init_glorot(float* dst, int R, int C, int Rp, int Cp, int swizzled, uint64_t seed, uint64_t stream_id,
                    void* stream);
int loc_init_uniform(float* dst, int64_t n, float limit, uint64_t seed, uint64_t stream_id, void* stream);
/* Keep-masks for Dropout (locator.py:321): mask[i] = philox(seed, offset+i) >= p ? 1 : 0. */
int loc_dropout_mask_fill(uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);
/* Bootstrap resample of SNP columns (locator.py:648-653): dst[r][j] = src[r][site_order[j]]. */
int loc_gather_columns(const uint8_t* src, int64_t src_pitch, const int32_t* site_order, int K,
                       uint8_t* dst, int64_t dst_pitch, int n_rows, void* stream);
/* Replicate summarisation, the step right after --windows / --bootstrap (replaces kdepred + centroid of
 * /root/reference/locator_py/plot_locator.py:26-44 and scripts/plot_locator.R:57-113): sample s owns the points
 * xy[offsets[s] .. offsets[s+1]) ([x, y] pairs, float64, map units) = its replicate predictions.  Per sample:
 *   out[4s+2], out[4s+3]  geographic centroid (mean of the points)
 *   peak_index[s]         FIRST index of the maximum of the Gaussian kernel density estimate (bandwidth h) evaluated at the
 *                         points themselves - sklearn KernelDensity(kernel='gaussian', bandwidth=h).score_samples + argmax
 *   out[4s+0], out[4s+1]  that point.  A sample with a non-finite coordinate (or no points) has no estimate: peak_index -1
 *                         and the centroid instead (the reference's `except` branch).
 * One workgroup per sample, float64, fixed summation order (launch-independent results). */
int loc_kde_peak_batch(const double* xy, const int64_t* offsets, int n_samples, double bandwidth, int32_t* peak_index,
                       double* out, void* stream);
/* out[0] = max(out[0], largest byte of X[0..n_rows)[0..K)) (uint32, zero it first): which number formats can carry the
 * genotypes exactly (int8 needs <= 127). */
int loc_genotype_max(const uint8_t* X, int64_t x_pitch, int n_rows, int K, uint32_t* out, void* stream);
/* Row-major (K x H, Keras) <-> swizzled W1S, on device. */
int loc_w1_swizzle(const float* w_kh, int K, int H, float* w1s, int Kp, int Hp, void* stream);
int loc_w1_unswizzle(const float* w1s, int Kp, int Hp, float* w_kh, int K, int H, void* stream);

/* ---- BatchNormalization on the input (locator.py:318) ---- */
/* Training: per-SNP batch mean / biased variance over the n_b rows `rows[0..n_b)`,
 * out4 = [scale | shift | mean | rstd] (4*Kp), and the moving-statistics update. */
int loc_bn_batch_stats(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, int K, int Kp,
                       const float* gamma, const float* beta, float* mov_mean, float* mov_var,
                       float* out4, void* stream);
/* Whole epoch at once (the statistics depend only on X and the permutation): rows_all holds the epoch's
 * n_steps minibatches back to back (`batch` rows each, the last one n_last rows).  Writes
 * stats_ep[step] = [mean | biased var] (n_steps*2*Kp floats), applies the n_steps moving-statistics
 * updates in order, and leaves bn4 = [scale|shift|mean|rstd] of step 0.  Later steps get their bn4 from
 * loc_l1_backward_adam(bn_next_stats = stats_ep + 2*Kp*(step+1)) right after gamma/beta are updated. */
int loc_bn_epoch_stats(const uint8_t* X, int64_t x_pitch, const int32_t* rows_all, int batch, int n_last, int n_steps,
                       int K, int Kp, const float* gamma, const float* beta, float* mov_mean, float* mov_var,
                       float* stats_ep, float* bn4, void* stream);
/* The two halves of loc_bn_epoch_stats on their own: the batch statistics of an epoch's minibatches (they depend on X and
 * the permutation only, so the NEXT epoch's can be computed while this one trains), and the n_steps moving-statistics
 * updates of an epoch (+ step 0's [scale|shift|mean|rstd] into bn4 unless bn4 is NULL: an epoch whose first layer-1
 * forward was chained into the previous epoch's last step finds them there already). */
int loc_bn_epoch_stats_only(const uint8_t* X, int64_t x_pitch, const int32_t* rows_all, int batch, int n_last, int n_steps,
                            int K, int Kp, float* stats_ep, void* stream);
int loc_bn_epoch_finish(int n_steps, int K, int Kp, const float* gamma, const float* beta, float* mov_mean, float* mov_var,
                        const float* stats_ep, float* bn4, void* stream);
/* Inference: scale/shift from the moving statistics. */
int loc_bn_infer_scale_shift(int K, int Kp, const float* gamma, const float* beta, const float* mov_mean,
                             const float* mov_var, float* out4, void* stream);

/* ---- layer 1: Dense(width, elu) on the genotype matrix (locator.py:319-320) ---- */
/* a1[b][h] = ELU(sum_k xhat[b][k] W1[k][h] + b1[h]) for up to 32 rows; xhat = x*scale+shift.
 * partial: grid*32*Hp floats of scratch.  mask/a1_drop non-NULL applies Dropout to this layer. */
int loc_l1_forward(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, const loc_dims* d,
                   const float* scale_shift, const float* w1s, const float* b1, float* partial, int grid,
                   float* a1, float* a1_drop, const uint8_t* mask, float keep_scale, void* stream);
/* The same with Dropout directly on the BatchNorm output (--nlayers 1: floor(1/2) = 0 Dense layers precede the Dropout
 * layer, locator.py:319-323): in_mask = keep flags [32][Kp], xhat -> xhat * mask * keep_scale before the contraction. */
int loc_l1_forward_in_dropout(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, const loc_dims* d,
                              const float* scale_shift, const float* w1s, const float* b1, float* partial, int grid,
                              float* a1, const uint8_t* in_mask, float keep_scale, void* stream);
/* Large-M inference form of the same layer (model.predict / the validation pass of model.fit,
 * locator.py:414, :441, :367-376): a1[m][h] for n rows (any n >= 1; a1 must hold ceil(n/128)*128 rows) on
 * the bf16 matrix pipe.  Each fp32 weight (times the BatchNorm scale of its SNP) is split on chip into
 * `pieces` bf16 values: 3 pieces reproduce it exactly, so with genotypes exact in bf16 every product is
 * exact and the result equals the fp32 contraction up to summation order; 1 or 2 pieces are faster and
 * approximate (2^-9 / 2^-17 relative per weight).  partial: scratch of partial_floats floats (at least
 * ceil(n/128)*128*Hp; more scratch = more SNP groups = more workgroups, up to target_blocks, 0 = 256).
 * loc_l1_rows_supported: whether the double-buffered bf16 tiles of this width fit the 160 KB LDS. */
int loc_l1_rows_supported(int Hp, int pieces);
int loc_l1_forward_rows(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n, const loc_dims* d,
                        const float* scale_shift, const float* w1s, const float* b1, float* partial,
                        int64_t partial_floats, float* a1, int pieces, int target_blocks, const loc_tuning* tune,
                        void* stream);
/* The same contraction for MANY rows with the weight conversion taken out of the K loop (model.predict over all
 * samples, locator.py:414, :441; the --jacknife replicate predictions, :683-747): loc_l1_image_build streams W1S once
 * into `image` -- bf16 tiles of s_k*W1 (`pieces` per weight, 3 = exact) laid out as the GEMM's LDS tiles, plus the
 * per-unit shift term sum_k t_k W1[k][h] -- and loc_l1_forward_gemm multiplies any number of row sets against it on
 * the bf16 matrix pipe (u8 -> bf16 genotype widening is the only vector work in its K loop).  The image is valid
 * until W1, gamma/beta or the BatchNorm statistics change.  Padded width 256 only (loc_l1_gemm_supported);
 * loc_l1_image_bytes = size of `image`.  partial / partial_floats / a1 / target_blocks as for loc_l1_forward_rows. */
int loc_l1_gemm_supported(int Hp, int pieces);
int64_t loc_l1_image_bytes(const loc_dims* d, int pieces);
int loc_l1_image_build(const loc_dims* d, const float* scale_shift, const float* w1s, int pieces, void* image,
                       void* stream);
int loc_l1_forward_gemm(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n, const loc_dims* d,
                        const void* image, int pieces, const float* b1, float* partial, int64_t partial_floats,
                        float* a1, int target_blocks, void* stream);
/* The same contraction on the INT8 matrix pipe (l1_gemm_i8.hip), for genotypes known to lie in 0..127 (x_max): a
 * genotype is an int8 as it stands, so the row operand is never widened; each weight s_k*W1[k][h] is carried as
 * `digits` base-256 signed digits of a fixed-point number scaled per unit by a power of two chosen from the unit's
 * largest weight (loc_l1_image_i8_build runs the scan pass - per-unit max, mean magnitude and shift term - then writes the
 * digit planes; one extra workgroup of that second launch turns the scan's shares into the guard and the shift vector).  Products and
 * sums are exact integers: 3 digits = 24 bits against the unit's largest weight (no worse than fp32 accumulation
 * rounding), 2 digits = 16 bits.  Refuses x_max outside 1..127 and SNP groups long enough to overflow int32
 * (x_max * 128 * SNPs per group >= 2^31).  Other arguments as loc_l1_forward_gemm. */
int loc_l1_gemm_i8_supported(int Hp, int digits);
int64_t loc_l1_image_i8_bytes(const loc_dims* d, int digits);
int loc_l1_image_i8_build(const loc_dims* d, const float* scale_shift, const float* w1s, int digits, void* image,
                          void* stream);
/* The first pass of loc_l1_image_i8_build on its own: per-unit max_k |s_k W1[k][h]| into the image header, and the
 * dynamic-range guard computed from it and the per-unit rms in the same pass over W1: four floats at byte offset
 * loc_l1_image_i8_guard_offset() of `image` = { median_h R_h, max_h R_h, digit planes the guard allows (2, 3, or -1: none -
 * use the bf16 x 3 pieces), the same when the caller insists on the exact mode (3 or -1) }.  image: at least
 * loc_l1_image_i8_bytes(d, 2) bytes.  loc_l1_image_i8_build_scanned = loc_l1_image_i8_build without that pass (the header
 * must hold a scan of the same weights and scale/shift). */
int loc_l1_quant_scan(const loc_dims* d, const float* scale_shift, const float* w1s, void* image, void* stream);
int64_t loc_l1_image_i8_guard_offset(void);
/* Byte offset of the digit-plane tiles inside `image` (they follow the header: shift vector, per-unit step, per-unit max,
 * guard, and the scan's per-workgroup shares); tile (64-SNP block b, plane p) = 16 KB at offset + (b * digits + p) * 16384,
 * laid out [16-SNP chunk][unit][16 SNPs] int8.  For tools and tests that decode an image. */
int64_t loc_l1_image_i8_tiles_offset(const loc_dims* d);
int loc_l1_image_i8_build_scanned(const loc_dims* d, const float* scale_shift, const float* w1s, int digits, void* image,
                                  void* stream);
int loc_l1_forward_gemm_i8(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n, const loc_dims* d,
                           const void* image, int digits, int x_max, const float* b1, float* partial,
                           int64_t partial_floats, float* a1, int target_blocks, const loc_tuning* tune, void* stream);
/* loc_l1_forward_gemm_i8 (packed != 0: loc_l1_forward_gemm_i8_packed, X / x_pitch then are the packed matrix's) WITHOUT its
 * reduction launch: the SNP-group partial sums stay in `partial` as [*h_groups][ceil(n/128)*128][256] floats and *cvec8
 * points at the image's 8 x 256 shift-term slices; loc_stack_forward_eval_partial adds them up (+ b1, ELU) in the input
 * stage of the hidden-stack launch - the same association as the reduction kernel, so the same bits.  h_groups and cvec8
 * are HOST pointers written before the call returns. */
int loc_l1_forward_gemm_i8_partial(const uint8_t* X, int64_t x_pitch, int packed, const int32_t* rows, int n,
                                   const loc_dims* d, const void* image, int digits, int x_max, float* partial,
                                   int64_t partial_floats, int target_blocks, const loc_tuning* tune, int* h_groups,
                                   const float** cvec8, void* stream);
/* Fused: dW1 = xhat^T dZ1, dxhat = dZ1 W1^T -> dgamma/dbeta, Adam on W1/gamma/beta/b1.
 * dW1 and dxhat are never written to memory.  gb_scratch: (Kp/32)*128 floats (per-wave partial
 * sums for dgamma/dbeta, combined in a fixed order by a trailing per-SNP kernel).  If bn_next_stats
 * ([mean|var] of the NEXT minibatch, from loc_bn_epoch_stats) is non-NULL that kernel also writes the next
 * step's [scale|shift|mean|rstd] to bn4_out from the just-updated gamma/beta.  ev_after_main: optional
 * hipEvent_t recorded between the main kernel and the trailing per-SNP kernel (kernel timing).
 * n_b <= 32 rows per weight tile, or up to LOC_MAX_BATCH on the fused-stack widths 64/128/256: every tile then
 * takes the gradient of all ceil(n_b/32) row blocks (dz1 rows [0, 32*ceil(n_b/32))) before its one Adam update. */
int loc_l1_backward_adam(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, const loc_dims* d,
                         const float* bn4, const float* dz1, float* w1s, float* m1s, float* v1s,
                         float* gamma, float* beta, float* m_gamma, float* v_gamma, float* m_beta, float* v_beta,
                         float* b1, float* m_b1, float* v_b1, float* gb_scratch, const float* alpha_tab,
                         int alpha_tab_len, const float* lr, const int* t_base, int t_off, int grid,
                         const float* bn_next_stats, float* bn4_out, void* ev_after_main, const loc_tuning* tune,
                         void* stream);

/* loc_l1_backward_adam for --nlayers 1 (Dropout on the BatchNorm output, see loc_l1_forward_in_dropout): dW1 uses
 * xhat * mask * keep_scale and the gradient reaching gamma / beta is dxhat * mask * keep_scale.  n_b <= 32. */
int loc_l1_backward_adam_in_dropout(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, const loc_dims* d,
                                    const float* bn4, const float* dz1, float* w1s, float* m1s, float* v1s,
                                    float* gamma, float* beta, float* m_gamma, float* v_gamma, float* m_beta,
                                    float* v_beta, float* b1, float* m_b1, float* v_b1, float* gb_scratch,
                                    const float* alpha_tab, int alpha_tab_len, const float* lr, const int* t_base,
                                    int t_off, int grid, const float* bn_next_stats, float* bn4_out,
                                    const loc_tuning* tune, const uint8_t* in_mask, float keep_scale, void* stream);

/* The main kernel of loc_l1_backward_adam alone (W1, b1 and the gamma/beta partial sums in gb_scratch); the
 * caller runs the gamma/beta update itself -- loc_train_step folds it into loc_stack_dw_adam_tail. */
int loc_l1_backward_adam_main(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, const loc_dims* d,
                              const float* bn4, const float* dz1, float* w1s, float* m1s, float* v1s, float* b1,
                              float* m_b1, float* v_b1, float* gb_scratch, const float* alpha_tab, int alpha_tab_len,
                              const float* lr, const int* t_base, int t_off, int grid, const loc_tuning* tune,
                              void* stream);

/* Layer-1 backward + Adam of one minibatch CHAINED with the layer-1 forward of the next one (width padding to 512, 256, 128
 * or 64, n_b <= 32; locator.py:367-376: consecutive steps of model.fit).  One pass over W1 / m / v: Adam on W1 and b1, the
 * BatchNorm gamma / beta update of loc_l1_backward_adam, the next step's [scale|shift|mean|rstd] into bn4 (in place;
 * bn_next_stats = [mean|var] of the next minibatch), and -- rows_next non-NULL -- partial[g][32][256] of the next
 * minibatch's layer-1 pre-activations from the updated weights, g < min(grid, ceil(Kp/32 / s)) workgroups x s =
 * loc_l1_chain_groups_per_workgroup(Hp) groups each (partial[g * s + slot][32][Hp]), for the reduction of loc_l1_forward.  rows_next NULL: backward only (last step of an epoch). */
int loc_l1_chain_supported(int Hp);            /* padded width 512, 256, 128 or 64 */
int loc_l1_chain_groups_per_workgroup(int Hp); /* k-tiles a workgroup owns at a time = partial groups it leaves: 1, 2, 4 */
int loc_l1_backward_adam_chain(const uint8_t* X, int64_t x_pitch, const int32_t* rows, int n_b, const int32_t* rows_next,
                               int n_b_next, const loc_dims* d, float* bn4, const float* bn_next_stats, const float* dz1,
                               float* w1s, float* m1s, float* v1s, float* gamma, float* beta, float* m_gamma,
                               float* v_gamma, float* m_beta, float* v_beta, float* b1, float* m_b1, float* v_b1,
                               const float* alpha_tab, int alpha_tab_len, const float* lr, const int* t_base, int t_off,
                               int grid, float* partial, int64_t partial_floats, const loc_tuning* tune, void* stream);

/* ---- hidden Dense(width, elu) layers + Dropout (locator.py:319-323) ---- */
int loc_dense_forward(const float* in, const float* W, const float* b, int Hp, float* out, float* out_drop,
                      const uint8_t* mask, float keep_scale, void* stream);
/* One backward launch: (a) dz_prev = (dz W^T) * dropmask * ELU'(a_prev);
 * (b) optionally dW/db + Adam for ANOTHER layer `W2` (the one above), whose old
 * weights are no longer needed.  Either half may be skipped with NULL pointers. */
int loc_dense_backward(const float* dz, const float* W, const float* a_prev, const uint8_t* mask,
                       float keep_scale, float* dz_prev, const float* in2, const float* dz2, float* W2,
                       float* mW2, float* vW2, float* b2, float* mb2, float* vb2, int Hp,
                       const float* alpha_tab, int alpha_tab_len, const float* lr, const int* t_base,
                       int t_off, void* stream);

/* ---- Dense(2), Dense(2), euclidean_distance_loss (locator.py:314-315, :324-325) ---- */
/* Training: loss (batch mean, to *loss_out), head gradients + Adam, dz_last = dA * ELU'(a). */
int loc_head_train(const float* a, int Hp, int n_b, const int32_t* rows, const float* Y, float* wa, float* ba,
                   float* wb, float* bb, float* m, float* v, int64_t off_wa, int64_t off_ba, int64_t off_wb,
                   int64_t off_bb, float* dz_last, float* loss_out, const float* alpha_tab, int alpha_tab_len,
                   const float* lr, const int* t_base, int t_off, void* stream);
/* Inference: yhat[b][0..1]; if rows/Y non-NULL also dist[b] = ||yhat - Y[rows[b]]||. */
int loc_head_eval(const float* a, int Hp, int n_b, const float* wa, const float* ba, const float* wb,
                  const float* bb, float* yhat, const int32_t* rows, const float* Y, float* dist, void* stream);

/* ---- fused hidden stack (widths that pad to 64/128/256/512; otherwise the per-layer entry points above) ---- */
int loc_stack_fused_supported(int Hp);
/* WhT[l] = Wh[l]^T for the n_hidden = L-1 hidden kernels. */
int loc_transpose_hidden(const float* Wh, float* WhT, int Hp, int n_hidden, void* stream);
/* ONE launch: layers 2..L forward (+Dropout), Dense(2) x2, per-sample loss, and the whole backward chain down
 * to dz of layer 1.  Row-parallel (2 batch rows per workgroup), no inter-workgroup traffic.  Outputs: acts
 * [L][32][Hp] (ELU outputs; slot 0 = layer 1 is an input), adrop, dz [L][32][Hp], head_out [32][8]. */
int loc_stack_forward_backward(const float* a1_in, const float* Wh, const float* WhT, const float* bh,
                               const float* wa, const float* ba, const float* wb, const float* bb,
                               const uint8_t* mask, float keep_scale, int Hp, int L, int n_pre, int n_b,
                               int slot_rows, const int32_t* rows, const float* Y, float* acts, float* adrop,
                               float* dz, float* head_out, const loc_tuning* tune, void* stream);
/* Inference counterpart: layers 2..L + heads for n_b rows (any n_b; a1 is [n_b][Hp]); yhat[n_b][2], optional dist[n_b]. */
int loc_stack_forward_eval(const float* a1, const float* Wh, const float* bh, const float* wa, const float* ba,
                           const float* wb, const float* bb, int Hp, int L, int n_b, const int32_t* rows,
                           const float* Y, float* yhat, float* dist, void* stream);
/* loc_stack_forward_eval whose layer-1 activations are still the many-row GEMM's group partial sums
 * (loc_l1_forward_gemm_i8_partial): a1[m][h] = ELU(sum_g partial[g * group_stride + m * Hp + h] + sum_s cvec8[s * Hp + h] + b1[h]). */
int loc_stack_forward_eval_partial(const float* partial, int groups, int64_t group_stride, const float* cvec8,
                                   const float* b1, const float* Wh, const float* bh, const float* wa, const float* ba,
                                   const float* wb, const float* bb, int Hp, int L, int n_b, const int32_t* rows,
                                   const float* Y, float* yhat, float* dist, int rows_form, void* stream);
/* Both inference entry points take MANY rows (>= loc_stack_rows_min_rows(), padded width 256) through stack_rows.hip: 32 rows
 * per workgroup on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: fp32 products and sums, the weights streamed once per 32
 * rows), or 16 rows per workgroup (v_mfma_f32_16x16x4_f32) where 32-row tiles would leave compute units idle (up to 4096
 * rows on 256 compute units, and 8193..12288), instead of 2 / 4 / 8 rows per workgroup on the vector ALU (up to 8 x compute
 * units rows).  rows_form (loc_stack_forward_eval_form /
 * loc_stack_forward_eval_partial; loc_tuning.stack_rows in loc_predict): 0 = by row count, 1 = always the 32-row form where
 * supported, 2 = always the 16-row form, -1 / -2 / -3 = 2 / 4 / 8 rows per workgroup on the vector ALU.  Same arithmetic, different summation order: predictions agree to fp32
 * round-off (tests/test_gpu_stack_rows.py). */
int loc_stack_rows_supported(int Hp, int L);
int loc_stack_rows_min_rows(void);
int loc_stack_forward_eval_form(const float* a1, const float* Wh, const float* bh, const float* wa, const float* ba,
                                const float* wb, const float* bb, int Hp, int L, int n_b, const int32_t* rows, const float* Y,
                                float* yhat, float* dist, int rows_form, void* stream);
/* ONE launch: dW, db + Adam for every hidden layer (one workgroup per 32x32 tile, W^T refreshed), head
 * gradients + Adam, and the batch-mean loss (to *loss_out). */
int loc_stack_dw_adam(int Hp, int L, int n_pre, int n_b, int use_drop, const float* acts, const float* adrop,
                      const float* dz, const float* head_out, float* params, float* adam_m, float* adam_v,
                      float* WhT, int64_t off_wh, int64_t off_bh, int64_t off_wa, int64_t off_ba, int64_t off_wb,
                      int64_t off_bb, float* loss_out, const float* alpha_tab, int alpha_tab_len, const float* lr,
                      const int* t_base, int t_off, void* stream);
/* The same launch with the step's other row-reducing tail riding along: extra workgroups apply the BatchNorm
 * gamma/beta Adam update (gb non-NULL; see loc_gb_tail).  Must run after the layer-1 backward of the step. */
int loc_stack_dw_adam_tail(int Hp, int L, int n_pre, int n_b, int slot_rows, int use_drop, const float* acts, const float* adrop,
                           const float* dz, const float* head_out, float* params, float* adam_m, float* adam_v,
                           float* WhT, int64_t off_wh, int64_t off_bh, int64_t off_wa, int64_t off_ba,
                           int64_t off_wb, int64_t off_bb, float* loss_out, const float* alpha_tab,
                           int alpha_tab_len, const float* lr, const int* t_base, int t_off, const loc_gb_tail* gb,
                           void* stream);

/* ---- composites: what model.fit / model.predict enqueue (locator.py:367-376, :414, :441) ---- */
/* One minibatch step: BN stats -> forward -> loss -> backward -> Adam, on rows[0..n_b).
 * n_b <= 32, or <= LOC_MAX_BATCH when net->slot_rows = LOC_BATCH_SLOT (--batch_size > 32: needs the fused-stack
 * widths 64/128/256, epoch-level BN statistics for steps of more than 32 rows, and Dropout not directly after layer 1).
 * mask: keep flags for this step, [rows][Hp] with rows = 32*ceil(n_b/32) - [32][Kp] when L == 1 (Dropout on the
 * BatchNorm output) - (NULL iff drop_p == 0).  loss_out: 1 float.
 * bn_ready != 0: this step's [scale|shift|mean|rstd] is already in the workspace (loc_bn_epoch_stats or the
 * previous step's bn_next_stats) and the per-step statistics kernel is skipped.
 * bn_next_stats: [mean|var] of the next minibatch or NULL (see loc_l1_backward_adam).
 * ev_l1b0 / ev_l1b1: optional hipEvent_t recorded around the layer-1 backward kernel. */
int loc_train_step(const loc_net* net, const int32_t* rows, int n_b, int t_off, const uint8_t* mask,
                   float* loss_out, int bn_ready, const float* bn_next_stats, void* ev_l1b0, void* ev_l1b1,
                   void* stream);
/* The same step inside an epoch whose minibatches are known in advance (model.fit draws the epoch's permutation
 * before its first step): this step's layer-1 backward also produces the NEXT minibatch's layer-1 partial sums
 * (loc_l1_backward_adam_chain), and a step with fwd_done != 0 starts from the partial sums its predecessor left instead
 * of running the layer-1 forward.  The epoch-level BN statistics are required (bn_ready is implied; bn_next_stats as in
 * loc_train_step, mandatory with rows_next).  rows_next NULL = last step of the epoch.  Results equal loc_train_step's
 * up to the summation order of the BatchNorm gamma / beta gradient and of the layer-1 partial sums.
 * loc_train_chain_supported: width padding to 64, 128, 256 or 512, nlayers >= 2, batch <= 32 (<= 64 at width 256: net->slot_rows =
 * 64, two 32-row blocks per step, the partial sums then [group][64][Hp]), Dropout not on the BatchNorm output.
 * CONTRACT: a step with fwd_done != 0 must follow, on the same stream, a chained step whose rows_next / n_b_next /
 * bn_next_stats described it -- the hand-over lives in the workspace (layer-1 partial sums, scale/shift), so nothing that
 * uses net->ws (loc_train_step, loc_bn_epoch_stats, loc_predict without net->ws_predict) may run in between.  The library
 * cannot check this.  What MAY run in between: loc_predict / loc_predict_scan when net->ws_predict gives them their own
 * workspace, loc_bn_epoch_stats_only, loc_bn_epoch_finish with bn4 = NULL, the callback kernels -- which is how the last
 * step of an epoch chains into the first step of the next across the validation sweep (locator_amd/train.py, xchain). */
int loc_train_chain_supported(const loc_net* net);
int loc_train_step_chain(const loc_net* net, const int32_t* rows, int n_b, int t_off, const uint8_t* mask,
                         float* loss_out, const float* bn_next_stats, const int32_t* rows_next, int n_b_next,
                         int fwd_done, void* ev_l1b0, void* ev_l1b1, void* stream);
/* The workspace's bn4 block (where loc_bn_epoch_stats must leave step 0's values). */
float* loc_workspace_bn4(const loc_net* net);
/* Inference forward over n rows (any n >= 0): yhat[n][2]; dist[n] if with_targets.  More than 32 rows go
 * through loc_l1_forward_rows in chunks of LOC_PREDICT_CHUNK (net->predict_pieces), then one hidden-stack
 * launch per chunk; up to 32 rows (or predict_pieces < 0) use the 32-row kernels. */
int loc_predict(const loc_net* net, const int32_t* rows, int n, float* yhat, int with_targets, float* dist,
                void* stream);

/* What a caller that lets the guard choose the digit planes runs before loc_predict when the parameters have changed:
 * BatchNorm inference scale/shift into the workspace, then loc_l1_quant_scan on net->l1_image (needs
 * loc_l1_image_i8_bytes(d, 2) bytes there).  The caller reads the four guard floats back (the one synchronising step of a
 * many-row predict), sets net->predict_digits / predict_pieces accordingly and net->l1_scan_ready = 1. */
int loc_predict_scan(const loc_net* net, void* stream);

/* Which weight image loc_predict builds and uses for n rows of this net: 0 = none (in-loop conversion or the 32-row
 * kernels), 1..3 = bf16 pieces (loc_l1_image_build), 12 / 13 = int8 with 2 / 3 digit planes (loc_l1_image_i8_build). */
int loc_predict_image_mode(const loc_net* net, int n);

/* ---- 2-bit packed genotypes for the many-row int8 GEMM (diploid calls 0 / 1 / 2: locator.py:196-218) ----
 * loc_pack_genotypes_2bit: X2[r][j] = sum_i (X[r][4 j + i] & 3) << 2 i  for a matrix whose values are known to be <= 3
 * (loc_genotype_max); X2 pitch >= Kp / 4, 4-byte aligned.  loc_l1_forward_gemm_i8_packed = loc_l1_forward_gemm_i8 reading
 * that matrix: a quarter of the genotype bytes and HBM lines, expanded to int8 in LDS; results are bit-identical to the
 * unpacked call (integer arithmetic).  loc_predict uses it when net->X2 is set. */
int loc_pack_genotypes_2bit(const uint8_t* X, int64_t x_pitch, int n_rows, int Kp, uint8_t* X2, int64_t x2_pitch,
                            void* stream);
int loc_l1_forward_gemm_i8_packed(const uint8_t* X2, int64_t x2_pitch, const int32_t* rows, int n, const loc_dims* d,
                                  const void* image, int digits, const float* b1, float* partial, int64_t partial_floats,
                                  float* a1, int target_blocks, const loc_tuning* tune, void* stream);

/* ---- filter_snps + the split's transposes on the device, for the --windows loop (locator.py:265-273, :295-308, :539-545) ----
 * gt: the window's calls as the zarr store holds them, int8 [n_variants][n_samples][ploidy] (negative = missing).
 * loc_filter_snps_flags: keep[v] = 1 iff exactly two distinct alleles occur at variant v (allel's is_biallelic of
 * count_alleles) and - unless min_mac == 1, when the reference skips that filter - allele 1 occurs at least min_mac times;
 * pos = exclusive prefix sum of keep, *n_kept = their number (the SNP count K of the window's model).
 * loc_filter_snps_rows: X[r][pos[v]] = number of allele-1 copies of sample sample_order[r] at kept variant v
 * (to_allele_counts()[:, :, 1] transposed and row-gathered: `ac[:, rows].T`), r < n_out; X is [n_out][x_pitch] uint8 and must
 * be zeroed beyond K by the caller.  No --impute_missing / --max_SNPs here (both consume the NumPy stream: host path). */
int loc_filter_snps_flags(const int8_t* gt, int64_t n_variants, int n_samples, int ploidy, int min_mac, uint8_t* keep,
                          int32_t* pos, int32_t* n_kept, void* stream);
int loc_filter_snps_rows(const int8_t* gt, int64_t n_variants, int n_samples, int ploidy, const uint8_t* keep,
                         const int32_t* pos, const int32_t* sample_order, int n_out, uint8_t* X, int64_t x_pitch, void* stream);

/* ---- the three callbacks of a fit, on the device (locator.py:330-362) ----
 * State of ModelCheckpoint(best only) -> EarlyStopping -> ReduceLROnPlateau, all on val_loss, strict '<', min_delta 0.
 * The host fills it once (bests = +inf, waits 0, lr = the fit's starting rate, epoch 0, stopped 0, stop_epoch / best_epoch
 * -1) and uploads it; loc_epoch_callbacks, enqueued after the validation sweep of every epoch, does what the three
 * on_epoch_end calls do, and loc_snapshot_if copies params -> best when that epoch improved val_loss.  With these two in
 * the epoch's stream (or captured graph) the host can enqueue epochs ahead of the device and read the history rows with
 * a lag (locator_amd/train.py); once early stopping has fired the state, the LR and `best` are frozen, so epochs already
 * enqueued behind the stop epoch are harmless and their rows are dropped. */
typedef struct loc_cb_state {
    double ck_best, es_best, rl_best; /* best val_loss seen by each callback                                            */
    float lr;                         /* current learning rate (fp32, as Keras keeps it); mirrored into *lr             */
    float lr_factor;                  /* ReduceLROnPlateau factor (0.5, locator.py:352)                                 */
    int es_wait, rl_wait;
    int patience, lr_patience;        /* --patience; int(patience / 6) (locator.py:343, :354)                           */
    int epoch;                        /* epochs completed                                                               */
    int stopped;                      /* 1 once EarlyStopping fired                                                     */
    int stop_epoch;                   /* index of that epoch, or -1                                                     */
    int best_epoch;                   /* index of the epoch whose weights `best` holds, or -1                           */
    int save_now;                     /* 1 between loc_epoch_callbacks and loc_snapshot_if of an improving epoch        */
    int reserved;
} loc_cb_state;
/* stats: the epoch's `steps` per-minibatch losses followed by its n_val validation distances (what loc_train_step* and
 * loc_predict(with_targets) wrote); the last minibatch has n_last rows, the others `batch`.  Writes hist[4 * epoch ..] =
 * { loss, val_loss, learning rate the epoch trained with, flags: 1 = checkpoint saved, 2 = early stopping fired, 4 = LR
 * reduced } for epoch < hist_cap, unless early stopping fired at an earlier epoch. */
int loc_epoch_callbacks(const float* stats, int steps, int batch, int n_last, int n_val, loc_cb_state* state, float* lr,
                        double* hist, int hist_cap, void* stream);
/* params -> best (n floats, n % 4 == 0, 16-byte aligned) iff state->save_now. */
int loc_snapshot_if(const loc_cb_state* state, const float* params, float* best, int64_t n, void* stream);

/* thin event helpers so a ctypes host can time a kernel on the stream it runs on */
int loc_event_create(void** ev);
int loc_event_create_notiming(void** ev);
int loc_event_destroy(void* ev);
int loc_event_record(void* ev, void* stream);
int loc_event_elapsed_ms(void* ev0, void* ev1, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* LOCATOR_HIP_H */
