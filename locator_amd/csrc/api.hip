// C-ABI glue: layout helpers, error plumbing, and the composite step / predict entry points
// that stand where model.fit's inner step and model.predict stand in the reference
// (/root/reference/locator/locator.py:367-376, :414, :441).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "stack_tail.h"

static thread_local char g_err[512] = "";

void loc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* loc_last_error(void) { return g_err; }
extern "C" int loc_version(void) { return 1; }

extern "C" int loc_make_dims(int K, int H, int L, loc_dims* out) {
    if (K < 1 || H < 1 || H > LOC_MAX_WIDTH || L < 1) {
        loc_set_error("loc_make_dims: need K >= 1, 1 <= width <= %d, nlayers >= 1 (got K=%d H=%d L=%d)", LOC_MAX_WIDTH, K, H,
                      L);
        return -1;
    }
    out->K = K;
    out->Kp = (K + 31) / 32 * 32;
    out->H = H;
    out->Hp = (H + 31) / 32 * 32;
    out->L = L;
    out->n_pre = L / 2;
    return 0;
}

extern "C" int loc_param_layout(const loc_dims* d, loc_layout* o) {
    int64_t off = 0;
    const int64_t Kp = d->Kp, Hp = d->Hp, L = d->L;
    o->w1 = off;    off += Hp * Kp;
    o->gamma = off; off += Kp;
    o->beta = off;  off += Kp;
    o->b1 = off;    off += Hp;
    o->wh = off;    off += (L - 1) * Hp * Hp;
    o->bh = off;    off += (L - 1) * Hp;
    o->wa = off;    off += Hp * 2;
    o->ba = off;    off += 2;
    o->wb = off;    off += 4;
    o->bb = off;    off += 2;
    off = (off + 3) / 4 * 4;
    o->n_trainable = off;
    o->mov_mean = off; off += Kp;
    o->mov_var = off;  off += Kp;
    o->n_total = off;
    return 0;
}

extern "C" int64_t loc_w1s_index(int h, int k, int Hp) { return w1s_index(h, k, Hp / 32); }

struct ws_view {
    float *bn4, *gbs, *partial, *acts, *adrop, *dz, *head_out, *a1_rows;
    int64_t partial_floats;
};
// scratch of the layer-1 forward partial sums: the 32-row kernel needs grid*32*Hp, the large-M kernel
// LOC_ROWS_BLOCKS tiles of 128*Hp
static int64_t partial_floats_of(const loc_dims* d) {
    const int64_t a = (int64_t)LOC_MAX_FWD_GRID * 32 * d->Hp, b = (int64_t)LOC_ROWS_BLOCKS * LOC_ROWS_TILE * d->Hp;
    return a > b ? a : b;
}
// The per-step scratch (activations, dz, head outputs) exists twice, selected by step parity (a step's tail
// launch may still be reading its activations when a future overlapped schedule starts the next forward).  One activation slot holds `slot` rows (32, or
// LOC_BATCH_SLOT when --batch_size > 32); the workspace is always sized for the larger one.
// (--batch_size > LOC_MAX_BATCH: the slot is the batch rounded up to 128 rows and the workspace comes from
// loc_workspace_floats_batch.)
static int slot_of(const loc_net* net) {
    if (net->slot_rows > LOC_BATCH_SLOT) return (net->slot_rows + 127) / 128 * 128;
    return net->slot_rows > LOC_ROWS ? LOC_BATCH_SLOT : LOC_ROWS;
}
static int slot_cap_of(const loc_net* net) { const int s = slot_of(net); return s > LOC_BATCH_SLOT ? s : LOC_BATCH_SLOT; }
static int64_t per_step_floats(const loc_dims* d, int slot) {
    return (2 * (int64_t)d->L + 1) * slot * d->Hp + 8 * (int64_t)slot;
}
static ws_view carve(const loc_dims* d, float* ws, int parity = 0, int slot = LOC_ROWS, int cap = LOC_BATCH_SLOT) {
    ws_view v;
    const int64_t blk = (int64_t)slot * d->Hp;
    const int64_t per_step = per_step_floats(d, slot);
    v.bn4 = ws;
    v.gbs = v.bn4 + 4 * (int64_t)d->Kp;
    v.partial = v.gbs + 4 * (int64_t)d->Kp;
    v.partial_floats = partial_floats_of(d);
    v.acts = v.partial + v.partial_floats + (parity & 1) * per_step;
    v.adrop = v.acts + d->L * blk;
    v.dz = v.adrop + blk;
    v.head_out = v.dz + d->L * blk;
    v.a1_rows = v.partial + v.partial_floats + 2 * per_step_floats(d, cap);
    return v;
}
extern "C" int64_t loc_workspace_floats_batch(const loc_dims* d, int batch) {
    const int cap = batch > LOC_BATCH_SLOT ? (batch + 127) / 128 * 128 : LOC_BATCH_SLOT;
    return 8 * (int64_t)d->Kp + partial_floats_of(d) + 2 * per_step_floats(d, cap) + (int64_t)LOC_PREDICT_CHUNK * d->Hp;
}
extern "C" int64_t loc_workspace_floats(const loc_dims* d) { return loc_workspace_floats_batch(d, LOC_BATCH_SLOT); }

#define TRY(x)                 \
    do {                       \
        int rc__ = (x);        \
        if (rc__) return rc__; \
    } while (0)

extern "C" float* loc_workspace_bn4(const loc_net* net) { return carve(&net->d, net->ws).bn4; }
// rows one --batch_size step may carry with this net's scratch
static int max_rows_of(const loc_net* net) {
    const int s = slot_of(net);
    return s > LOC_BATCH_SLOT ? (s < LOC_BIG_BATCH_MAX ? s : LOC_BIG_BATCH_MAX) : (s > LOC_ROWS ? LOC_MAX_BATCH : LOC_ROWS);
}

// workgroups of the chained layer-1 kernel: one 8-wave workgroup where the plain backward runs two of 4 waves
static int chain_grid_of(const loc_net* net) {
    const int ktw = loc_l1_chain_groups_per_workgroup(net->d.Hp) > 0 ? loc_l1_chain_groups_per_workgroup(net->d.Hp) : 1;
    const int nkt = net->d.Kp / 32, n_super = (nkt + ktw - 1) / ktw;      // a workgroup owns ktw k-tiles at a time
    int g = net->l1_bwd_grid / 2;
    if (g < 1) g = 1;
    return g > n_super ? n_super : g;
}
// layer-1 partial groups a chained step leaves for the reduction (workgroups x k-tile slots per workgroup)
static int chain_groups_of(const loc_net* net) { return chain_grid_of(net) * loc_l1_chain_groups_per_workgroup(net->d.Hp); }
// 32-row blocks a chained step carries: 1, or 2 for --batch_size 33..64 (net->slot_rows = 64; width 256 only)
#define LOC_CHAIN_MAX_ROWS 64
static int chain_rb_of(const loc_net* net) { return net->slot_rows > LOC_ROWS ? 2 : 1; }

extern "C" int loc_train_chain_supported(const loc_net* net) {
    const loc_dims* d = &net->d;
    const bool in_drop = net->drop_p > 0.f && d->n_pre == 0;
    return d->L >= 2 && net->wht && loc_stack_fused_supported(d->Hp) && loc_l1_chain_supported(d->Hp) &&
           (net->slot_rows <= LOC_ROWS || (net->slot_rows <= LOC_CHAIN_MAX_ROWS && d->Hp == 256)) && !in_drop &&
           (int64_t)chain_groups_of(net) * 32 * chain_rb_of(net) * d->Hp <= partial_floats_of(d) &&
           (int64_t)d->Kp * 1024 < ((int64_t)1 << 32) && (net->x_pitch % 16) == 0;
}

static int train_step_impl(const loc_net* net, const int32_t* rows, int n_b, int t_off, const uint8_t* mask,
                           float* loss_out, int bn_ready, const float* bn_next_stats, void* ev_l1b0, void* ev_l1b1,
                           bool chain, const int32_t* rows_next, int n_b_next, int fwd_done, void* stream) {
    const loc_dims* d = &net->d;
    const int slot = slot_of(net);
    const int max_b = max_rows_of(net);
    if (n_b < 1 || n_b > max_b) { loc_set_error("loc_train_step: n_b=%d out of 1..%d", n_b, max_b); return -1; }
    const bool use_drop = net->drop_p > 0.f;
    if (use_drop && !mask) { loc_set_error("loc_train_step: dropout_prop > 0 needs a keep mask"); return -1; }
    const float ks = use_drop ? 1.0f / (1.0f - net->drop_p) : 1.0f;
    loc_layout lay;
    loc_param_layout(d, &lay);
    float *P = net->params, *M = net->adam_m, *V = net->adam_v;
    ws_view w = carve(d, net->ws, t_off & 1, slot, slot_cap_of(net));
    const int Hp = d->Hp, L = d->L, npre = d->n_pre;
    const int64_t blk = (int64_t)slot * Hp, HH = (int64_t)Hp * Hp;
    auto act = [&](int l) { return w.acts + (l - 1) * blk; };        // ELU output of layer l (1-based)
    auto dzl = [&](int l) { return w.dz + (l - 1) * blk; };          // dLoss/dz of layer l
    auto in_of = [&](int l) { return (use_drop && l - 1 == npre) ? w.adrop : act(l - 1); };  // input of layer l >= 2
    const float* at = net->alpha_tab;
    const int atl = net->alpha_tab_len;

    // --nlayers 1: no hidden Dense layer, and the Dropout layer sits directly on the BatchNorm output (floor(1/2) = 0
    // Dense layers before it, locator.py:319-323): the keep mask is [rows][Kp] and goes into the layer-1 kernels
    const bool in_drop = use_drop && npre == 0;
    const bool fused = L >= 2 && net->wht && loc_stack_fused_supported(Hp);
    if (slot > LOC_ROWS) {
        // --batch_size > 32: the step is linear in the rows (BatchNorm is the first layer, so its batch statistics
        // depend on the data only), hence the same kernels with two 32-row blocks per weight tile
        if (!fused || (use_drop && npre == 1) || !loc_l1_rows_supported(Hp, 3)) {
            loc_set_error("loc_train_step: --batch_size > 32 needs width 64/128/256 (after padding) and nlayers >= 4 "
                          "when dropout is on");
            return -1;
        }
        if (!bn_ready && n_b > LOC_ROWS) {
            loc_set_error("loc_train_step: more than 32 rows need the epoch-level BN statistics (loc_bn_epoch_stats)");
            return -1;
        }
    }
    if (!bn_ready)
        TRY(loc_bn_batch_stats(net->X, net->x_pitch, rows, n_b, d->K, d->Kp, P + lay.gamma, P + lay.beta,
                               P + lay.mov_mean, P + lay.mov_var, w.bn4, stream));
    if (chain && fwd_done) {
        // the previous step's chained kernel left this minibatch's layer-1 partial sums: only add them up
        const bool dr = use_drop && npre == 1;
        TRY(loc_l1_reduce_launch_drop(w.partial, chain_groups_of(net), 32 * chain_rb_of(net), Hp, P + lay.b1, act(1),
                                      dr ? w.adrop : nullptr, dr ? mask : nullptr, ks, stream));
    } else if (n_b > LOC_ROWS) {
        // large-M forward, exact fp32 products (3 bf16 pieces); fills whole 128-row tiles of the activation slot
        TRY(loc_l1_forward_rows(net->X, net->x_pitch, rows, n_b, d, w.bn4, P + lay.w1, P + lay.b1, w.partial,
                                w.partial_floats, act(1), 3, 0, &net->tune, stream));
    } else if (in_drop) {
        TRY(loc_l1_forward_in_dropout(net->X, net->x_pitch, rows, n_b, d, w.bn4, P + lay.w1, P + lay.b1, w.partial,
                                      net->l1_fwd_grid, act(1), mask, ks, stream));
    } else {
        const bool dr = use_drop && npre == 1;
        TRY(loc_l1_forward(net->X, net->x_pitch, rows, n_b, d, w.bn4, P + lay.w1, P + lay.b1, w.partial,
                           net->l1_fwd_grid, act(1), dr ? w.adrop : nullptr, dr ? mask : nullptr, ks, stream));
    }
    if (fused) {
        // fused row-parallel hidden stack -> layer-1 backward -> ONE tail launch for everything that reduces over the
        // batch rows: hidden-layer dW/db + Adam, heads, batch loss, and the BatchNorm gamma/beta update (plus the next
        // step's scale/shift).  The hidden tail only needs what the stack kernel left in scratch, so it can sit after
        // the layer-1 backward and share a launch with the gamma/beta tail.
        TRY(loc_stack_forward_backward(in_of(2), P + lay.wh, net->wht, P + lay.bh, P + lay.wa, P + lay.ba, P + lay.wb,
                                       P + lay.bb, use_drop ? mask : nullptr, ks, Hp, L, npre, n_b, slot, rows, net->Y,
                                       w.acts, w.adrop, w.dz, w.head_out, &net->tune, stream));
        if (ev_l1b0) (void)hipEventRecord((hipEvent_t)ev_l1b0, (hipStream_t)stream);
        if (chain) {
            // layer-1 backward + Adam (W1, b1, gamma, beta, the next step's scale/shift) and -- rows_next given -- the
            // next minibatch's layer-1 forward partial sums from the weights while they are in registers
            // ... and, unless tune.chain_tail says otherwise, the step's hidden-layer / head Adam tail as trailing
            // workgroups of the same launch (it only needs what the stack kernel left in scratch)
            const bool merged = net->tune.chain_tail >= 0;
            loc_dw_tail_args ta;
            ta.L = L; ta.n_pre = npre; ta.n_b = n_b; ta.use_drop = use_drop ? 1 : 0;
            ta.acts = w.acts; ta.adrop = w.adrop; ta.dz = w.dz; ta.head_out = w.head_out;
            ta.P = P; ta.M = M; ta.V = V; ta.WhT = net->wht;
            ta.off_wh = lay.wh; ta.off_bh = lay.bh; ta.off_wa = lay.wa; ta.off_ba = lay.ba; ta.off_wb = lay.wb; ta.off_bb = lay.bb;
            ta.loss_out = loss_out; ta.alpha_tab = at; ta.alpha_tab_len = atl; ta.lr = net->lr; ta.t_base = net->t_base;
            ta.t_off = t_off; ta.slot_rows = slot;
            TRY(l1_chain_launch(net->X, net->x_pitch, rows, n_b, rows_next, n_b_next, d, w.bn4, bn_next_stats, dzl(1),
                                P + lay.w1, M + lay.w1, V + lay.w1, P + lay.gamma, P + lay.beta, M + lay.gamma,
                                V + lay.gamma, M + lay.beta, V + lay.beta, P + lay.b1, M + lay.b1, V + lay.b1, at, atl,
                                net->lr, net->t_base, t_off, chain_grid_of(net), w.partial, w.partial_floats, &net->tune,
                                merged ? &ta : nullptr, chain_rb_of(net), stream));
            if (ev_l1b1) (void)hipEventRecord((hipEvent_t)ev_l1b1, (hipStream_t)stream);
            if (!merged)
                TRY(loc_stack_dw_adam_tail(Hp, L, npre, n_b, slot, use_drop ? 1 : 0, w.acts, w.adrop, w.dz, w.head_out, P,
                                           M, V, net->wht, lay.wh, lay.bh, lay.wa, lay.ba, lay.wb, lay.bb, loss_out, at,
                                           atl, net->lr, net->t_base, t_off, nullptr, stream));
            return 0;
        }
        TRY(loc_l1_backward_adam_main(net->X, net->x_pitch, rows, n_b, d, w.bn4, dzl(1), P + lay.w1, M + lay.w1,
                                      V + lay.w1, P + lay.b1, M + lay.b1, V + lay.b1, w.gbs, at, atl, net->lr,
                                      net->t_base, t_off, net->l1_bwd_grid, &net->tune, stream));
        if (ev_l1b1) (void)hipEventRecord((hipEvent_t)ev_l1b1, (hipStream_t)stream);
        loc_gb_tail gb;
        gb.K = d->K; gb.Kp = d->Kp; gb.gbs = w.gbs;
        gb.gamma = P + lay.gamma; gb.beta = P + lay.beta;
        gb.m_gamma = M + lay.gamma; gb.v_gamma = V + lay.gamma;
        gb.m_beta = M + lay.beta; gb.v_beta = V + lay.beta;
        gb.next_stats = bn_next_stats; gb.bn4 = w.bn4;
        TRY(loc_stack_dw_adam_tail(Hp, L, npre, n_b, slot, use_drop ? 1 : 0, w.acts, w.adrop, w.dz, w.head_out, P, M, V,
                                   net->wht, lay.wh, lay.bh, lay.wa, lay.ba, lay.wb, lay.bb, loss_out, at, atl, net->lr,
                                   net->t_base, t_off, &gb, stream));
        return 0;
    }
    for (int l = 2; l <= L; ++l) {
        const bool dr = use_drop && l == npre;
        TRY(loc_dense_forward(in_of(l), P + lay.wh + (l - 2) * HH, P + lay.bh + (int64_t)(l - 2) * Hp, Hp, act(l),
                              dr ? w.adrop : nullptr, dr ? mask : nullptr, ks, stream));
    }
    TRY(loc_head_train(act(L), Hp, n_b, rows, net->Y, P + lay.wa, P + lay.ba, P + lay.wb, P + lay.bb, M, V, lay.wa,
                       lay.ba, lay.wb, lay.bb, dzl(L), loss_out, at, atl, net->lr, net->t_base, t_off, stream));
    for (int l = L; l >= 2; --l) {
        // dx through layer l; dW/Adam for layer l+1 (its dx was produced by the previous launch)
        const bool dr = use_drop && l - 1 == npre;
        const bool dw = l + 1 <= L;
        TRY(loc_dense_backward(dzl(l), P + lay.wh + (l - 2) * HH, act(l - 1), dr ? mask : nullptr, ks, dzl(l - 1),
                               dw ? in_of(l + 1) : nullptr, dw ? dzl(l + 1) : nullptr,
                               dw ? P + lay.wh + (l - 1) * HH : nullptr, dw ? M + lay.wh + (l - 1) * HH : nullptr,
                               dw ? V + lay.wh + (l - 1) * HH : nullptr,
                               dw ? P + lay.bh + (int64_t)(l - 1) * Hp : nullptr,
                               dw ? M + lay.bh + (int64_t)(l - 1) * Hp : nullptr,
                               dw ? V + lay.bh + (int64_t)(l - 1) * Hp : nullptr, Hp, at, atl, net->lr, net->t_base,
                               t_off, stream));
    }
    // dW/Adam for layer 2
    if (L >= 2)
        TRY(loc_dense_backward(nullptr, nullptr, nullptr, nullptr, ks, nullptr, in_of(2), dzl(2), P + lay.wh,
                               M + lay.wh, V + lay.wh, P + lay.bh, M + lay.bh, V + lay.bh, Hp, at, atl, net->lr,
                               net->t_base, t_off, stream));
    if (ev_l1b0) (void)hipEventRecord((hipEvent_t)ev_l1b0, (hipStream_t)stream);
    if (in_drop) {
        TRY(loc_l1_backward_adam_in_dropout(net->X, net->x_pitch, rows, n_b, d, w.bn4, dzl(1), P + lay.w1, M + lay.w1,
                                            V + lay.w1, P + lay.gamma, P + lay.beta, M + lay.gamma, V + lay.gamma,
                                            M + lay.beta, V + lay.beta, P + lay.b1, M + lay.b1, V + lay.b1, w.gbs, at,
                                            atl, net->lr, net->t_base, t_off, net->l1_bwd_grid, bn_next_stats, w.bn4,
                                            &net->tune, mask, ks, stream));
        if (ev_l1b1) (void)hipEventRecord((hipEvent_t)ev_l1b1, (hipStream_t)stream);
        return 0;
    }
    TRY(loc_l1_backward_adam(net->X, net->x_pitch, rows, n_b, d, w.bn4, dzl(1), P + lay.w1, M + lay.w1, V + lay.w1,
                             P + lay.gamma, P + lay.beta, M + lay.gamma, V + lay.gamma, M + lay.beta, V + lay.beta,
                             P + lay.b1, M + lay.b1, V + lay.b1, w.gbs, at, atl, net->lr, net->t_base, t_off,
                             net->l1_bwd_grid, bn_next_stats, w.bn4, ev_l1b1, &net->tune, stream));
    return 0;
}

extern "C" int loc_train_step(const loc_net* net, const int32_t* rows, int n_b, int t_off, const uint8_t* mask,
                              float* loss_out, int bn_ready, const float* bn_next_stats, void* ev_l1b0,
                              void* ev_l1b1, void* stream) {
    return train_step_impl(net, rows, n_b, t_off, mask, loss_out, bn_ready, bn_next_stats, ev_l1b0, ev_l1b1, false,
                           nullptr, 0, 0, stream);
}

extern "C" int loc_train_step_chain(const loc_net* net, const int32_t* rows, int n_b, int t_off, const uint8_t* mask,
                                    float* loss_out, const float* bn_next_stats, const int32_t* rows_next,
                                    int n_b_next, int fwd_done, void* ev_l1b0, void* ev_l1b1, void* stream) {
    if (!loc_train_chain_supported(net)) {
        loc_set_error("loc_train_step_chain: needs a width that pads to 64, 128, 256 or 512, nlayers >= 2, --batch_size <= 32 (<= 64 "
                      "at width 256) and no Dropout on the BatchNorm output (loc_train_chain_supported)");
        return -1;
    }
    if (n_b > LOC_ROWS * chain_rb_of(net)) {
        loc_set_error("loc_train_step_chain: n_b=%d out of 1..%d", n_b, LOC_ROWS * chain_rb_of(net));
        return -1;
    }
    if (rows_next && !bn_next_stats) {
        loc_set_error("loc_train_step_chain: rows_next needs the next minibatch's batch statistics");
        return -1;
    }
    return train_step_impl(net, rows, n_b, t_off, mask, loss_out, 1, bn_next_stats, ev_l1b0, ev_l1b1, true, rows_next,
                           n_b_next, fwd_done, stream);
}

extern "C" int loc_predict_image_mode(const loc_net* net, int n) {
    const loc_dims* d = &net->d;
    const int Hp = d->Hp;
    const int pieces = net->predict_pieces == 0 ? 3 : net->predict_pieces;
    const int digits = net->predict_digits == 0 ? 3 : net->predict_digits;
    if (!(n > LOC_ROWS && pieces > 0 && loc_stack_fused_supported(Hp) && loc_l1_rows_supported(Hp, pieces)) || !net->l1_image)
        return 0;
    const int64_t snps_max = (int64_t)d->Kp + 256;          // one SNP group at most: the whole K range
    if (digits > 0 && net->x_max >= 1 && net->x_max <= 127 && n >= LOC_GEMM_I8_MIN_ROWS(digits) &&
        loc_l1_gemm_i8_supported(Hp, digits) && net->l1_image_bytes >= loc_l1_image_i8_bytes(d, digits) &&
        (int64_t)net->x_max * 128 * snps_max < ((int64_t)1 << 31))
        return 10 + digits;
    if (n >= LOC_GEMM_MIN_ROWS(pieces) && loc_l1_gemm_supported(Hp, pieces) &&
        net->l1_image_bytes >= loc_l1_image_bytes(d, pieces))
        return pieces;
    return 0;
}

extern "C" int loc_predict_scan(const loc_net* net, void* stream) {
    const loc_dims* d = &net->d;
    if (!net->l1_image || net->l1_image_bytes < loc_l1_image_i8_bytes(d, 2)) {
        loc_set_error("loc_predict_scan: net->l1_image must hold loc_l1_image_i8_bytes(d, 2) bytes");
        return -1;
    }
    loc_layout lay;
    loc_param_layout(d, &lay);
    const float* P = net->params;
    ws_view w = carve(d, net->ws_predict ? net->ws_predict : net->ws, 0, LOC_ROWS, slot_cap_of(net));
    TRY(loc_bn_infer_scale_shift(d->K, d->Kp, P + lay.gamma, P + lay.beta, P + lay.mov_mean, P + lay.mov_var, w.bn4, stream));
    return loc_l1_quant_scan(d, w.bn4, P + lay.w1, net->l1_image, stream);
}

extern "C" int loc_predict(const loc_net* net, const int32_t* rows, int n, float* yhat, int with_targets,
                           float* dist, void* stream) {
    if (n <= 0) return 0;
    const loc_dims* d = &net->d;
    loc_layout lay;
    loc_param_layout(d, &lay);
    const float* P = net->params;
    ws_view w = carve(d, net->ws_predict ? net->ws_predict : net->ws, 0, LOC_ROWS, slot_cap_of(net));
    const int Hp = d->Hp, L = d->L;
    const int64_t blk = 32 * (int64_t)Hp, HH = (int64_t)Hp * Hp;
    TRY(loc_bn_infer_scale_shift(d->K, d->Kp, P + lay.gamma, P + lay.beta, P + lay.mov_mean, P + lay.mov_var,
                                 w.bn4, stream));
    const int pieces = net->predict_pieces == 0 ? 3 : net->predict_pieces;
    const int digits = net->predict_digits == 0 ? 3 : net->predict_digits;
    if (n > LOC_ROWS && pieces > 0 && loc_stack_fused_supported(Hp) && loc_l1_rows_supported(Hp, pieces)) {
        // large-M layer 1 on the matrix pipe, then ONE row-parallel stack launch per chunk.  Many rows: the weights are
        // converted once into the caller's image buffer (or found there: l1_image_ready) and every chunk runs a pure-MFMA
        // GEMM - on the int8 pipe when the genotypes are known to fit (x_max <= 127) and no SNP group can overflow
        // int32, else on bf16 pieces
        const int mode = loc_predict_image_mode(net, n);
        const bool i8 = mode >= 12, gemm = mode >= 1 && mode <= 3;
        if (mode && net->l1_image_ready != mode) {
            if (i8 && net->l1_scan_ready) TRY(loc_l1_image_i8_build_scanned(d, w.bn4, P + lay.w1, digits, net->l1_image, stream));
            else if (i8) TRY(loc_l1_image_i8_build(d, w.bn4, P + lay.w1, digits, net->l1_image, stream));
            else TRY(loc_l1_image_build(d, w.bn4, P + lay.w1, pieces, net->l1_image, stream));
        }
        for (int c0 = 0; c0 < n; c0 += LOC_PREDICT_CHUNK) {
            const int nc = n - c0 < LOC_PREDICT_CHUNK ? n - c0 : LOC_PREDICT_CHUNK;
            if (i8 && L >= 2 && net->tune.gemm_reduce == 1) {
                // measurement switch: int8 GEMM -> hidden stack with the group reduction in the stack kernel's input stage
                // (measured slower than the dedicated reduction launch, include/locator_hip.h)
                const bool pk = net->X2 && net->x_max <= 3 && nc >= LOC_GEMM_I8_PACKED_MIN_ROWS;
                int groups = 0;
                const float* cvec8 = nullptr;
                TRY(loc_l1_forward_gemm_i8_partial(pk ? net->X2 : net->X, pk ? net->x2_pitch : net->x_pitch, pk ? 1 : 0,
                                                   rows + c0, nc, d, net->l1_image, digits, net->x_max, w.partial,
                                                   w.partial_floats, 0, &net->tune, &groups, &cvec8, stream));
                const int64_t mp = (int64_t)((nc + LOC_ROWS_TILE - 1) / LOC_ROWS_TILE) * LOC_ROWS_TILE;
                TRY(loc_stack_forward_eval_partial(w.partial, groups, mp * Hp, cvec8, P + lay.b1, P + lay.wh, P + lay.bh,
                                                   P + lay.wa, P + lay.ba, P + lay.wb, P + lay.bb, Hp, L, nc,
                                                   with_targets ? rows + c0 : nullptr, with_targets ? net->Y : nullptr,
                                                   yhat + 2 * (int64_t)c0, with_targets ? dist + c0 : nullptr,
                                                   net->tune.stack_rows, stream));
                continue;
            }
            if (i8 && net->X2 && net->x_max <= 3 && nc >= LOC_GEMM_I8_PACKED_MIN_ROWS)
                TRY(loc_l1_forward_gemm_i8_packed(net->X2, net->x2_pitch, rows + c0, nc, d, net->l1_image, digits,
                                                  P + lay.b1, w.partial, w.partial_floats, w.a1_rows, 0, &net->tune, stream));
            else if (i8)
                TRY(loc_l1_forward_gemm_i8(net->X, net->x_pitch, rows + c0, nc, d, net->l1_image, digits, net->x_max,
                                           P + lay.b1, w.partial, w.partial_floats, w.a1_rows, 0, &net->tune, stream));
            else if (gemm)
                TRY(loc_l1_forward_gemm(net->X, net->x_pitch, rows + c0, nc, d, net->l1_image, pieces, P + lay.b1,
                                        w.partial, w.partial_floats, w.a1_rows, 0, stream));
            else
                TRY(loc_l1_forward_rows(net->X, net->x_pitch, rows + c0, nc, d, w.bn4, P + lay.w1, P + lay.b1,
                                        w.partial, w.partial_floats, w.a1_rows, pieces, 0, &net->tune, stream));
            if (L == 1) {                 // no hidden layer: Dense(2), Dense(2) straight on a1, 32 rows per launch
                for (int i = 0; i < nc; i += LOC_ROWS) {
                    const int nb = nc - i < LOC_ROWS ? nc - i : LOC_ROWS;
                    TRY(loc_head_eval(w.a1_rows + (int64_t)i * Hp, Hp, nb, P + lay.wa, P + lay.ba, P + lay.wb, P + lay.bb,
                                      yhat + 2 * (int64_t)(c0 + i), with_targets ? rows + c0 + i : nullptr,
                                      with_targets ? net->Y : nullptr, with_targets ? dist + c0 + i : nullptr, stream));
                }
                continue;
            }
            TRY(loc_stack_forward_eval_form(w.a1_rows, P + lay.wh, P + lay.bh, P + lay.wa, P + lay.ba, P + lay.wb,
                                            P + lay.bb, Hp, L, nc, with_targets ? rows + c0 : nullptr,
                                            with_targets ? net->Y : nullptr, yhat + 2 * (int64_t)c0,
                                            with_targets ? dist + c0 : nullptr, net->tune.stack_rows, stream));
        }
        return 0;
    }
    if (L >= 2 && loc_stack_fused_supported(Hp)) {
        // layer 1 per 32-row block into consecutive scratch slots (the L activation slots hold 32*L rows),
        // then ONE row-parallel stack launch for the whole chunk
        const int chunk = LOC_ROWS * L;
        for (int c0 = 0; c0 < n; c0 += chunk) {
            const int nc = n - c0 < chunk ? n - c0 : chunk;
            for (int i = 0; i < nc; i += LOC_ROWS) {
                const int nb = nc - i < LOC_ROWS ? nc - i : LOC_ROWS;
                TRY(loc_l1_forward(net->X, net->x_pitch, rows + c0 + i, nb, d, w.bn4, P + lay.w1, P + lay.b1,
                                   w.partial, net->l1_fwd_grid, w.acts + (int64_t)(i / LOC_ROWS) * blk, nullptr,
                                   nullptr, 1.f, stream));
            }
            TRY(loc_stack_forward_eval(w.acts, P + lay.wh, P + lay.bh, P + lay.wa, P + lay.ba, P + lay.wb,
                                       P + lay.bb, Hp, L, nc, with_targets ? rows + c0 : nullptr,
                                       with_targets ? net->Y : nullptr, yhat + 2 * (int64_t)c0,
                                       with_targets ? dist + c0 : nullptr, stream));
        }
        return 0;
    }
    for (int i = 0; i < n; i += LOC_ROWS) {
        const int nb = n - i < LOC_ROWS ? n - i : LOC_ROWS;
        TRY(loc_l1_forward(net->X, net->x_pitch, rows + i, nb, d, w.bn4, P + lay.w1, P + lay.b1, w.partial,
                           net->l1_fwd_grid, w.acts, nullptr, nullptr, 1.f, stream));
        for (int l = 2; l <= L; ++l)
            TRY(loc_dense_forward(w.acts + (l - 2) * blk, P + lay.wh + (l - 2) * HH,
                                  P + lay.bh + (int64_t)(l - 2) * Hp, Hp, w.acts + (l - 1) * blk, nullptr, nullptr,
                                  1.f, stream));
        TRY(loc_head_eval(w.acts + (L - 1) * blk, Hp, nb, P + lay.wa, P + lay.ba, P + lay.wb, P + lay.bb,
                          yhat + 2 * (int64_t)i, with_targets ? rows + i : nullptr, with_targets ? net->Y : nullptr,
                          with_targets ? dist + i : nullptr, stream));
    }
    return 0;
}

extern "C" int loc_event_create_notiming(void** ev) {
    hipEvent_t e;
    hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) { loc_set_error("hipEventCreateWithFlags: %s", hipGetErrorString(rc)); return (int)rc; }
    *ev = (void*)e;
    return 0;
}
extern "C" int loc_event_create(void** ev) {
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) { loc_set_error("hipEventCreate: %s", hipGetErrorString(rc)); return (int)rc; }
    *ev = (void*)e;
    return 0;
}
extern "C" int loc_event_destroy(void* ev) { return (int)hipEventDestroy((hipEvent_t)ev); }
extern "C" int loc_event_record(void* ev, void* stream) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream); }
extern "C" int loc_event_elapsed_ms(void* ev0, void* ev1, float* ms) {
    hipError_t rc = hipEventSynchronize((hipEvent_t)ev1);
    if (rc == hipSuccess) rc = hipEventElapsedTime(ms, (hipEvent_t)ev0, (hipEvent_t)ev1);
    if (rc != hipSuccess) { loc_set_error("hipEventElapsedTime: %s", hipGetErrorString(rc)); return (int)rc; }
    return 0;
}
