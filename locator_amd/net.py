"""Device-side model state for locator's network (reference: load_network,
/root/reference/locator/locator.py:311-327) and thin wrappers over the C-ABI entry points.

torch is plumbing here: it owns device buffers and streams.  All arithmetic is in
liblocator_hip.so.  Layouts are described in include/locator_hip.h and DESIGN.md §3.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

ALPHA_TAB_LEN = 32769            # beyond this t the Adam bias correction is 1 to fp32 precision
ADAM_B1, ADAM_B2 = 0.9, 0.999
LOC_ROWS = 32
LOC_MAX_BATCH = 128      # include/locator_hip.h: four 32-row blocks per step (the row-block kernels)
LOC_BIG_BATCH_MAX = 4096 # include/locator_hip.h: --batch_size limit (above 128: row blocks streamed from L2)
LOC_BATCH_SLOT = 128     # rows per activation slot of the training scratch when batch > 32
LOC_MAX_FWD_GRID = 512
LOC_GEMM_MIN_ROWS = {3: 1152, 2: 768, 1: 640}  # include/locator_hip.h, by bf16 pieces
LOC_GEMM_I8_MIN_ROWS = 512                     # include/locator_hip.h: int8 image + GEMM
LOC_GEMM_I8_PACKED_MIN_ROWS = 3072             # include/locator_hip.h: rows per chunk from which 2-bit packed genotypes pay


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_NCU, _NCU_LOCK = {}, __import__("threading").Lock()


def _compute_units(dev):
    """Compute units of the device, asked once per process (fits on several threads of one process construct their nets at
    the same time; torch's device-property query is not re-entrant on its first use)."""
    key = str(dev)
    with _NCU_LOCK:
        if key not in _NCU:
            _NCU[key] = int(torch.cuda.get_device_properties(dev).multi_processor_count)
        return _NCU[key]


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.LocatorHipError("locator_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU fallback.")


def upload_genotypes(x_nk, device="cuda:0"):
    """(n_samples, K) integer allele counts -> device uint8 [n_samples][Kp] with zero padding.
    This is the layout `traingen` already has in the reference (locator.py:303)."""
    x_nk = np.ascontiguousarray(x_nk)
    n, K = x_nk.shape
    Kp = (K + 31) // 32 * 32
    host = np.zeros((n, Kp), np.uint8)
    host[:, :K] = x_nk.astype(np.uint8, copy=False)
    return torch.from_numpy(host).to(device)


class LocatorNet:
    """BatchNormalization -> nlayers x Dense(width, elu) with Dropout in the middle -> Dense(2) -> Dense(2),
    Adam, Euclidean loss — the model of locator.py:311-327, resident on one GPU."""

    def __init__(self, X, Y, K, width=256, nlayers=10, dropout_prop=0.25, seed=0, replicate=0, device="cuda:0",
                 predict_pieces=3, predict_digits=3, tuning=None):
        """predict_digits: int8 digit planes per weight in the many-row inference forward: 0 = AUTO (the default of the
        command line): two planes (16-bit fixed point against each unit's largest weight) while the dynamic-range guard
        of the weights allows it (include/locator_hip.h, LOC_GUARD_*: measured 5e-5 relative on the predictions of the
        converged metric fit), else three, else the exact bf16 pieces; 3 = three planes (24 bits: no worse than an fp32
        accumulation) under the same guard's exact limit; 2 = two planes unconditionally; -1 = never use the int8 pipe.
        predict_pieces: bf16 pieces per weight where the int8 GEMM does not apply - few rows, genotypes above 127 -
        (3 = exact fp32 products; 1 or 2 trade accuracy for speed, -1 keeps every row block on the 32-row fp32-MFMA
        kernel).  tuning: dict of loc_tuning fields (include/locator_hip.h) - speed hints and measurement switches,
        never results."""
        require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.d = _lib.make_dims(K, width, nlayers)
        self.lay = _lib.param_layout(self.d)
        assert X.dtype == torch.uint8 and X.is_cuda and X.shape[1] == self.d.Kp and X.is_contiguous()
        assert Y.dtype == torch.float32 and Y.is_cuda and Y.shape[1] == 2 and Y.is_contiguous()
        self.X, self.Y = X, Y
        self.drop_p = float(dropout_prop)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.replicate = int(replicate)
        dev = self.device
        self.params = torch.zeros(self.lay.n_total, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(self.lay.n_trainable, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(self.lay.n_trainable, dtype=torch.float32, device=dev)
        self.best = None
        t = np.arange(ALPHA_TAB_LEN, dtype=np.float64)
        tab = np.ones(ALPHA_TAB_LEN)
        tab[1:] = np.sqrt(1.0 - ADAM_B2 ** t[1:]) / (1.0 - ADAM_B1 ** t[1:])
        self.alpha_tab = torch.from_numpy(tab.astype(np.float32)).to(dev)
        self.lr_t = torch.full((1,), 1e-3, dtype=torch.float32, device=dev)
        self.t_base_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ws = torch.empty(self.lib.loc_workspace_floats(C.byref(self.d)), dtype=torch.float32, device=dev)
        self.ws_predict = None             # second workspace for predicts (train.EpochRunner(xchain=True) allocates it)
        # transposed hidden kernels for the fused backward chain (derived state, see refresh_transposed)
        # --nlayers 1 has no hidden stack (and its Dropout acts on the BatchNorm output): per-layer kernels
        self.use_fused = bool(self.lib.loc_stack_fused_supported(self.d.Hp)) and self.d.L >= 2
        self.wht = (torch.zeros((self.d.L - 1) * self.d.Hp * self.d.Hp, dtype=torch.float32, device=dev)
                    if self.use_fused else None)
        nkt = self.d.Kp // 32
        ncu = _compute_units(dev)
        self.l1_fwd_grid = max(1, min(nkt, ncu, LOC_MAX_FWD_GRID))
        self.l1_bwd_grid = max(1, 2 * ncu)          # 2 blocks x 4 waves per CU, all resident
        self.predict_pieces = int(predict_pieces)
        self.predict_digits = int(predict_digits)
        self.tuning = _lib.Tuning(**{k: int(v) for k, v in (tuning or {}).items()})
        self.l1_image = None               # image of s_k*W1 for many-row predicts (allocated on first use)
        self._image_mode = 0               # loc_predict_image_mode() of the image l1_image holds for the current parameters
        self._guard = None                 # (median R, max R, digits allowed, digits allowed in exact mode) of the current parameters
        # many-row predicts build a 2-bit packed copy of the matrix on their own only when asked (--predict_packed /
        # pack_genotypes()): measured in round 6 (profiles/r06_gemm_packed_crossover.jsonl, medians of interleaved replays) the
        # packed GEMM is 2-12 % faster on rows that stream from HBM and 3-6 % SLOWER on rows that repeat a cache-resident matrix,
        # while the packing pass costs 21 us per 100 MB - eight 4096-row predicts' worth of the gain.  Rounds 4-5 packed from
        # 3072 rows automatically; a matrix that is predicted from once (every CLI flow) lost by it.
        self.auto_pack = False
        self.slot_rows = LOC_ROWS          # rows per activation slot; set_batch() widens it for --batch_size > 32
        self._net = None
        self.init_weights()

    @property
    def mask_width(self):
        """Keep flags per row of one step's dropout mask: Hp, or Kp when the Dropout layer sits on the BatchNorm output
        (--nlayers 1, locator.py:319-323)."""
        return self.d.Kp if self.d.n_pre == 0 else self.d.Hp

    def params_changed(self):
        """Weights, gamma/beta or BatchNorm moving statistics are about to change: a kept many-row weight image is stale."""
        self._image_mode = 0
        self._guard = None

    def set_batch(self, batch_size):
        """Rows per training step (--batch_size).  Up to 32 rows use the 32-row kernels; 33..128 rows run two to
        four row blocks per weight tile (same weight traffic per step) on a 128-row activation scratch; above 128 the
        layer-1 backward streams its row blocks from L2 and the scratch grows to the batch (correct, not tuned)."""
        if not 1 <= batch_size <= LOC_BIG_BATCH_MAX:
            raise ValueError(f"--batch_size must be in 1..{LOC_BIG_BATCH_MAX} for the HIP path (got {batch_size})")
        if batch_size > LOC_ROWS:
            if self.d.L < 2:
                raise ValueError("--batch_size > 32 needs --nlayers >= 2")
            if not self.use_fused or self.d.Hp > 256:
                raise ValueError("--batch_size > 32 needs a --width that pads to 64, 128 or 256 "
                                 "(33..64, 97..128 or 225..256)")
            if self.drop_p > 0 and self.d.n_pre < 2:
                raise ValueError("--batch_size > 32 with dropout needs --nlayers >= 4")
        if batch_size > LOC_MAX_BATCH:
            self.slot_rows = (batch_size + 127) // 128 * 128
            need = self.lib.loc_workspace_floats_batch(C.byref(self.d), int(batch_size))
            if self.ws.numel() < need:
                self.ws = torch.empty(need, dtype=torch.float32, device=self.device)
                if self.ws_predict is not None:
                    self.ws_predict = torch.empty_like(self.ws)
        else:
            # 33..64 rows: 64 tells the library that steps of two 32-row blocks may be chained (loc_train_chain_supported);
            # the scratch layout is the 128-row one either way
            self.slot_rows = LOC_ROWS if batch_size <= LOC_ROWS else (64 if batch_size <= 64 else LOC_BATCH_SLOT)
        self._net = None
        return self.slot_rows

    # ------------------------------------------------------------------ C struct
    def cnet(self):
        n = _lib.Net()
        n.d = self.d
        n.params, n.adam_m, n.adam_v = self.params.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr()
        n.alpha_tab, n.alpha_tab_len = self.alpha_tab.data_ptr(), ALPHA_TAB_LEN
        n.lr, n.t_base = self.lr_t.data_ptr(), self.t_base_t.data_ptr()
        n.X, n.x_pitch, n.Y = self.X.data_ptr(), self.X.stride(0), self.Y.data_ptr()
        n.drop_p = self.drop_p
        n.wht = self.wht.data_ptr() if self.wht is not None else None
        n.ws = self.ws.data_ptr()
        n.ws_predict = self.ws_predict.data_ptr() if self.ws_predict is not None else None
        n.l1_fwd_grid, n.l1_bwd_grid = self.l1_fwd_grid, self.l1_bwd_grid
        n.slot_rows = self.slot_rows
        n.predict_pieces = self.predict_pieces
        n.predict_digits = self.predict_digits if self.predict_digits != 0 else 3
        n.x_max = int(getattr(self.X, "loc_x_max", 0))      # set by genotype_max() on the tensor object itself
        if self.l1_image is not None:
            n.l1_image, n.l1_image_bytes = self.l1_image.data_ptr(), self.l1_image.numel()
        x2 = getattr(self.X, "loc_x2", None)                # set by pack_genotypes() on the tensor object itself
        if x2 is not None:
            n.X2, n.x2_pitch = x2.data_ptr(), x2.stride(0)
        n.tune = self.tuning
        self._net = n
        return n

    # ------------------------------------------------------------------ parameters
    def _sec(self, name, n):
        off = getattr(self.lay, name)
        return self.params[off:off + n]

    def init_weights(self):
        """Keras defaults [K]: glorot_uniform kernels, zero biases, gamma=1, beta=0, moving mean 0 / var 1
        (SURVEY.md A.1).  Streams are keyed by (seed, replicate, layer) so a replicate's init does not
        depend on which GPU or in which order it runs."""
        d, lay, lib = self.d, self.lay, self.lib
        self.params_changed()
        self.params.zero_()
        self.adam_m.zero_()
        self.adam_v.zero_()
        self._sec("gamma", d.K).fill_(1.0)
        self._sec("mov_var", d.K).fill_(1.0)
        sid = lambda layer: (self.replicate << 16) | layer
        p = self.params.data_ptr()
        st = _stream()
        _lib.check(lib.loc_init_glorot(p + 4 * lay.w1, d.K, d.H, d.Kp, d.Hp, 1, self.seed, sid(0), st), "init W1")
        for i in range(d.L - 1):
            _lib.check(lib.loc_init_glorot(p + 4 * (lay.wh + i * d.Hp * d.Hp), d.H, d.H, d.Hp, d.Hp, 0, self.seed,
                                           sid(1 + i), st), "init Wh")
        _lib.check(lib.loc_init_glorot(p + 4 * lay.wa, d.H, 2, d.Hp, 2, 0, self.seed, sid(d.L), st), "init Wa")
        _lib.check(lib.loc_init_glorot(p + 4 * lay.wb, 2, 2, 2, 2, 0, self.seed, sid(d.L + 1), st), "init Wb")
        self.t_base_t.zero_()
        self.lr_t.fill_(1e-3)
        self.refresh_transposed()

    def refresh_transposed(self):
        """WhT[l] = Wh[l]^T; needed whenever the hidden kernels are written by anything but the Adam kernel."""
        if self.wht is not None and self.d.L >= 2:
            _lib.check(self.lib.loc_transpose_hidden(self.params.data_ptr() + 4 * self.lay.wh, self.wht.data_ptr(),
                                                     self.d.Hp, self.d.L - 1, _stream()), "loc_transpose_hidden")

    def _export_flat(self, flat, with_moving=True):
        """Flat device buffer -> dict of NumPy arrays in Keras orientation, un-padded (the format the tests compare against)."""
        d, lay, lib = self.d, self.lay, self.lib
        out = {}
        w1 = torch.empty((d.K, d.H), dtype=torch.float32, device=self.device)
        _lib.check(lib.loc_w1_unswizzle(flat.data_ptr() + 4 * lay.w1, d.Kp, d.Hp, w1.data_ptr(), d.K, d.H, _stream()),
                   "unswizzle")
        W = [w1.cpu().numpy()]
        b = [flat[lay.b1:lay.b1 + d.H].cpu().numpy()]
        for i in range(d.L - 1):
            wh = flat[lay.wh + i * d.Hp * d.Hp: lay.wh + (i + 1) * d.Hp * d.Hp].view(d.Hp, d.Hp)
            W.append(wh[:d.H, :d.H].cpu().numpy().copy())
            b.append(flat[lay.bh + i * d.Hp: lay.bh + i * d.Hp + d.H].cpu().numpy())
        W.append(flat[lay.wa:lay.wa + 2 * d.Hp].view(d.Hp, 2)[:d.H].cpu().numpy().copy())
        b.append(flat[lay.ba:lay.ba + 2].cpu().numpy())
        W.append(flat[lay.wb:lay.wb + 4].view(2, 2).cpu().numpy().copy())
        b.append(flat[lay.bb:lay.bb + 2].cpu().numpy())
        out["W"], out["b"] = W, b
        out["gamma"] = flat[lay.gamma:lay.gamma + d.K].cpu().numpy()
        out["beta"] = flat[lay.beta:lay.beta + d.K].cpu().numpy()
        if with_moving:
            out["mov_mean"] = flat[lay.mov_mean:lay.mov_mean + d.K].cpu().numpy()
            out["mov_var"] = flat[lay.mov_var:lay.mov_var + d.K].cpu().numpy()
        return out

    def export_params(self):
        return self._export_flat(self.params)

    def export_adam(self):
        return self._export_flat(self.adam_m, False), self._export_flat(self.adam_v, False)

    def _import_flat(self, flat, p, with_moving=True):
        d, lay, lib = self.d, self.lay, self.lib
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        flat.zero_()
        w1 = f32(p["W"][0])
        _lib.check(lib.loc_w1_swizzle(w1.data_ptr(), d.K, d.H, flat.data_ptr() + 4 * lay.w1, d.Kp, d.Hp, _stream()),
                   "swizzle")
        torch.cuda.current_stream().synchronize()
        flat[lay.b1:lay.b1 + d.H] = f32(p["b"][0])
        for i in range(d.L - 1):
            wh = flat[lay.wh + i * d.Hp * d.Hp: lay.wh + (i + 1) * d.Hp * d.Hp].view(d.Hp, d.Hp)
            wh[:d.H, :d.H] = f32(p["W"][1 + i])
            flat[lay.bh + i * d.Hp: lay.bh + i * d.Hp + d.H] = f32(p["b"][1 + i])
        flat[lay.wa:lay.wa + 2 * d.Hp].view(d.Hp, 2)[:d.H] = f32(p["W"][d.L])
        flat[lay.ba:lay.ba + 2] = f32(p["b"][d.L])
        flat[lay.wb:lay.wb + 4] = f32(p["W"][d.L + 1]).reshape(-1)
        flat[lay.bb:lay.bb + 2] = f32(p["b"][d.L + 1])
        flat[lay.gamma:lay.gamma + d.K] = f32(p["gamma"])
        flat[lay.beta:lay.beta + d.K] = f32(p["beta"])
        if with_moving:
            flat[lay.mov_mean:lay.mov_mean + d.K] = f32(p["mov_mean"])
            flat[lay.mov_var:lay.mov_var + d.K] = f32(p["mov_var"])

    def check_params(self, p, with_moving=True):
        """Shapes of an oracle-format parameter dict against this net, BEFORE any device call: the swizzle kernel
        reads W[0] as K x H whatever the file holds, so a file from another SNP count / --width / --nlayers would be
        an out-of-bounds device read."""
        d = self.d
        W, b = list(p["W"]), list(p["b"])
        if len(W) != d.L + 2 or len(b) != d.L + 2:
            raise ValueError(f"weights hold {len(W)} Dense layers; this network has {d.L + 2} "
                             f"(--nlayers {d.L} + Dense(2) + Dense(2))")
        want = [(d.K, d.H)] + [(d.H, d.H)] * (d.L - 1) + [(d.H, 2), (2, 2)]
        for i, (w, bi, shp) in enumerate(zip(W, b, want)):
            if tuple(np.shape(w)) != shp or tuple(np.shape(bi)) != (shp[1],):
                raise ValueError(f"dense_{i}: kernel {tuple(np.shape(w))} / bias {tuple(np.shape(bi))} in the weights, "
                                 f"this network needs {shp} / ({shp[1]},) (SNPs {d.K}, --width {d.H}, --nlayers {d.L})")
        for k in ("gamma", "beta") + (("mov_mean", "mov_var") if with_moving else ()):
            if tuple(np.shape(p[k])) != (d.K,):
                raise ValueError(f"{k}: length {np.shape(p[k])} in the weights, this network has {d.K} SNPs")

    def import_params(self, p):
        self.check_params(p)
        self.params_changed()
        self._import_flat(self.params, p)
        self.refresh_transposed()

    # ------------------------------------------------------------------ ops
    def train_step(self, rows, n_b, t_off, mask, loss_out, ev0=None, ev1=None, bn_ready=False, bn_next=None):
        """One minibatch step on X[rows[:n_b]].  rows: int32 device tensor (>= n_b entries),
        mask: uint8 device tensor [32*Hp] of keep flags ([32*Kp] when nlayers == 1: mask_width) or None, loss_out:
        1-element float32 view.
        bn_ready / bn_next: epoch-level BN statistics (see epoch_bn_stats)."""
        self.params_changed()
        net = self._net or self.cnet()
        _lib.check(self.lib.loc_train_step(C.byref(net), _ptr(rows), int(n_b), int(t_off), _ptr(mask),
                                           _ptr(loss_out), 1 if bn_ready else 0, _ptr(bn_next), ev0, ev1, _stream()),
                   "loc_train_step")

    def chain_supported(self):
        """True when consecutive steps of an epoch may be chained (loc_train_step_chain: the layer-1 backward of step t
        also produces the layer-1 forward of step t + 1): width padding to 64 / 128 / 256 / 512, nlayers >= 2, batch <= 32 (<= 64 at
        width 256), Dropout not
        on the BatchNorm output."""
        net = self._net or self.cnet()
        return bool(self.lib.loc_train_chain_supported(C.byref(net)))

    def train_step_chain(self, rows, n_b, t_off, mask, loss_out, bn_next, rows_next, n_b_next, fwd_done,
                         ev0=None, ev1=None):
        """train_step inside an epoch with epoch-level BN statistics: rows_next / n_b_next / bn_next describe the next
        minibatch (None for the epoch's last step); fwd_done = the previous chained step already left this
        minibatch's layer-1 partial sums."""
        self.params_changed()
        net = self._net or self.cnet()
        _lib.check(self.lib.loc_train_step_chain(C.byref(net), _ptr(rows), int(n_b), int(t_off), _ptr(mask),
                                                 _ptr(loss_out), _ptr(bn_next), _ptr(rows_next), int(n_b_next),
                                                 1 if fwd_done else 0, ev0, ev1, _stream()), "loc_train_step_chain")

    def epoch_bn_stats(self, rows_all, batch, n_last, n_steps, stats_ep):
        """BN batch statistics of every minibatch of the epoch in one launch, the epoch's moving-statistics
        updates, and step 0's scale/shift (loc_bn_epoch_stats)."""
        self.params_changed()
        net = self._net or self.cnet()
        d, lay, P = self.d, self.lay, self.params.data_ptr()
        _lib.check(self.lib.loc_bn_epoch_stats(self.X.data_ptr(), self.X.stride(0), _ptr(rows_all), int(batch),
                                               int(n_last), int(n_steps), d.K, d.Kp, P + 4 * lay.gamma,
                                               P + 4 * lay.beta, P + 4 * lay.mov_mean, P + 4 * lay.mov_var,
                                               _ptr(stats_ep), self.lib.loc_workspace_bn4(C.byref(net)), _stream()),
                   "loc_bn_epoch_stats")

    def epoch_bn_stats_only(self, rows_all, batch, n_last, n_steps, stats_ep):
        """Only the batch statistics of an epoch's minibatches (loc_bn_epoch_stats_only): nothing of the model changes."""
        d = self.d
        _lib.check(self.lib.loc_bn_epoch_stats_only(self.X.data_ptr(), self.X.stride(0), _ptr(rows_all), int(batch), int(n_last),
                                                    int(n_steps), d.K, d.Kp, _ptr(stats_ep), _stream()), "loc_bn_epoch_stats_only")

    def epoch_bn_finish(self, n_steps, stats_ep):
        """Only the epoch's moving-statistics updates (loc_bn_epoch_finish with bn4 = NULL)."""
        self.params_changed()
        d, lay, P = self.d, self.lay, self.params.data_ptr()
        _lib.check(self.lib.loc_bn_epoch_finish(int(n_steps), d.K, d.Kp, P + 4 * lay.gamma, P + 4 * lay.beta,
                                                P + 4 * lay.mov_mean, P + 4 * lay.mov_var, _ptr(stats_ep), None, _stream()),
                   "loc_bn_epoch_finish")

    def predict_rows(self, rows, n, yhat, dist=None, in_fit=False):
        """Inference-mode forward for X[rows[:n]] into yhat [n,2]; dist [n] = distance to Y.
        in_fit: the call is an epoch's validation sweep (train.EpochRunner.enqueue): its arithmetic must not depend on
        whether the epoch is being captured, replayed or enqueued eagerly - val_loss drives the strict-'<' checkpoint /
        early-stopping / LR decisions - so the dynamic-range guard (a host read-back per set of weights), the automatic
        packing and the kept image are all out: many validation rows (>= 512) take three digit planes unconditionally."""
        lib, d = self.lib, self.d
        capturing = torch.cuda.is_current_stream_capturing() or bool(in_fit)
        digits = self.predict_digits if self.predict_digits != 0 else 3
        use_i8 = (self.predict_pieces > 0 and digits > 0 and n >= LOC_GEMM_I8_MIN_ROWS
                  and lib.loc_l1_gemm_i8_supported(d.Hp, digits))
        if use_i8 and self.genotype_max() > 127:
            use_i8 = False
        guarded = use_i8 and self.predict_digits in (0, 3) and not capturing
        need = 0
        if use_i8:
            need = lib.loc_l1_image_i8_bytes(C.byref(d), 3 if self.predict_digits == 0 else digits)
        elif (self.predict_pieces > 0 and n >= LOC_GEMM_MIN_ROWS.get(self.predict_pieces, 1 << 30)
              and lib.loc_l1_gemm_supported(d.Hp, self.predict_pieces)):
            need = lib.loc_l1_image_bytes(C.byref(d), self.predict_pieces)
        if guarded:
            need = max(need, lib.loc_l1_image_bytes(C.byref(d), 3))      # the guard may send the weights to bf16 x 3
        if need and (self.l1_image is None or self.l1_image.numel() < need):
            # many rows: W1 is converted once per call into this buffer (include/locator_hip.h, loc_net.l1_image)
            self.l1_image = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._image_mode = 0           # a fresh buffer holds nothing
            self._guard = None
            self._net = None
        net = self._net or self.cnet()
        net.l1_scan_ready = 0
        if guarded:
            # the dynamic-range guard decides the digit planes (LOC_GUARD_*): one pass over W1 that the image build would
            # make anyway, four floats read back - the one synchronising step of a many-row predict, once per set of weights
            fresh = self._guard is None
            g = self.quant_guard()
            allowed = int(g[2] if self.predict_digits == 0 else g[3])
            net.predict_digits = allowed                   # 2, 3 or -1 (bf16 x 3 pieces)
            if allowed < 0:
                net.predict_pieces = 3
            net.l1_scan_ready = 1 if (fresh and allowed > 0) else 0   # the header still holds this scan
        if (use_i8 and net.predict_digits > 0 and self.auto_pack and n >= LOC_GEMM_I8_PACKED_MIN_ROWS and not capturing
                and getattr(self.X, "loc_x2", None) is None and self.genotype_max() <= 3):
            self.pack_genotypes()          # one pass, kept on the matrix: every later many-row predict reads a quarter of the bytes
            keep = (net.predict_digits, net.predict_pieces, net.l1_scan_ready)
            net = self.cnet()
            net.predict_digits, net.predict_pieces, net.l1_scan_ready = keep
        # predict_locs predicts twice with the same weights (locator.py:414, :441): the second call finds the image
        mode = lib.loc_predict_image_mode(C.byref(net), int(n))
        net.l1_image_ready = mode if (mode and mode == self._image_mode and not capturing) else 0
        try:
            _lib.check(self.lib.loc_predict(C.byref(net), _ptr(rows), int(n), _ptr(yhat), 1 if dist is not None else 0,
                                            _ptr(dist), _stream()), "loc_predict")
            if mode:
                self._image_mode = mode
        finally:
            # per-call choices never outlive the call, whether it succeeded or raised (the struct is cached)
            net.predict_digits = self.predict_digits if self.predict_digits != 0 else 3
            net.predict_pieces = self.predict_pieces
            net.l1_scan_ready = 0                 # (l1_image_ready is set afresh by every call and records what the last one used)

    def quant_guard(self):
        """(median R, largest R, digit planes allowed, digit planes allowed in exact mode) of the current parameters, R_h =
        largest / rms scaled first-layer weight of unit h (loc_l1_quant_scan; include/locator_hip.h LOC_GUARD_*).  Synchronises
        the current stream; cached until the parameters change."""
        if self._guard is None:
            lib, d = self.lib, self.d
            need = lib.loc_l1_image_i8_bytes(C.byref(d), 2)
            if self.l1_image is None or self.l1_image.numel() < need:
                self.l1_image = torch.empty(max(need, lib.loc_l1_image_i8_bytes(C.byref(d), 3)), dtype=torch.uint8, device=self.device)
                self._image_mode = 0
                self._net = None
            net = self._net or self.cnet()
            _lib.check(lib.loc_predict_scan(C.byref(net), _stream()), "loc_predict_scan")
            off = int(lib.loc_l1_image_i8_guard_offset())
            g = self.l1_image[off:off + 16].view(torch.float32).cpu().numpy()
            self._guard = tuple(float(v) for v in g)
            self._image_mode = 0           # the scan borrowed the image's shift-term section
        return self._guard

    def genotype_max(self):
        """Largest genotype value of the current X (one streaming pass, cached per matrix): decides whether the rows
        are int8 operands as they stand (include/locator_hip.h, loc_net.x_max)."""
        if getattr(self.X, "loc_x_max", None) is None:
            out = torch.zeros(1, dtype=torch.int32, device=self.device)
            _lib.check(self.lib.loc_genotype_max(self.X.data_ptr(), self.X.stride(0), self.X.shape[0], self.d.K,
                                                 out.data_ptr(), _stream()), "loc_genotype_max")
            self.X.loc_x_max = int(out.item())     # X is never modified in place (resamples make new tensors)
            self._net = None
        return self.X.loc_x_max

    def pack_genotypes(self):
        """Keep a 2-bit packed copy of X (four genotypes per byte) next to it: many-row predicts of at least
        LOC_GEMM_I8_PACKED_MIN_ROWS (3072) rows per chunk then stream a quarter of the genotype bytes, with bit-identical
        activations.  One pass over X (21 us per 100 MB) against 11 us saved per 4096-row predict and 67 us per 16,384-row
        one over rows that stream from HBM: worth it for a large matrix predicted from about ten times or more (opt-in).
        Returns False (and does nothing) when X holds values above 3."""
        if getattr(self.X, "loc_x2", None) is not None:
            return True
        if self.genotype_max() > 3:
            return False
        x2 = torch.zeros((self.X.shape[0], self.d.Kp // 4), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.loc_pack_genotypes_2bit(self.X.data_ptr(), self.X.stride(0), self.X.shape[0], self.d.Kp,
                                                    x2.data_ptr(), x2.stride(0), _stream()), "loc_pack_genotypes_2bit")
        self.X.loc_x2 = x2
        self._net = None
        return True

    def fill_dropout_masks(self, mask_buf, n, offset):
        _lib.check(self.lib.loc_dropout_mask_fill(_ptr(mask_buf), int(n), self.drop_p,
                                                  (self.seed ^ 0x64726F70) + (self.replicate << 40), int(offset),
                                                  _stream()), "loc_dropout_mask_fill")

    def snapshot(self):
        """ModelCheckpoint(save_best_only, save_weights_only) as a device-side copy (locator.py:332-348)."""
        if self.best is None:
            self.best = torch.empty_like(self.params)
        self.best.copy_(self.params)

    def restore_best(self):
        """model.load_weights(best) (locator.py:379-388)."""
        if self.best is not None:
            self.params_changed()
            self.params.copy_(self.best)
            self.refresh_transposed()


def gather_columns(X, site_order, K):
    """Bootstrap resample of SNP columns on device (locator.py:648-653)."""
    lib = _lib.load()
    so = torch.as_tensor(np.asarray(site_order, dtype=np.int32)).to(X.device)
    out = torch.zeros_like(X)
    _lib.check(lib.loc_gather_columns(_ptr(X), X.stride(0), _ptr(so), int(K), _ptr(out), out.stride(0),
                                      X.shape[0], _stream()), "loc_gather_columns")
    return out


def filter_snps_device(gt, sample_order, min_mac=2):
    """filter_snps (locator.py:265-273, no --impute_missing / --max_SNPs) and the split's `ac[:, rows].T` (:295-308) on the
    device.  gt: int8 device tensor [n_variants][n_samples][ploidy] (the window's zarr slice, uploaded as it is);
    sample_order: the rows wanted, in output order (train | validation | prediction).  Returns (X uint8 [len(order)][Kp],
    K).  One host synchronisation: the SNP count decides the allocation."""
    lib = _lib.load()
    assert gt.dtype == torch.int8 and gt.is_cuda and gt.is_contiguous() and gt.dim() == 3
    V, N, P = (int(v) for v in gt.shape)
    dev = gt.device
    keep = torch.empty(V, dtype=torch.uint8, device=dev)
    pos = torch.empty(V, dtype=torch.int32, device=dev)
    nk = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.loc_filter_snps_flags(_ptr(gt), V, N, P, int(min_mac), _ptr(keep), _ptr(pos), _ptr(nk), _stream()),
               "loc_filter_snps_flags")
    K = int(nk.item())
    order = torch.as_tensor(np.asarray(sample_order, dtype=np.int32)).to(dev)
    Kp = (max(K, 1) + 31) // 32 * 32
    X = torch.zeros((len(order), Kp), dtype=torch.uint8, device=dev)
    if K > 0:
        _lib.check(lib.loc_filter_snps_rows(_ptr(gt), V, N, P, _ptr(keep), _ptr(pos), _ptr(order), len(order), _ptr(X),
                                            X.stride(0), _stream()), "loc_filter_snps_rows")
    return X, K
