"""model.fit + the three Keras callbacks, MI355X edition (reference: load_callbacks / train_network,
/root/reference/locator/locator.py:330-394; semantics restated in SURVEY.md A.4/A.5).

One epoch = ceil(n_train / batch) minibatch steps (partial last batch kept) + one validation
sweep in inference mode + the three callbacks, all on the device: `loc_epoch_callbacks` turns the epoch's per-step
losses and validation distances into loss / val_loss and runs ModelCheckpoint / EarlyStopping / ReduceLROnPlateau on
a small state struct in HBM, `loc_snapshot_if` copies the parameters when val_loss improved.  The whole epoch is
captured once into a HIP graph and replayed; the host (`FitLoop`) only draws and uploads the next permutation, fills
the dropout masks, replays - up to `depth` epochs ahead of the device - and reads one 32-byte history row per epoch
with a lag.  `Callbacks` below is the host restatement of the same state machines (the tests replay the device's
decisions through it); `EpochRunner.run_epoch` is the one-epoch-at-a-time form the parity tests drive.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from .net import LocatorNet


# HIP stream-capture mode of the epoch graphs.  Fits that share a process (replicates.py: one thread and stream each) capture
# at different times, and the other thread keeps allocating, launching and waiting on events meanwhile.
CAPTURE_ERROR_MODE = "thread_local"


class _DeviceLock:
    """Process-wide mutex that makes HIP-graph capture safe when several fits share a process (one thread and stream each).

    Round 4 found capture process-sensitive even in thread-local capture mode: a sibling thread's device-wide
    synchronize made the capture fail (hipErrorStreamCaptureUnsupported), and a sibling destroying a graph / an event (a
    finished fit going away) aborted the process now and then.  All of those calls live OUTSIDE a fit's steady epoch loop -
    building the net and its buffers, reading results back, predicting, tearing the fit down - while the loop itself only
    enqueues on its own stream, replays, records and waits on events.  So:
      * a fit thread holds the lock for everything it does on the device EXCEPT its epoch loop (`with DEVICE_LOCK:` around
        the unit in locator._fit_unit; FitLoop.run() lets go of it with `released()` while it loops);
      * a capture (two short events per fit, EpochRunner.start_epoch) takes the lock: no sibling is setting up, reading
        back or tearing down while a capture is open, and no two captures overlap.
    Re-entrant per thread; `released()` drops the thread's hold entirely for the duration of a block."""

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._tls = threading.local()

    def __enter__(self):
        depth = getattr(self._tls, "depth", 0)
        if depth == 0:
            self._lock.acquire()
        self._tls.depth = depth + 1
        return self

    def __exit__(self, *exc):
        self._tls.depth -= 1
        if self._tls.depth == 0:
            self._lock.release()
        return False

    def held(self):
        return getattr(self._tls, "depth", 0) > 0

    def released(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            depth = getattr(self._tls, "depth", 0)
            if depth:
                self._tls.depth = 0
                self._lock.release()
            try:
                yield
            finally:
                if depth:
                    self._lock.acquire()
                    self._tls.depth = depth
        return cm()


DEVICE_LOCK = _DeviceLock()


class Callbacks:
    """ModelCheckpoint(best only) -> EarlyStopping -> ReduceLROnPlateau on val_loss, in that order
    (locator.py:362).  All comparisons are strict '<' with min_delta 0 (locator.py:349-361)."""

    def __init__(self, patience=100, lr0=1e-3, lr_patience=None, lr_factor=0.5):
        self.patience = int(patience)
        self.lr_patience = int(patience / 6) if lr_patience is None else int(lr_patience)     # locator.py:354
        self.lr_factor = float(lr_factor)
        self.lr = float(np.float32(lr0))              # Keras keeps the LR in an fp32 variable
        self.ck_best = np.inf
        self.es_best = np.inf
        self.es_wait = 0
        self.rl_best = np.inf
        self.rl_wait = 0

    def on_epoch_end(self, epoch, val_loss):
        lr_logged = self.lr
        save = val_loss < self.ck_best
        if save:
            self.ck_best = val_loss
        self.es_wait += 1
        if val_loss < self.es_best:
            self.es_best = val_loss
            self.es_wait = 0
        stop = self.es_wait >= self.patience and epoch > 0
        if val_loss < self.rl_best:
            self.rl_best = val_loss
            self.rl_wait = 0
        else:
            self.rl_wait += 1
            if self.rl_wait >= self.lr_patience:
                self.lr = float(np.float32(max(np.float32(self.lr) * np.float32(self.lr_factor), 0.0)))
                self.rl_wait = 0
        return save, stop, lr_logged


class History:
    """Stand-in for keras.callbacks.History: `.history` is a dict of per-epoch lists in Keras 3
    column order loss, val_loss, learning_rate (locator.py:468, :479-482)."""

    def __init__(self):
        self.history = {"loss": [], "val_loss": [], "learning_rate": []}
        self.epoch_seconds = []


class EpochRunner:
    """Enqueues (and optionally graph-captures) one epoch on a LocatorNet."""

    def __init__(self, net: LocatorNet, train_rows, val_rows, batch_size=32, use_graph=True, chain=None, xchain=False,
                 side_stats=False):
        """chain: None = chain consecutive steps where the library supports it (LocatorNet.chain_supported), False =
        one layer-1 forward launch per step (the unchained schedule; tests compare the two).
        xchain: chain ACROSS the epoch boundary as well (needs chained steps and start_epoch(perm, perm_next=...)): the last
        step of epoch e also computes the layer-1 forward of epoch e + 1's first minibatch, whose rows and batch statistics
        are known an epoch early (they depend on X and the next permutation only).  The unchained first forward of every
        epoch but the first disappears; the validation sweep, which sits between the two steps, works in a second workspace
        (loc_net.ws_predict) so that the hand-over survives it."""
        self.slot_rows = net.set_batch(int(batch_size))      # validates 1..128 and the shape constraints
        self.net = net
        dev = net.device
        self.batch = int(batch_size)
        self.train_rows = np.asarray(train_rows, dtype=np.int32)
        self.n_train = len(self.train_rows)
        self.steps = (self.n_train + self.batch - 1) // self.batch
        self.n_val = len(val_rows)
        self.val_rows = torch.as_tensor(np.asarray(val_rows, dtype=np.int32)).to(dev)
        self.perm_ring = [torch.empty(self.steps * self.batch, dtype=torch.int32).pin_memory()]
        self.perm_host = self.perm_ring[0]
        self.epochs_started = 0
        self.cb = None                       # device-side callbacks (enable_device_callbacks)
        self.perm_dev = torch.zeros(self.steps * self.batch, dtype=torch.int32, device=dev)
        # keep flags per step: [rows][Hp], or [32][Kp] when the Dropout layer sits on the BatchNorm output (--nlayers 1)
        self.mask_stride = self.slot_rows * net.mask_width
        self.masks = (torch.zeros(self.steps * self.mask_stride, dtype=torch.uint8, device=dev)
                      if net.drop_p > 0 else None)
        self.stats = torch.zeros(self.steps + max(self.n_val, 1), dtype=torch.float32, device=dev)
        self.stats_host = torch.empty_like(self.stats, device="cpu").pin_memory()
        self.val_yhat = torch.zeros((max(self.n_val, 1), 2), dtype=torch.float32, device=dev)
        # [steps][mean | var][Kp]: BN batch statistics of every minibatch of the epoch (one launch per epoch)
        self.stats_ep = torch.zeros(self.steps * 2 * net.d.Kp, dtype=torch.float32, device=dev)
        self.use_graph = use_graph
        self.graph = None
        self.epochs_run = 0
        self.step_sizes = np.array([min(self.batch, self.n_train - j * self.batch) for j in range(self.steps)])
        net.cnet()
        self.chain = net.chain_supported() if chain is None else (bool(chain) and net.chain_supported())
        self.xchain = bool(xchain) and self.chain
        if self.xchain:
            # permutation and batch statistics exist twice, by epoch parity: epoch e trains from [e % 2] while [1 - e % 2]
            # receives epoch e + 1's; the captured graph differs by parity, so there are two of them
            self.perm_dev2 = [self.perm_dev, torch.zeros_like(self.perm_dev)]
            self.stats_ep2 = [self.stats_ep, torch.zeros_like(self.stats_ep)]
            self.graphs = [None, None]
            self._perm_promised = None       # what the last start_epoch uploaded for the epoch after it
            self.side_stats = bool(side_stats)
            self._side = None
            if len(self.perm_ring) < 3:       # two uploads at the first epoch, one per epoch after: never reuse a pending buffer
                self.perm_ring = [torch.empty(self.steps * self.batch, dtype=torch.int32).pin_memory() for _ in range(3)]
            if getattr(net, "ws_predict", None) is None:
                net.ws_predict = torch.empty_like(net.ws)
            net.cnet()

    def enable_device_callbacks(self, patience, lr0, lr_patience, lr_factor, max_epochs, depth=2):
        """ModelCheckpoint / EarlyStopping / ReduceLROnPlateau evaluated on the device at the end of every epoch
        (loc_epoch_callbacks + loc_snapshot_if, part of the epoch's stream / captured graph), so epochs can be enqueued
        `depth` ahead of the device.  Must be called before the first epoch."""
        import ctypes as C
        from . import _lib
        assert self.epochs_started == 0 and self.graph is None
        net, dev = self.net, self.net.device
        st = _lib.CbState()
        st.ck_best = st.es_best = st.rl_best = float("inf")
        st.lr = float(np.float32(lr0))
        st.lr_factor = float(lr_factor)
        st.patience = int(patience)
        st.lr_patience = int(patience / 6) if lr_patience is None else int(lr_patience)      # locator.py:354
        st.stop_epoch = st.best_epoch = -1
        raw = np.frombuffer(bytes(st), dtype=np.uint8).copy()
        cap = int(max_epochs)
        self.cb = {"state": torch.from_numpy(raw).to(dev), "hist": torch.zeros((cap, 4), dtype=torch.float64, device=dev),
                   "hist_host": torch.zeros((cap, 4), dtype=torch.float64).pin_memory(), "cap": cap, "depth": int(depth),
                   "events": [torch.cuda.Event() for _ in range(int(depth) + 2)], "state_size": C.sizeof(_lib.CbState)}
        if net.best is None:
            net.best = torch.empty_like(net.params)
        net.lr_t.fill_(st.lr)
        self.perm_ring = [torch.empty(self.steps * self.batch, dtype=torch.int32).pin_memory() for _ in range(int(depth) + 3)]
        self.perm_host = self.perm_ring[0]

    def read_cb_state(self):
        """The device's callback state (synchronises): a _lib.CbState."""
        from . import _lib
        raw = self.cb["state"].cpu().numpy().tobytes()
        return _lib.CbState.from_buffer_copy(raw)

    def enqueue(self, ev=None, epoch=0):
        net = self.net
        sz = 2 * net.d.Kp
        n_last = int(self.step_sizes[-1])
        if self.xchain:
            p = epoch & 1
            perm_dev, stats_ep = self.perm_dev2[p], self.stats_ep2[p]
            perm_nx, stats_nx = self.perm_dev2[1 - p], self.stats_ep2[1 - p]
            if epoch == 0:
                net.epoch_bn_stats(perm_dev, self.batch, n_last, self.steps, stats_ep)       # statistics, moving updates, step 0's bn4
            else:
                net.epoch_bn_finish(self.steps, stats_ep)                                      # moving updates only: bn4 is the hand-over
            # the NEXT epoch's batch statistics, an epoch early - on a side stream (forked here, joined before the epoch's last
            # step, which is the first to need them): 22 us of streaming that the device can run beside the first steps'
            # hidden-stack phases, which keep 16 of 256 compute units busy
            if self.side_stats:
                cur = torch.cuda.current_stream()
                if self._side is None:
                    self._side = torch.cuda.Stream(device=net.device)
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    net.epoch_bn_stats_only(perm_nx, self.batch, n_last, self.steps, stats_nx)
            else:
                net.epoch_bn_stats_only(perm_nx, self.batch, n_last, self.steps, stats_nx)
        else:
            perm_dev, stats_ep = self.perm_dev, self.stats_ep
            net.epoch_bn_stats(perm_dev, self.batch, n_last, self.steps, stats_ep)
        for j in range(self.steps):
            nb = int(self.step_sizes[j])
            mask = self.masks[j * self.mask_stride:] if self.masks is not None else None
            e0, e1 = (ev[j] if ev is not None else (None, None))
            nxt = stats_ep[(j + 1) * sz:] if j + 1 < self.steps else None
            if self.chain:
                # the epoch's minibatches are all known: step j's layer-1 backward also computes step j + 1's layer-1
                # forward from the updated weights while they are in registers (one pass over W1 per step)
                last = j + 1 >= self.steps
                rows_next = None if last else perm_dev[(j + 1) * self.batch:]
                nb_next = 0 if last else int(self.step_sizes[j + 1])
                fwd_done = j > 0
                if self.xchain:
                    if last:                     # ... and the epoch's last step computes the next epoch's first forward
                        rows_next, nb_next, nxt = perm_nx, int(self.step_sizes[0]), stats_nx
                        if self.side_stats:
                            torch.cuda.current_stream().wait_stream(self._side)
                    fwd_done = j > 0 or epoch > 0
                net.train_step_chain(perm_dev[j * self.batch:], nb, j + 1, mask, self.stats[j:], nxt, rows_next, nb_next,
                                     fwd_done, e0, e1)
                continue
            net.train_step(perm_dev[j * self.batch:], nb, j + 1, mask, self.stats[j:], e0, e1,
                           bn_ready=True, bn_next=nxt)
        if self.n_val:
            net.predict_rows(self.val_rows, self.n_val, self.val_yhat, self.stats[self.steps:], in_fit=True)
        if self.cb is not None:
            from . import _lib
            from .net import _ptr, _stream
            cb = self.cb
            _lib.check(net.lib.loc_epoch_callbacks(_ptr(self.stats), self.steps, self.batch, int(self.step_sizes[-1]), self.n_val,
                                                   _ptr(cb["state"]), _ptr(net.lr_t), _ptr(cb["hist"]), cb["cap"], _stream()),
                       "loc_epoch_callbacks")
            _lib.check(net.lib.loc_snapshot_if(_ptr(cb["state"]), _ptr(net.params), _ptr(net.best), net.params.numel(),
                                               _stream()), "loc_snapshot_if")
        net.t_base_t.add_(self.steps)

    def _upload_perm(self, perm, dst, slot):
        rows = self.train_rows[np.asarray(perm)]
        # pinned staging buffers rotate: with epochs enqueued ahead of the device the copy of epoch e is still pending
        # when the host prepares epoch e + 1 (the ring is depth + 2 long and FitLoop waits for epoch e - depth first)
        host = self.perm_ring[slot % len(self.perm_ring)]
        host[:self.n_train] = torch.from_numpy(rows)
        host[self.n_train:] = 0
        dst.copy_(host, non_blocking=True)
        self.perm_host = host

    def start_epoch(self, perm, ev=None, perm_next=None):
        """Enqueue one epoch on the CURRENT stream and return without waiting (several fits on their own streams can
        be in flight at once, replicates.py / bench.py --replicates-per-gpu).  perm: permutation of range(n_train)
        (Keras shuffle=True draws it unseeded; here it is an input).  perm_next (xchain): the NEXT epoch's permutation
        (None = there is none: the chained forward for it is computed from this epoch's and never used)."""
        net = self.net
        net.params_changed()                 # a graph replay trains without passing through net.train_step
        e = self.epochs_started
        if self.xchain:
            p = e & 1
            if e == 0:
                self._upload_perm(perm, self.perm_dev2[0], 0)
            elif self._perm_promised is None or not np.array_equal(np.asarray(perm), self._perm_promised):
                # epoch e trains on what the PREVIOUS call uploaded as perm_next (its first forward was computed from it):
                # a caller that did not announce this permutation then (run_epoch, start_epoch without perm_next) would
                # silently train on another one
                raise ValueError("cross-epoch chaining: start_epoch(perm) must receive the permutation the previous call "
                                 "passed as perm_next (use FitLoop, or EpochRunner(xchain=False) for one epoch at a time)")
            self._upload_perm(perm if perm_next is None else perm_next, self.perm_dev2[1 - p], e + 1)
            self._perm_promised = np.array(perm if perm_next is None else perm_next)
        else:
            self._upload_perm(perm, self.perm_dev, e)
        if self.masks is not None:
            net.fill_dropout_masks(self.masks, self.masks.numel(), e * self.masks.numel())
        self.epochs_started += 1
        if self.use_graph and ev is None:
            # epoch 0 runs eagerly = warm-up (with xchain it also differs from the others: it has no hand-over to start
            # from); with xchain so does epoch 1, and the two parities get a graph each
            slot = (e & 1) if self.xchain else 0
            graphs = self.graphs if self.xchain else [self.graph]
            if graphs[slot] is None and e >= (2 if self.xchain else 1):
                g = torch.cuda.CUDAGraph()
                cur = torch.cuda.current_stream()
                # capture is not allowed on the default stream: torch then captures on a side stream of its own
                kw = {} if cur == torch.cuda.default_stream() else {"stream": cur}
                # thread-local capture mode: another fit of this process (its own thread and stream, replicates.py) keeps
                # allocating, synchronising and launching while this one captures
                # ... and no cyclic garbage collection while the capture is open: a collection pass triggered by the ~100
                # small allocations of enqueue() may run the destructor of some unreachable torch object (an older fit's
                # graph, an event), whose HIP call is illegal inside a capture and aborts the process - seen once in a few
                # runs of the GPU suite (`Fatal Python error: Aborted ... Garbage-collecting`).  torch.cuda.graph() itself
                # collects everything collectable before the capture begins.
                # ... and no sibling fit of this process outside its epoch loop while the capture is open (DEVICE_LOCK)
                import gc
                with DEVICE_LOCK:
                    gc_was = gc.isenabled()
                    gc.disable()
                    try:
                        with torch.cuda.graph(g, capture_error_mode=CAPTURE_ERROR_MODE, **kw):
                            self.enqueue(epoch=e)
                    finally:
                        if gc_was:
                            gc.enable()
                graphs[slot] = g
                if not self.xchain:
                    self.graph = g
            if graphs[slot] is not None:
                graphs[slot].replay()
            else:
                self.enqueue(epoch=e)
        else:
            self.enqueue(ev, epoch=e)
        self.stats_host.copy_(self.stats, non_blocking=True)
        self._stream = torch.cuda.current_stream()

    def finish_epoch(self):
        """Wait for the epoch started last and return (loss, val_loss) as Keras logs them."""
        self._stream.synchronize()
        self.epochs_run += 1
        s = self.stats_host.numpy()
        loss = float(np.dot(s[:self.steps].astype(np.float64), self.step_sizes) / self.n_train)
        val = float(s[self.steps:self.steps + self.n_val].astype(np.float64).mean()) if self.n_val else float("nan")
        return loss, val

    def run_epoch(self, perm, ev=None):
        self.start_epoch(perm, ev)
        return self.finish_epoch()


class FitLoop:
    """model.fit's epoch loop with the callbacks on the device (locator.py:365-376): `submit()` enqueues the next epoch
    (permutation upload, dropout masks, the epoch's kernels or its captured graph, the callback kernel, the predicated
    checkpoint copy, and a copy of the epoch's history row to pinned host memory) and returns at once; `collect(lag)`
    waits for the epochs at least `lag` behind the last one submitted and folds their rows into the History.  The host
    therefore runs up to `depth` epochs ahead of the device; what it reads back (loss, val_loss, the LR the epoch trained
    with, what the callbacks did) was decided on the device, so a synchronous run (depth 0) and a pipelined one produce
    the same history, the same best weights and the same predictions bit for bit - the only difference is up to
    `depth` epochs enqueued behind the stop epoch, which change nothing that is kept (frozen state, include/locator_hip.h)."""

    def __init__(self, net, train_rows, val_rows, *, batch_size=32, max_epochs=5000, patience=100, lr_patience=None,
                 lr_factor=0.5, perm_fn=None, use_graph=True, chain=None, depth=2, verbose=0, log=print, xchain=True,
                 side_stats=False):
        if len(val_rows) == 0:
            raise ValueError("fit needs validation rows: checkpoint, early stopping and the LR plateau all monitor val_loss")
        self.net = net
        self.runner = EpochRunner(net, train_rows, val_rows, batch_size, use_graph, chain=chain, xchain=xchain,
                                  side_stats=side_stats)
        self._perms = {}            # permutations drawn ahead (cross-epoch chaining needs epoch e + 1's when e is enqueued)
        self.max_epochs, self.depth = int(max_epochs), max(0, int(depth))
        self.runner.enable_device_callbacks(patience, 1e-3, lr_patience, lr_factor, self.max_epochs, self.depth)
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([net.seed, net.replicate, 0x7065726D])))
        self.perm_fn = perm_fn if perm_fn is not None else (lambda epoch: rng.permutation(self.runner.n_train))
        self.hist = History()
        self.submitted = 0          # epochs enqueued
        self.collected = 0          # epochs whose history row has been read
        self.stop_epoch = None      # index of the epoch at which early stopping fired, once seen
        self.verbose, self.log = verbose, log
        self._t_last = time.perf_counter()

    @property
    def done(self):
        return self.stop_epoch is not None or self.collected >= self.max_epochs

    def submit(self, ev=None):
        """Enqueue the next epoch on the current stream (no-op once max_epochs are enqueued or the stop was seen)."""
        if self.stop_epoch is not None or self.submitted >= self.max_epochs:
            return False
        e, cb = self.submitted, self.runner.cb
        perm = self._perms.pop(e) if e in self._perms else self.perm_fn(e)       # every epoch's drawn once, in epoch order
        perm_next = None
        if self.runner.xchain and e + 1 < self.max_epochs:
            perm_next = self._perms[e + 1] = self.perm_fn(e + 1)
        self.runner.start_epoch(perm, ev, perm_next)
        cb["hist_host"][e].copy_(cb["hist"][e], non_blocking=True)
        cb["events"][e % len(cb["events"])].record(torch.cuda.current_stream())
        self.submitted += 1
        return True

    def collect(self, lag=None):
        """Fold in the history rows of every epoch at least `lag` behind the newest submitted one (default: depth; 0 =
        wait for everything submitted).  Returns True once the fit is over (early stopping seen or max_epochs collected)."""
        lag = self.depth if lag is None else lag
        cb = self.runner.cb
        while self.stop_epoch is None and self.collected < self.submitted - lag:
            e = self.collected
            cb["events"][e % len(cb["events"])].synchronize()
            loss, val, lr_logged, flags = (float(v) for v in cb["hist_host"][e])
            flags = int(flags)
            h = self.hist.history
            h["loss"].append(loss)
            h["val_loss"].append(val)
            h["learning_rate"].append(lr_logged)
            now = time.perf_counter()
            self.hist.epoch_seconds.append(now - self._t_last)
            self._t_last = now
            self.collected += 1
            self.runner.epochs_run = self.collected
            if self.verbose:
                if flags & 4:
                    self.log(f"\nEpoch {e + 1}: ReduceLROnPlateau reducing learning rate to "
                             f"{float(np.float32(lr_logged) * np.float32(self.runner_lr_factor))}.")
                self.log(f"Epoch {e + 1}/{self.max_epochs} - loss: {loss:.4f} - val_loss: {val:.4f} - "
                         f"learning_rate: {lr_logged:.4e}")
            if flags & 2:
                self.stop_epoch = e
        return self.done

    @property
    def runner_lr_factor(self):
        return float(self.runner.read_cb_state().lr_factor) if self.verbose else 0.5

    def finish(self):
        """After the last collect: wait for the device, check that a checkpoint exists, reload the best weights
        (locator.py:379-388).  Returns the History."""
        self.collect(0)
        self.runner._stream = torch.cuda.current_stream()
        self.runner._stream.synchronize()
        st = self.runner.read_cb_state()
        h = self.hist.history
        if st.best_epoch < 0:
            # val_loss was never finite (diverged fit): the reference dies here too, load_weights finds no checkpoint file
            raise RuntimeError(f"training produced no checkpoint: val_loss was never finite in {len(h['loss'])} "
                               f"epochs (last loss {h['loss'][-1]}, last val_loss {h['val_loss'][-1]})")
        self.hist.best_epoch = int(st.best_epoch)
        self.net.restore_best()
        return self.hist

    def run(self):
        # the steady loop only enqueues on this thread's stream, replays graphs, records and waits on events: it runs
        # without the process-wide device lock (a sibling fit may be capturing meanwhile); finish() reads back under it
        with DEVICE_LOCK.released():
            while not self.done:
                if not self.submit():
                    self.collect(0)
                    break
                self.collect()
        with DEVICE_LOCK:
            return self.finish()


def fit(net: LocatorNet, train_rows, val_rows, *, batch_size=32, max_epochs=5000, patience=100, lr_patience=None,
        lr_factor=0.5, perm_fn=None, use_graph=True, verbose=0, log=print, chain=None, pipelined=True, depth=2, xchain=True):
    """train_network (locator.py:365-394): fit with checkpoint / early-stop / LR-plateau callbacks, then
    reload the best weights.  Returns a History.  lr_patience None = int(patience / 6) (locator.py:354).
    pipelined: the host enqueues epochs `depth` ahead of the device (False = wait for every epoch before enqueueing the
    next; same kernels, same decisions - they are taken on the device either way - same results bit for bit)."""
    loop = FitLoop(net, train_rows, val_rows, batch_size=batch_size, max_epochs=max_epochs, patience=patience,
                   lr_patience=lr_patience, lr_factor=lr_factor, perm_fn=perm_fn, use_graph=use_graph, chain=chain,
                   depth=depth if pipelined else 0, verbose=verbose, log=log, xchain=xchain)
    return loop.run()
