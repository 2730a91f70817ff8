"""CPU oracle for locator's genotype -> coordinate regression hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``locator_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported baseline.

PARITY UNPINNED.  The reference (``/root/reference/locator/locator.py``) does
all of its arithmetic inside TensorFlow/Keras, which is an un-vendored,
un-pinned dependency (``setup.py:17`` ``tensorflow>=2.10.0``) that is not
installed in this environment and cannot be installed (no network).  The
reference ships no tests and no golden vectors.  This file therefore restates
the *published* Keras semantics that the reference's call sites imply
(SURVEY.md Appendix A); it has been pinned only against
  * the NumPy-side known answers (split / bootstrap RNG chain),
  * an independent torch-autograd float64 derivation of every gradient
    (tests/test_oracle.py),
not against outputs of the reference itself.

Reference call sites restated here (file:line in /root/reference):
  locator/locator.py:284-292   normalize_locs
  locator/locator.py:295-308   split_train_test
  locator/locator.py:311-327   load_network  (architecture, loss, optimizer)
  locator/locator.py:330-362   load_callbacks (checkpoint / early stop / LR plateau)
  locator/locator.py:365-394   train_network (model.fit semantics, best-weight reload)
  locator/locator.py:397-470   predict_locs  (predict, de-normalise, metrics)
  locator/locator.py:635-653   bootstrap reseed + site_order chain

Everything is plain NumPy.  ``dtype`` selects float64 (the oracle proper) or
float32 (used to size fp32 round-off and as the timed CPU baseline).
"""
from __future__ import annotations

import numpy as np

# Keras defaults implied by locator.py:318 (BatchNormalization()) and :326 (optimizer="Adam")
BN_EPS = 1e-3
BN_MOMENTUM = 0.99
ADAM_B1 = 0.9
ADAM_B2 = 0.999
ADAM_EPS = 1e-7
ADAM_LR0 = 1e-3


# ----------------------------------------------------------------------------
# host-side NumPy pieces of the path (exactly restatable: NumPy only)
# ----------------------------------------------------------------------------
def normalize_locs(locs):
    """locator.py:284-292 — z-score x and y with nanmean / nanstd (ddof=0)."""
    locs = np.asarray(locs, dtype=np.float64)
    meanlong = np.nanmean(locs[:, 0])
    sdlong = np.nanstd(locs[:, 0])
    meanlat = np.nanmean(locs[:, 1])
    sdlat = np.nanstd(locs[:, 1])
    out = np.empty_like(locs)
    out[:, 0] = (locs[:, 0] - meanlong) / sdlong
    out[:, 1] = (locs[:, 1] - meanlat) / sdlat
    return meanlong, sdlong, meanlat, sdlat, out


def split_train_test(ac, locs, train_split=0.9):
    """locator.py:295-308 — consumes exactly one np.random.choice from the
    *global legacy* NumPy stream.  ``ac`` is SNP-major (K x N)."""
    train = np.argwhere(~np.isnan(locs[:, 0]))[:, 0]
    pred = np.array([x for x in range(len(locs)) if x not in set(train.tolist())], dtype=np.int64)
    test = np.random.choice(train, round((1 - train_split) * len(train)), replace=False)
    tset = set(test.tolist())
    train = np.array([x for x in train if x not in tset])
    traingen = np.transpose(ac[:, train])
    testgen = np.transpose(ac[:, test])
    predgen = np.transpose(ac[:, pred]) if len(pred) else np.zeros((0, ac.shape[0]), ac.dtype)
    return train, test, traingen, testgen, locs[train], locs[test], pred, predgen


def bootstrap_chain(nboots, nsites):
    """locator.py:635-650 — per replicate: reseed the global stream with a draw
    from itself, then draw ``site_order``.  Returns [(reseed, site_order), ...]."""
    out = []
    for _ in range(nboots):
        s = np.random.choice(range(int(1e6)), 1)
        np.random.seed(s)
        site_order = np.random.choice(nsites, nsites, replace=True)
        out.append((int(s[0]), site_order))
    return out


# ----------------------------------------------------------------------------
# model state
# ----------------------------------------------------------------------------
def layer_dims(K, width, nlayers):
    """Dense kernel shapes in forward order (locator.py:317-325):
    nlayers ELU layers (first is K x width), then Dense(2), Dense(2)."""
    dims = [(K, width)] + [(width, width)] * (nlayers - 1) + [(width, 2), (2, 2)]
    return dims


def init_params(K, width=256, nlayers=10, rng=None, dtype=np.float64):
    """Keras defaults [K]: glorot_uniform kernels, zero biases, BN gamma=1,
    beta=0, moving_mean=0, moving_var=1."""
    rng = np.random.default_rng(0) if rng is None else rng
    W, b = [], []
    for fi, fo in layer_dims(K, width, nlayers):
        lim = np.sqrt(6.0 / (fi + fo))
        W.append(rng.uniform(-lim, lim, size=(fi, fo)).astype(dtype))
        b.append(np.zeros(fo, dtype))
    return {
        "gamma": np.ones(K, dtype), "beta": np.zeros(K, dtype),
        "mov_mean": np.zeros(K, dtype), "mov_var": np.ones(K, dtype),
        "W": W, "b": b,
    }


def copy_params(p):
    return {k: ([a.copy() for a in v] if isinstance(v, list) else v.copy()) for k, v in p.items()}


def cast_params(p, dtype):
    return {k: ([a.astype(dtype) for a in v] if isinstance(v, list) else v.astype(dtype)) for k, v in p.items()}


def zeros_like_trainable(p):
    return {"gamma": np.zeros_like(p["gamma"]), "beta": np.zeros_like(p["beta"]),
            "W": [np.zeros_like(a) for a in p["W"]], "b": [np.zeros_like(a) for a in p["b"]]}


def n_pre(nlayers):
    """ELU layers before the Dropout (locator.py:319)."""
    return int(np.floor(nlayers / 2))


# ----------------------------------------------------------------------------
# forward / backward
# ----------------------------------------------------------------------------
def elu(z):
    return np.where(z > 0, z, np.expm1(np.minimum(z, 0)))


def forward(p, x, training, drop_mask=None, drop_p=0.25, update_moving=True):
    """One forward pass.

    x          (n, K) genotype counts (any integer/float dtype; cast on entry)
    training   True: batch statistics + dropout; False: moving statistics
    drop_mask  (n, width) array of {0,1} keep flags (training only).  Keras draws
               it from its own unseeded RNG, so the oracle takes it as an input.
               With nlayers == 1 the Dropout layer follows the BatchNormalization
               directly (locator.py:319-323: floor(1/2) = 0 Dense layers before it),
               so the mask is (n, K) and acts on the normalised genotypes.
    Returns (yhat (n,2), cache).  In training mode p's moving stats are updated
    in place (the forward pass owns that update in Keras)."""
    dt = p["gamma"].dtype
    x = np.asarray(x).astype(dt)
    nl = len(p["W"]) - 2
    npre = n_pre(nl)
    eps = dt.type(BN_EPS)
    if training:
        mu = x.mean(axis=0)
        var = ((x - mu) ** 2).mean(axis=0)          # biased
        if update_moving:
            mom = dt.type(BN_MOMENTUM)
            p["mov_mean"] = p["mov_mean"] * mom + mu * (1 - mom)
            p["mov_var"] = p["mov_var"] * mom + var * (1 - mom)
    else:
        mu, var = p["mov_mean"], p["mov_var"]
    rstd = 1.0 / np.sqrt(var + eps)
    xn = (x - mu) * rstd
    xh = xn * p["gamma"] + p["beta"]
    acts_in = []        # input to each dense layer
    acts_out = []       # ELU output of each hidden layer (pre-dropout)
    a = xh
    if npre == 0 and training and drop_p > 0:
        assert drop_mask is not None and drop_mask.shape == xh.shape
        a = a * (drop_mask.astype(dt) * dt.type(1.0 / (1.0 - drop_p)))
    for l in range(nl):
        acts_in.append(a)
        z = a @ p["W"][l] + p["b"][l]
        a = elu(z)
        acts_out.append(a)
        if l == npre - 1 and training and drop_p > 0:
            assert drop_mask is not None
            a = a * (drop_mask.astype(dt) * dt.type(1.0 / (1.0 - drop_p)))
    acts_in.append(a)
    y1 = a @ p["W"][nl] + p["b"][nl]
    acts_in.append(y1)
    y2 = y1 @ p["W"][nl + 1] + p["b"][nl + 1]
    cache = dict(xn=xn, acts_in=acts_in, acts_out=acts_out, drop_mask=drop_mask,
                 drop_p=drop_p, training=training, mu=mu, var=var)
    return y2, cache


def euclid(yhat, y):
    """locator.py:314-315 — per-sample sqrt(sum((ŷ-y)^2)); K.sqrt clamps at 0."""
    return np.sqrt(np.maximum(((yhat - y) ** 2).sum(axis=-1), 0))


def loss_and_grads(p, x, y, drop_mask=None, drop_p=0.25, update_moving=True):
    """Batch-mean Euclidean loss and its gradient w.r.t. every trainable tensor
    (SURVEY.md A.2/A.3).  d_i == 0 gives a zero gradient row (Keras gives NaN;
    the only intentional deviation, also made by the HIP path)."""
    dt = p["gamma"].dtype
    y = np.asarray(y).astype(dt)
    yhat, c = forward(p, x, True, drop_mask, drop_p, update_moving)
    n = x.shape[0]
    diff = yhat - y
    d = euclid(yhat, y)
    loss = d.mean()
    safe = np.where(d > 0, d, 1)
    dy2 = np.where(d[:, None] > 0, diff / safe[:, None], 0) / n
    nl = len(p["W"]) - 2
    npre = n_pre(nl)
    g = zeros_like_trainable(p)
    # Dense(2) #2
    g["W"][nl + 1] = c["acts_in"][nl + 1].T @ dy2
    g["b"][nl + 1] = dy2.sum(0)
    dy1 = dy2 @ p["W"][nl + 1].T
    # Dense(2) #1
    g["W"][nl] = c["acts_in"][nl].T @ dy1
    g["b"][nl] = dy1.sum(0)
    da = dy1 @ p["W"][nl].T
    for l in range(nl - 1, -1, -1):
        if l == npre - 1 and drop_p > 0:
            da = da * (c["drop_mask"].astype(dt) * dt.type(1.0 / (1.0 - drop_p)))
        a = c["acts_out"][l]
        dz = da * np.where(a > 0, 1, a + 1)       # ELU'(z) = 1 or e^z = a+1
        g["W"][l] = c["acts_in"][l].T @ dz
        g["b"][l] = dz.sum(0)
        da = dz @ p["W"][l].T
    if npre == 0 and drop_p > 0:                  # Dropout sits on the BatchNorm output (nlayers == 1)
        da = da * (c["drop_mask"].astype(dt) * dt.type(1.0 / (1.0 - drop_p)))
    # da is now d loss / d x̂  (B x K); BN gamma/beta only (input is data)
    g["gamma"] = (da * c["xn"]).sum(0)
    g["beta"] = da.sum(0)
    return loss, g, yhat


def adam_alpha(lr, t):
    """Keras Adam step size with bias correction folded in."""
    return lr * np.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)


def adam_apply(p, g, m, v, t, lr):
    """In-place Adam (Keras form: eps outside the root, on the un-corrected sqrt(v))."""
    dt = p["gamma"].dtype
    alpha = dt.type(adam_alpha(lr, t))
    c1 = dt.type(1 - ADAM_B1)
    c2 = dt.type(1 - ADAM_B2)
    eps = dt.type(ADAM_EPS)

    def upd(w, gg, mm, vv):
        mm += (gg - mm) * c1
        vv += (gg * gg - vv) * c2
        w -= (mm * alpha) / (np.sqrt(vv) + eps)

    upd(p["gamma"], g["gamma"], m["gamma"], v["gamma"])
    upd(p["beta"], g["beta"], m["beta"], v["beta"])
    for l in range(len(p["W"])):
        upd(p["W"][l], g["W"][l], m["W"][l], v["W"][l])
        upd(p["b"][l], g["b"][l], m["b"][l], v["b"][l])


def train_step(p, m, v, t, lr, x, y, drop_mask=None, drop_p=0.25):
    """forward(training) -> loss -> grads -> Adam.  Returns the
    batch loss measured before the update."""
    loss, g, _ = loss_and_grads(p, x, y, drop_mask, drop_p)
    adam_apply(p, g, m, v, t, lr)
    return loss


def predict(p, x, batch=4096):
    """Inference-mode forward; row blocking does not change results."""
    outs = []
    for i in range(0, x.shape[0], batch):
        outs.append(forward(p, x[i:i + batch], False)[0])
    dt = p["gamma"].dtype
    return np.concatenate(outs, 0) if outs else np.zeros((0, 2), dt)


# ----------------------------------------------------------------------------
# callbacks (locator.py:330-362)
# ----------------------------------------------------------------------------
class Callbacks:
    """ModelCheckpoint(best only) -> EarlyStopping -> ReduceLROnPlateau, all on
    val_loss, evaluated in that order at every epoch end."""

    def __init__(self, patience=100, lr0=ADAM_LR0):
        self.patience = patience
        self.lr_patience = int(patience / 6)
        self.lr = float(np.float32(lr0))   # Keras keeps (and logs) the LR as an fp32 variable
        self.ck_best = np.inf
        self.es_best = np.inf
        self.es_wait = 0
        self.rl_best = np.inf
        self.rl_wait = 0

    def on_epoch_end(self, epoch, val_loss):
        """Returns (save_weights, stop, lr_logged).  Mutates self.lr for the next epoch."""
        lr_logged = self.lr            # the LR this epoch trained with
        save = False
        if val_loss < self.ck_best:
            self.ck_best = val_loss
            save = True
        # EarlyStopping
        self.es_wait += 1
        if val_loss < self.es_best:
            self.es_best = val_loss
            self.es_wait = 0
        stop = self.es_wait >= self.patience and epoch > 0
        # ReduceLROnPlateau (cooldown 0, min_lr 0, factor .5, min_delta 0)
        if val_loss < self.rl_best:
            self.rl_best = val_loss
            self.rl_wait = 0
        else:
            self.rl_wait += 1
            if self.rl_wait >= self.lr_patience:
                # Keras keeps the LR in an fp32 variable
                self.lr = float(np.float32(max(np.float32(self.lr) * np.float32(0.5), 0.0)))
                self.rl_wait = 0
        return save, stop, lr_logged


# ----------------------------------------------------------------------------
# fit (locator.py:365-388)
# ----------------------------------------------------------------------------
def fit(p, traingen, trainlocs, testgen, testlocs, *, batch_size=32, max_epochs=5000,
        patience=100, drop_p=0.25, perm_fn=None, mask_fn=None, lr0=ADAM_LR0,
        m=None, v=None, t0=0):
    """model.fit with the three callbacks, then reload of the best weights.

    perm_fn(epoch) -> permutation of range(n_train)   (Keras: unseeded shuffle)
    mask_fn(epoch, step, n_b) -> (n_b, width) keep mask (Keras: unseeded dropout); (n_b, K) when nlayers == 1
    Returns (history dict, best_params).  ``p`` is left at the last epoch's weights."""
    n = traingen.shape[0]
    width = p["W"][0].shape[1]
    rng = np.random.default_rng(0)
    if perm_fn is None:
        perm_fn = lambda e: rng.permutation(n)
    if mask_fn is None:
        mask_w = width if n_pre(len(p["W"]) - 2) > 0 else p["W"][0].shape[0]
        mask_fn = lambda e, s, nb: (rng.random((nb, mask_w)) >= drop_p)
    m = zeros_like_trainable(p) if m is None else m
    v = zeros_like_trainable(p) if v is None else v
    cb = Callbacks(patience, lr0)
    hist = {"loss": [], "val_loss": [], "learning_rate": []}
    best = copy_params(p)
    t = t0
    dt = p["gamma"].dtype
    for epoch in range(max_epochs):
        perm = np.asarray(perm_fn(epoch))
        lsum, seen = 0.0, 0
        for s, i in enumerate(range(0, n, batch_size)):
            rows = perm[i:i + batch_size]
            nb = len(rows)
            t += 1
            mask = mask_fn(epoch, s, nb) if drop_p > 0 else None
            loss = train_step(p, m, v, t, dt.type(cb.lr), traingen[rows], trainlocs[rows], mask, drop_p)
            lsum += float(loss) * nb
            seen += nb
        val = float(euclid(predict(p, testgen), np.asarray(testlocs).astype(dt)).mean())
        save, stop, lr_logged = cb.on_epoch_end(epoch, val)
        hist["loss"].append(lsum / seen)
        hist["val_loss"].append(val)
        hist["learning_rate"].append(lr_logged)
        if save:
            best = copy_params(p)
        if stop:
            break
    return hist, best


# ----------------------------------------------------------------------------
# predict_locs metrics (locator.py:414-453)
# ----------------------------------------------------------------------------
def denormalize(pred, sdlong, meanlong, sdlat, meanlat):
    pred = np.asarray(pred, dtype=np.float64)
    return np.stack([pred[:, 0] * sdlong + meanlong, pred[:, 1] * sdlat + meanlat], axis=1)


def validation_metrics(p2, testlocs2):
    """R^2 per axis (squared Pearson), mean / median Euclidean error, original units."""
    r2_long = np.corrcoef(p2[:, 0], testlocs2[:, 0])[0][1] ** 2
    r2_lat = np.corrcoef(p2[:, 1], testlocs2[:, 1])[0][1] ** 2
    dists = np.sqrt(((p2 - testlocs2) ** 2).sum(1))
    return r2_long, r2_lat, float(np.mean(dists)), float(np.median(dists)), dists
