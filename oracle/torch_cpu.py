"""Multi-threaded CPU restatement of locator's training epoch on PyTorch-CPU fp32 — the timed `cpu_baseline`.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as oracle/locator_oracle.py: nothing under locator_amd/ may
import it).  BASELINE.md §3: TensorFlow cannot run here or on the GPU box, so the number reported beside the GPU
figure is this restatement of the reference's Keras semantics, labelled "restated reference on CPU (torch), not
TensorFlow".  It follows oracle/locator_oracle.py statement for statement (which cites the reference lines:
load_network /root/reference/locator/locator.py:311-327, model.fit :367-376) and is pinned to it by
tests/test_oracle.py::test_torch_cpu_restatement_matches_the_numpy_oracle.

Why not the NumPy oracle for timing: its Adam is single-threaded elementwise NumPy over 26 M weights x 3 arrays
(1.1 s of a 1.4 s step), which says nothing about what a multi-threaded framework does on the same cores.  Here
the contractions are torch.mm (MKL / OpenMP over all requested threads) and Adam is the Keras form written with
in-place torch._foreach ops (multi-threaded, 28 bytes of traffic per weight).  Gradients are written out by hand
(no autograd tape) exactly as in the oracle.
"""
from __future__ import annotations

import numpy as np
import torch

BN_EPS, BN_MOM = 1e-3, 0.99
B1, B2, ADAM_EPS = 0.9, 0.999, 1e-7


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    import os
    return max(1, (os.cpu_count() or 2) // 2)


class TorchCpuLocator:
    """BN(input) -> L x Dense(H, elu) with Dropout after floor(L/2) -> Dense(2) -> Dense(2); Keras-form Adam."""

    def __init__(self, params, drop_p=0.25, fused=None):
        """params: oracle-format dict (oracle.init_params / cast_params), copied to fp32 tensors.
        fused: single-pass Adam through torch._fused_adam_ (default: when the op exists)."""
        self.fused = hasattr(torch, "_fused_adam_") if fused is None else bool(fused)
        self.step_t = None
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).clone()
        self.W = [f(a) for a in params["W"]]
        self.b = [f(a) for a in params["b"]]
        self.gamma, self.beta = f(params["gamma"]), f(params["beta"])
        self.mov_mean, self.mov_var = f(params["mov_mean"]), f(params["mov_var"])
        self.nl = len(self.W) - 2
        self.npre = self.nl // 2
        self.drop_p = float(drop_p)
        self.train_params = [self.gamma, self.beta] + self.W + self.b
        self.m = [torch.zeros_like(p) for p in self.train_params]
        self.v = [torch.zeros_like(p) for p in self.train_params]
        self.t = 0
        self.step_t = [torch.zeros((), dtype=torch.float32) for _ in self.train_params]

    # ------------------------------------------------------------------ forward pieces
    @staticmethod
    def _elu(z):
        return torch.where(z > 0, z, torch.expm1(torch.clamp(z, max=0)))

    def predict(self, x_u8):
        xh = (x_u8.float() - self.mov_mean) * (torch.rsqrt(self.mov_var + BN_EPS) * self.gamma) + self.beta
        a = xh
        for l in range(self.nl):
            a = self._elu(torch.addmm(self.b[l], a, self.W[l]))
        y1 = torch.addmm(self.b[self.nl], a, self.W[self.nl])
        return torch.addmm(self.b[self.nl + 1], y1, self.W[self.nl + 1])

    def train_step(self, x_u8, y, keep_mask, lr):
        """One Keras train step on a batch; returns the batch loss measured before the update."""
        n = x_u8.shape[0]
        x = x_u8.float()
        mu = x.mean(0)
        var = x.var(0, unbiased=False)
        self.mov_mean.mul_(BN_MOM).add_(mu, alpha=1 - BN_MOM)
        self.mov_var.mul_(BN_MOM).add_(var, alpha=1 - BN_MOM)
        xn = (x - mu) * torch.rsqrt(var + BN_EPS)
        a = xn * self.gamma + self.beta
        ins, outs = [], []
        scale = 1.0 / (1.0 - self.drop_p) if self.drop_p > 0 else 1.0
        for l in range(self.nl):
            ins.append(a)
            a = self._elu(torch.addmm(self.b[l], a, self.W[l]))
            outs.append(a)
            if l == self.npre - 1 and self.drop_p > 0:
                a = a * (keep_mask * scale)
        ins.append(a)
        y1 = torch.addmm(self.b[self.nl], a, self.W[self.nl])
        ins.append(y1)
        y2 = torch.addmm(self.b[self.nl + 1], y1, self.W[self.nl + 1])
        diff = y2 - y
        d = torch.sqrt(torch.clamp((diff * diff).sum(1), min=0))
        loss = float(d.mean())
        safe = torch.where(d > 0, d, torch.ones_like(d))
        dy2 = torch.where(d[:, None] > 0, diff / safe[:, None], torch.zeros_like(diff)) / n
        gW, gb = [None] * (self.nl + 2), [None] * (self.nl + 2)
        gW[self.nl + 1] = ins[self.nl + 1].t().mm(dy2)
        gb[self.nl + 1] = dy2.sum(0)
        dy1 = dy2.mm(self.W[self.nl + 1].t())
        gW[self.nl] = ins[self.nl].t().mm(dy1)
        gb[self.nl] = dy1.sum(0)
        da = dy1.mm(self.W[self.nl].t())
        for l in range(self.nl - 1, -1, -1):
            if l == self.npre - 1 and self.drop_p > 0:
                da = da * (keep_mask * scale)
            ao = outs[l]
            dz = da * torch.where(ao > 0, torch.ones_like(ao), ao + 1)
            gW[l] = ins[l].t().mm(dz)
            gb[l] = dz.sum(0)
            da = dz.mm(self.W[l].t())
        grads = [(da * xn).sum(0), da.sum(0)] + gW + gb
        # Keras Adam: eps outside the root, bias correction folded into the step size:
        #     w -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
        # torch's fused CPU Adam computes  w -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps_t); with
        # eps_t = eps / sqrt(1-b2^t) that is the same number in ONE pass over (w, g, m, v) instead of eight
        self.t += 1
        if self.fused:
            torch._foreach_zero_(self.step_t)           # the op reads the step count from per-parameter tensors
            torch._foreach_add_(self.step_t, float(self.t))
            torch._fused_adam_(self.train_params, grads, self.m, self.v, [], self.step_t,
                               lr=float(lr), beta1=B1, beta2=B2, weight_decay=0.0,
                               eps=ADAM_EPS / float(np.sqrt(1.0 - B2 ** self.t)), amsgrad=False, maximize=False)
        else:
            alpha = float(np.float32(lr * np.sqrt(1.0 - B2 ** self.t) / (1.0 - B1 ** self.t)))
            torch._foreach_lerp_(self.m, grads, 1 - B1)
            torch._foreach_mul_(grads, grads)
            torch._foreach_lerp_(self.v, grads, 1 - B2)
            den = torch._foreach_sqrt(self.v)
            torch._foreach_add_(den, ADAM_EPS)
            torch._foreach_reciprocal_(den)
            torch._foreach_mul_(den, self.m)
            torch._foreach_add_(self.train_params, den, alpha=-alpha)
        return loss

    def fit_epoch(self, x_train, y_train, x_val, y_val, perm, masks, lr, batch=32):
        """One epoch of model.fit (A.4): shuffled minibatches with the partial last batch kept, then the validation
        sweep in inference mode.  Returns (loss, val_loss)."""
        n = x_train.shape[0]
        lsum = 0.0
        for s, i in enumerate(range(0, n, batch)):
            rows = perm[i:i + batch]
            nb = len(rows)
            lsum += self.train_step(x_train[rows], y_train[rows], masks[s][:nb] if masks is not None else None, lr) * nb
        yv = self.predict(x_val)
        dv = torch.sqrt(((yv - y_val) ** 2).sum(1))
        return lsum / n, float(dv.mean())

    def export(self):
        g = lambda t: t.numpy().copy()
        return {"gamma": g(self.gamma), "beta": g(self.beta), "mov_mean": g(self.mov_mean), "mov_var": g(self.mov_var),
                "W": [g(a) for a in self.W], "b": [g(a) for a in self.b]}


def _pick_threads(net, xt, yt, H, drop_p, candidates):
    """Fastest torch thread count for this step shape on this host: more threads are not always faster (M = 32
    contractions and 100k-long elementwise ops parallelise poorly across sockets).  Two timed steps per candidate."""
    import time
    best, best_t = candidates[0], float("inf")
    rows = np.arange(32)
    mask = torch.ones((32, H))
    for c in candidates:
        torch.set_num_threads(c)
        net.train_step(xt[rows], yt[rows], mask if drop_p > 0 else None, 1e-3)
        t0 = time.perf_counter()
        for _ in range(2):
            net.train_step(xt[rows], yt[rows], mask if drop_p > 0 else None, 1e-3)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = c, dt
    return best


def time_epochs(x_u8, y, train, val, K, H, L=10, drop_p=0.25, seconds=12.0, threads=None, seed=0):
    """Times whole epochs (26 steps of 32 + validation sweep on the 1000 x 100k workload) for about `seconds`.
    threads=None: the fastest of {physical cores, 1/2, 1/4, 1/8 of them} on a two-step probe.  Returns a dict for
    bench.py's cpu_baseline; `cores` = the threads actually used."""
    import time

    from . import locator_oracle as O
    rng = np.random.default_rng(seed)
    net = TorchCpuLocator(O.init_params(K, H, L, rng, dtype=np.float32), drop_p)
    xt = torch.from_numpy(np.ascontiguousarray(x_u8[train]))
    xv = torch.from_numpy(np.ascontiguousarray(x_u8[val]))
    yt = torch.from_numpy(np.ascontiguousarray(y[train], dtype=np.float32))
    yv = torch.from_numpy(np.ascontiguousarray(y[val], dtype=np.float32))
    n = len(train)
    steps = (n + 31) // 32
    phys = physical_cores()
    if threads is None:
        cands = sorted({max(1, phys // d) for d in (1, 2, 4, 8)}, reverse=True)
        threads = _pick_threads(net, xt, yt, H, drop_p, cands)
        probe = f"fastest of {cands} threads on a two-step probe ({phys} physical cores)"
    else:
        probe = f"{phys} physical cores"
    threads = int(threads)
    torch.set_num_threads(threads)
    epochs, t_used = 0, 0.0
    t0 = time.perf_counter()
    while True:
        perm = rng.permutation(n)
        masks = [torch.from_numpy((rng.random((32, H)) >= drop_p).astype(np.float32)) for _ in range(steps)]
        loss, val_loss = net.fit_epoch(xt, yt, xv, yv, perm, masks, 1e-3)
        epochs += 1
        t_used = time.perf_counter() - t0
        if t_used >= seconds:
            break
    return {"value": epochs * n / t_used, "unit": "samples/s", "cores": threads, "kind": "port",
            "impl": "restated reference on CPU (torch fp32, hand-written Keras-form Adam), not TensorFlow",
            "sample": f"{epochs} epochs of {n} training rows x {K} SNPs ({steps} minibatch steps of 32 each, partial "
                      f"last batch kept) + the {len(val)}-row validation sweep per epoch, {t_used:.1f} s, "
                      f"torch.set_num_threads({threads}): {probe}",
            "last_loss": round(loss, 5), "last_val_loss": round(val_loss, 5)}
