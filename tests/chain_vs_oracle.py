"""Measurement helper (GPU): chained and unchained schedules of the same 10 steps at 100,000 SNPs, each against the fp64
oracle fit -- which of the two fp32 associations of the gamma / beta / dW1 gradients is closer.
  python tests/chain_vs_oracle.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import locator_oracle as O
from tests.gpu_util import build_net, make_problem, maxerr, params_err
from tests.test_gpu_chain import _run_epochs
K, width, nlayers, n_train = 100000, 256, 10, 160
x, y, p, rng = make_problem(n_train + 20, K, width, nlayers, seed=100)
tr, va = np.arange(n_train), np.arange(n_train, n_train + 20)
perms = [np.random.default_rng(70 + e).permutation(n_train) for e in range(2)]
res = {}
for chain in (False, True):
    net, h, pp, m, v, masks = _run_epochs(x, y, p, tr, va, perms, chain, 0.25, True)
    res[chain] = (pp, masks)
pref = O.copy_params(p)
masks = res[True][1]
O.fit(pref, x[tr], y[tr], x[va], y[va], batch_size=32, max_epochs=2, patience=100, drop_p=0.25,
      perm_fn=lambda e: perms[e], mask_fn=lambda e, s, nb: masks[e][s, :nb, :width])
for name, a, b in (("unchained vs oracle", res[False][0], pref), ("chained vs oracle", res[True][0], pref), ("chained vs unchained", res[True][0], res[False][0])):
    out = []
    for k in ("gamma", "beta"):
        d = np.abs(a[k].astype(np.float64) - b[k]); out.append(f"{k}: max {d.max():.2e} frac>2e-6 {np.mean(d > 2e-6):.2e} mean {d.mean():.2e}")
    d = np.abs(a["W"][0].astype(np.float64) - b["W"][0]); out.append(f"W0: max {d.max():.2e} frac>2e-6 {np.mean(d > 2e-6):.2e} mean {d.mean():.2e}")
    print(name, " | ".join(out))
