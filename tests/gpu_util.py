"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch

from oracle import locator_oracle as O


def make_problem(n, K, width, nlayers, seed=0, n_na=0):
    rng = np.random.default_rng(seed)
    af = rng.beta(0.4, 0.9, K).clip(0.02, 0.98)
    x = rng.binomial(2, af, (n, K)).astype(np.uint8)
    w = rng.normal(0, 1, (K, 2)) / np.sqrt(K)
    y = (x - x.mean(0)) @ w
    y = (y - y.mean(0)) / y.std(0)
    p = randomize_params(O.init_params(K, width, nlayers, rng), rng, round_fp32=False)
    return x, y, p, rng


def randomize_params(p, rng, round_fp32=True):
    """Make every tensor non-trivial so a dropped term cannot hide.  round_fp32: keep the fp64 oracle's starting
    point exactly representable in fp32, so both sides start from identical values."""
    K = p["gamma"].shape[0]
    p["gamma"] = rng.uniform(0.7, 1.3, K)
    p["beta"] = rng.normal(0, 0.05, K)
    p["mov_mean"] = rng.uniform(0, 1, K)
    p["mov_var"] = rng.uniform(0.2, 1.2, K)
    for l in range(len(p["b"])):
        p["b"][l] = rng.normal(0, 0.05, p["b"][l].shape)
    if round_fp32:
        p = O.cast_params(O.cast_params(p, np.float32), np.float64)
    return p


def build_net(x, y, p, drop_p=0.25, seed=1, **net_kw):
    from locator_amd.net import LocatorNet, upload_genotypes
    K = x.shape[1]
    width = p["W"][0].shape[1]
    nlayers = len(p["W"]) - 2
    X = upload_genotypes(x)
    Y = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)).cuda()
    net = LocatorNet(X, Y, K, width, nlayers, drop_p, seed=seed, **net_kw)
    net.import_params(O.cast_params(p, np.float32))
    return net


def maxerr(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


def params_err(pa, pb):
    out = {}
    for k in ("gamma", "beta", "mov_mean", "mov_var"):
        if k in pa and k in pb:
            out[k] = maxerr(pa[k], pb[k])
    for l in range(len(pa["W"])):
        out[f"W{l}"] = maxerr(pa["W"][l], pb["W"][l])
        out[f"b{l}"] = maxerr(pa["b"][l], pb["b"][l])
    return out
