import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import locator_oracle as O
from tests.gpu_util import build_net, maxerr, params_err, randomize_params
from locator_amd.synth import split_indices, synth_genotypes, normalize_locs
from locator_amd.train import EpochRunner
x, locs = synth_genotypes(1000, 100_000, seed=20260101, n_na=100)
train, test, pred = split_indices(locs, 0.9, seed=12345)
y = np.nan_to_num(normalize_locs(locs)[4])
rng = np.random.default_rng(2)
p = randomize_params(O.init_params(100_000, 256, 10, rng), rng)
net = build_net(x, y, p, drop_p=0.25, seed=17)
runner = EpochRunner(net, train, test, 32, use_graph=True)
NE = 2
perms = [np.random.default_rng(50 + e).permutation(len(train)) for e in range(NE)]
masks, hist = [], {"loss": [], "val_loss": []}
for e in range(NE):
    l, vl = runner.run_epoch(perms[e]); masks.append(runner.masks.cpu().numpy().reshape(runner.steps, 32, 256).copy())
    hist["loss"].append(l); hist["val_loss"].append(vl)
kw = dict(batch_size=32, max_epochs=NE, patience=100, drop_p=0.25, perm_fn=lambda e: perms[e], mask_fn=lambda e, s, nb: masks[e][s, :nb, :256])
pref = O.copy_params(p); href, _ = O.fit(pref, x[train], y[train], x[test], y[test], **kw)
p32 = O.cast_params(p, np.float32); h32, _ = O.fit(p32, x[train], y[train].astype(np.float32), x[test], y[test].astype(np.float32), **kw)
print("hist hip", hist); print("hist f64", href["loss"], href["val_loss"]); print("hist f32", h32["loss"], h32["val_loss"])
got = net.export_params()
print("errs hip-f64", params_err(got, pref)); print("errs f32-f64", params_err(p32, pref))
yh = torch.zeros((len(pred), 2), device="cuda"); net.predict_rows(torch.from_numpy(pred.astype(np.int32)).cuda(), len(pred), yh); torch.cuda.synchronize()
yh = yh.cpu().numpy(); r64 = O.predict(pref, x[pred]); r32 = O.predict(p32, x[pred])
f = lambda a, b: (np.abs(a - b) / np.maximum(np.abs(b), 1.0))
print("pred rel hip-f64 max/mean", f(yh, r64).max(), f(yh, r64).mean(0))
print("pred rel f32-f64 max/mean", f(r32, r64).max(), f(r32, r64).mean(0))
print("pred rel hip-f32 max", f(yh, r32).max())
print("signed mean diff hip-f64", (yh - r64).mean(0), "f32-f64", (r32 - r64).mean(0))
# predictions with hip weights through the f64 oracle forward
g64 = O.cast_params(got, np.float64); print("pred oracle(hip weights) vs hip", f(O.predict(g64, x[pred]), yh).max())
