import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests.gpu_util import make_problem, build_net
from locator_amd.train import EpochRunner
K = int(sys.argv[1]); width = int(sys.argv[2])
x, y, p, rng = make_problem(116, K, width, 10, seed=3)
tr, va = np.arange(96), np.arange(96, 116)
for chain in (False, True):
    net = build_net(x, y, p, drop_p=0.25, seed=5)
    r = EpochRunner(net, tr, va, 32, use_graph=False, chain=chain)
    for e in range(2):
        l, v = r.run_epoch(np.random.default_rng(7 + e).permutation(96))
        torch.cuda.synchronize()
        print("chain", chain, "epoch", e, "loss", l, "val", v, "steps", r.stats[:3].cpu().numpy(), "grid", net.l1_bwd_grid)
