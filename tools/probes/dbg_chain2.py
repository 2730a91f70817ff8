
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.gpu_util import make_problem, build_net, maxerr
from tests.test_gpu_chain import _sync
def run(K, n_b, n_b_next):
    """loc_l1_backward_adam_chain through the C ABI against loc_l1_backward_adam followed by loc_l1_forward on the next
    minibatch, same inputs: W1 / m / v, b1, gamma / beta and their moments, the next step's [scale|shift|mean|rstd], and
    the next minibatch's layer-1 activations.  dZ1 and the batch statistics are synthetic (the kernels do not care)."""
    import ctypes as C
    from locator_amd import _lib
    x, y, p, rng = make_problem(80, K, 256, 2, seed=K % 89 + n_b)
    net = build_net(x, y, p, drop_p=0.0)
    lib, d, lay = net.lib, net.d, net.lay
    Kp, Hp = d.Kp, d.Hp
    dev = "cuda"
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = torch.from_numpy(rng.choice(80, 32, replace=False).astype(np.int32)).to(dev)
    rows_next = torch.from_numpy(rng.choice(80, 32, replace=False).astype(np.int32)).to(dev)
    dz = torch.zeros((32, Hp), device=dev)
    dz[:n_b] = torch.from_numpy(rng.normal(0, 0.05, (n_b, Hp)).astype(np.float32)).to(dev)      # rows beyond n_b are zero
    # non-trivial Adam state and a third step, so that every term of the update is exercised
    net.adam_m.copy_(torch.from_numpy(rng.normal(0, 1e-3, net.adam_m.numel()).astype(np.float32)))
    # (second moments of the size of the squared first moments, as in a real fit: with v far below m^2 one Adam step
    # amplifies the round-off of the gradient a thousandfold and the comparison would measure that)
    net.adam_v.copy_(torch.from_numpy(((0.5 + rng.random(net.adam_v.numel())) * 1e-6).astype(np.float32)))
    pad = torch.arange(Kp, device=dev) >= K                       # padded SNPs carry zero state, as after init / import
    for buf in (net.adam_m, net.adam_v):
        buf[lay.gamma:lay.gamma + Kp][pad] = 0
        buf[lay.beta:lay.beta + Kp][pad] = 0
        buf[lay.w1:lay.w1 + Hp * Kp].view(-1)[:] = buf[lay.w1:lay.w1 + Hp * Kp]  # (W1S padding rows stay as drawn: zero gradient either way)
    net.t_base_t.fill_(2)

    def stats_of(r, n):       # [mean | biased var] of a minibatch, [2][Kp]
        xb = x[r.cpu().numpy()[:n]].astype(np.float64)
        s = np.zeros((2, Kp), np.float32)
        s[0, :K], s[1, :K] = xb.mean(0), xb.var(0)
        return torch.from_numpy(s).to(dev)

    next_stats = stats_of(rows_next, n_b_next).contiguous()
    cur = stats_of(rows, n_b)
    P0, M0, V0 = net.params.clone(), net.adam_m.clone(), net.adam_v.clone()

    def bn4_now(params):
        g, b = params[lay.gamma:lay.gamma + Kp], params[lay.beta:lay.beta + Kp]
        rstd = torch.where(pad, torch.zeros_like(cur[1]), 1.0 / torch.sqrt(cur[1] + 1e-3))
        sc = g * rstd
        return torch.cat([sc, b - cur[0] * sc, cur[0], rstd]).contiguous()

    grid = max(1, min(net.l1_bwd_grid // 2, Kp // 32))
    out = {}
    for which in ("pair", "chain"):
        net.params.copy_(P0); net.adam_m.copy_(M0); net.adam_v.copy_(V0)
        Pp, Mp, Vp = net.params.data_ptr(), net.adam_m.data_ptr(), net.adam_v.data_ptr()
        o = lambda base, off: C.c_void_p(base + 4 * off)
        bn4 = bn4_now(net.params)
        partial = torch.zeros(512 * 32 * Hp, device=dev)
        a1 = torch.zeros((32, Hp), device=dev)
        common = [o(Pp, lay.w1), o(Mp, lay.w1), o(Vp, lay.w1), o(Pp, lay.gamma), o(Pp, lay.beta), o(Mp, lay.gamma),
                  o(Vp, lay.gamma), o(Mp, lay.beta), o(Vp, lay.beta), o(Pp, lay.b1), o(Mp, lay.b1), o(Vp, lay.b1)]
        tail = [C.c_void_p(net.alpha_tab.data_ptr()), len(net.alpha_tab), C.c_void_p(net.lr_t.data_ptr()),
                C.c_void_p(net.t_base_t.data_ptr()), 1]
        if which == "pair":
            gbs = torch.zeros(4 * Kp + Hp, device=dev)
            _lib.check(lib.loc_l1_backward_adam(C.c_void_p(net.X.data_ptr()), net.X.stride(0), C.c_void_p(rows.data_ptr()),
                                                n_b, C.byref(d), C.c_void_p(bn4.data_ptr()), C.c_void_p(dz.data_ptr()),
                                                *common, C.c_void_p(gbs.data_ptr()), *tail, net.l1_bwd_grid,
                                                C.c_void_p(next_stats.data_ptr()), C.c_void_p(bn4.data_ptr()), None,
                                                C.byref(net.tuning), st), "backward")
            _lib.check(lib.loc_l1_forward(C.c_void_p(net.X.data_ptr()), net.X.stride(0), C.c_void_p(rows_next.data_ptr()),
                                          n_b_next, C.byref(d), C.c_void_p(bn4.data_ptr()), o(Pp, lay.w1), o(Pp, lay.b1),
                                          C.c_void_p(partial.data_ptr()), net.l1_fwd_grid, C.c_void_p(a1.data_ptr()),
                                          None, None, C.c_float(1.0), st), "forward")
        else:
            _lib.check(lib.loc_l1_backward_adam_chain(C.c_void_p(net.X.data_ptr()), net.X.stride(0),
                                                      C.c_void_p(rows.data_ptr()), n_b, C.c_void_p(rows_next.data_ptr()),
                                                      n_b_next, C.byref(d), C.c_void_p(bn4.data_ptr()),
                                                      C.c_void_p(next_stats.data_ptr()), C.c_void_p(dz.data_ptr()), *common,
                                                      *tail, grid, C.c_void_p(partial.data_ptr()), partial.numel(),
                                                      C.byref(net.tuning), st), "chain")
            z = partial[:grid * 32 * Hp].view(grid, 32, Hp).sum(0) + net.params[lay.b1:lay.b1 + Hp]
            a1 = torch.where(z > 0, z, torch.expm1(z))
        _sync()
        out[which] = dict(P=net.params.cpu().numpy().copy(), M=net.adam_m.cpu().numpy().copy(),
                          V=net.adam_v.cpu().numpy().copy(), bn4=bn4.cpu().numpy().copy(), a1=a1.cpu().numpy().copy())

    a, b = out["pair"], out["chain"]
    Hp, Kp = d.Hp, d.Kp
    for nm in ("P", "M", "V"):
        da = np.abs(a[nm][lay.w1:lay.w1 + Hp * Kp] - b[nm][lay.w1:lay.w1 + Hp * Kp])
        bad = np.flatnonzero(da > 1e-5)
        print(nm, "max", da.max(), "n_bad", len(bad))
        if len(bad):
            unit = bad // 1024           # (kt * nht + ht) * 4 + q  blocks of 256 floats -> unit of 1024 floats = (kt, ht)
            kt, ht = unit // (Hp // 32), unit % (Hp // 32)
            print("   kt range", kt.min(), kt.max(), "distinct kt", len(np.unique(kt)), "first kts", np.unique(kt)[:12], "ht", np.unique(ht))
            inner = bad % 1024
            print("   inner q (0..3)", np.unique(inner // 256), "lanes", np.unique((inner % 256) // 4)[:10], "...")
    for nm, off in (("gamma", lay.gamma), ("beta", lay.beta)):
        da = np.abs(a["P"][off:off + Kp] - b["P"][off:off + Kp]); print(nm, "max", da.max(), "n_bad", (da > 1e-5).sum(), np.flatnonzero(da > 1e-5)[:8] // 32)
    print("a1", np.abs(a["a1"] - b["a1"]).max())
run(int(sys.argv[1]), 32, 32)
