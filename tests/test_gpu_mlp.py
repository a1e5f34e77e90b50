"""GPU parity of the weight-shared MLP path (forward, backward, ADAM, Polyak, DDPG update)
against the fp64/fp32 oracle, through the C ABI.  Tolerances: fp32 forward
<= 1e-5 rel, gradients <= 1e-4 rel vs the fp64 restatement; fp64 <= 1e-11 rel."""
import zlib

import numpy as np
import pytest

from util import to_dev

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def relerr(a, b):
    return np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max() / max(1e-30, np.abs(b).max())


def make_net(pkg, rng, dims, acts, dtype, max_cols):
    from oracle import nn
    P = nn.glorot_uniform(rng, dims, np.float64)
    for i in range(1, len(P), 2):
        P[i] = rng.standard_normal(P[i].shape) * 0.1
    code = {nn.RELU: "relu", nn.TANH: "tanh", nn.IDENT: None}
    net = pkg.HipMLP(dims, [code[a] for a in acts], P, dtype=dtype, max_cols=max_cols)
    npdt = np.float64 if dtype == torch.float64 else np.float32
    return net, [p.astype(npdt) for p in P]


SHAPES = [
    ("ks_actor_2l", 1, 1, 0.6, True, True),       # Dense(1,6,relu)->Dense(6,1,tanh)   KSSetup.jl:40-46
    ("ks_critic_2l", 1, 1, 7.0, False, True),     # 2->140->1
    ("c2_actor_3l", 3, 1, 1.6, True, False),      # 3->16->16->1
    ("c2_critic_3l", 3, 1, 7.0, False, False),    # 4->140->140->1
    ("kseg_critic", 12, 1, 17.0, False, True),    # 13->340->1
    ("global_actor", 8, 8, 4.8, True, True),      # 8->48->8
]


@pytest.mark.parametrize("name,ns,na,scale,is_actor,drop", SHAPES)
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_forward_backward(pkg, name, ns, na, scale, is_actor, drop, prec):
    from oracle import nn
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)      # (str hashes are salted per process: not a seed)
    dims, acts = nn.layer_sizes(ns, na, scale, is_actor, drop)
    dtype = torch.float64 if prec == "f64" else torch.float32
    cols = 777
    net, P = make_net(pkg, rng, dims, acts, dtype, cols)
    P64 = [p.astype(np.float64) for p in P]
    x = rng.standard_normal((dims[0], cols))
    dy = rng.standard_normal((dims[-1], cols))
    y, zs, as_ = nn.forward(P64, acts, x, keep=True)
    g, dx = nn.backward(P64, acts, zs, as_, dy)
    xd = to_dev(x.T, dtype)
    yd = net(xd).cpu().numpy().T
    gg, dxd = net.backward(xd, to_dev(dy.T, dtype))
    tol_f, tol_g = (1e-11, 1e-10) if prec == "f64" else (1e-5, 1e-4)
    assert relerr(yd, y) <= tol_f
    assert relerr(dxd.cpu().numpy().T, dx) <= tol_g
    for a, b in zip(gg, g):
        assert relerr(a, b) <= tol_g, name
    # params round-trip (checkpoint / copyto!)
    for a, b in zip(net.params(), P):
        assert np.array_equal(a, b)


def test_adam_polyak_match_flux_semantics(pkg):
    from oracle import nn
    rng = np.random.default_rng(3)
    dims, acts = nn.layer_sizes(3, 1, 1.6, True, False)
    net, P = make_net(pkg, rng, dims, acts, torch.float32, 64)
    tgt, PT = make_net(pkg, rng, dims, acts, torch.float32, 64)
    opt = nn.Adam(P, 5e-4)
    app = pkg.CustomNeuralNetworkApproximator(net, pkg.ADAM(5e-4))
    x = rng.standard_normal((3, 64)).astype(np.float32)
    dy = rng.standard_normal((1, 64)).astype(np.float32)
    for it in range(3):
        y, zs, as_ = nn.forward(P, acts, x, keep=True)
        g, _ = nn.backward(P, acts, zs, as_, dy)
        P = opt.step(P, [gi.astype(np.float32) for gi in g])
        net.backward(to_dev(x.T, torch.float32), to_dev(dy.T, torch.float32), want_dx=False)
        app.update()
    for a, b in zip(net.params(), P):
        assert relerr(a, b) <= 2e-5
    pkg._lib.check(net.lib.pdec_polyak(tgt.handle, net.handle, 0.995))
    PT = nn.polyak(PT, P, np.float32(0.995))
    for a, b in zip(tgt.params(), PT):
        assert relerr(a, b) <= 2e-6


@pytest.mark.parametrize("quirk", [1, 0])
@pytest.mark.parametrize("prec,Bu", [("f64", 300), ("f32", 4096), ("f32", 1000), ("f32", 77), ("f32", 40000)])
def test_ddpg_update_matches_oracle(pkg, quirk, prec, Bu):
    """Whole update (src/PDEagent.jl:363-418): losses, all four networks after one and two
    updates.  quirk=1 is the reference's (1xBu).+(Bu) reward broadcast."""
    import ctypes as C
    from oracle import nn
    rng = np.random.default_rng(11 + quirk)
    ns, na = 3, 1
    da, aa = nn.layer_sizes(ns, na, 1.6, True, False)
    dc, ac = nn.layer_sizes(ns, na, 7.0, False, False)
    dtype = torch.float64 if prec == "f64" else torch.float32
    npdt = np.float64 if prec == "f64" else np.float32
    A, PA = make_net(pkg, rng, da, aa, dtype, Bu)
    Cn, PC = make_net(pkg, rng, dc, ac, dtype, Bu)
    At, PAt = make_net(pkg, rng, da, aa, dtype, Bu)
    Ct, PCt = make_net(pkg, rng, dc, ac, dtype, Bu)
    optA, optC = nn.Adam(PA, 5e-4), nn.Adam(PC, 1e-3)
    lib = A.lib
    tol = 1e-9 if prec == "f64" else 2e-4
    for it in range(2):
        s = rng.standard_normal((ns, Bu)).astype(npdt)
        sn = rng.standard_normal((ns, Bu)).astype(npdt)
        a = rng.uniform(-1, 1, (na, Bu)).astype(npdt)
        r = -rng.uniform(0, 1, Bu).astype(npdt)
        t = (rng.uniform(0, 1, Bu) < 0.1).astype(npdt)
        out = nn.ddpg_update(PA, PC, PAt, PCt, optA, optC, aa, ac, s, a, r, t, sn, npdt(np.float32(0.99)), np.float32(0.995), bool(quirk))
        al, cl = C.c_double(), C.c_double()
        ds, da_, dr, dt_, dsn = (to_dev(s.T, dtype), to_dev(a.T, dtype), to_dev(r, dtype), to_dev(t, dtype),
                                 to_dev(sn.T, dtype))   # keep alive across the call
        pkg._lib.check(lib.pdec_ddpg_update(
            A.handle, Cn.handle, At.handle, Ct.handle, pkg._lib.ptr(ds), pkg._lib.ptr(da_), pkg._lib.ptr(dr),
            pkg._lib.ptr(dt_), pkg._lib.ptr(dsn), Bu, 0.99, 0.995, quirk, 5e-4, 1e-3, C.byref(al), C.byref(cl)))
        assert abs(cl.value - out["critic_loss"]) <= tol * max(1.0, abs(out["critic_loss"]))
        assert abs(al.value - out["actor_loss"]) <= tol * max(1.0, abs(out["actor_loss"]))
        for net, P in ((A, PA), (Cn, PC), (At, PAt), (Ct, PCt)):
            for x, y in zip(net.params(), P):
                assert relerr(x, y) <= tol


def test_policy_act_noise_clamp(pkg):
    from oracle import nn
    rng = np.random.default_rng(5)
    dims, acts = nn.layer_sizes(3, 1, 1.6, True, False)
    net, P = make_net(pkg, rng, dims, acts, torch.float64, 512)
    state = rng.standard_normal((3, 512))
    noise = rng.standard_normal((1, 512))
    ref = nn.policy_act(P, acts, state, noise, 1.2, 1.0)
    out = torch.empty((512, 1), dtype=torch.float64, device="cuda:0")
    dstate, dnoise = to_dev(state.T, torch.float64), to_dev(noise.T, torch.float64)
    pkg._lib.check(net.lib.pdec_policy_act(net.handle, pkg._lib.ptr(dstate), pkg._lib.ptr(dnoise), 512, 1.2, 1.0,
                                           pkg._lib.ptr(out)))
    assert np.abs(out.cpu().numpy().T - ref).max() <= 1e-12
    assert np.abs(ref).max() <= 1.0


def test_randn_moments_and_determinism(pkg):
    rng = np.random.default_rng(0)
    dims = [1, 1]
    net = pkg.HipMLP(dims, [None], None, max_cols=1)
    n = 1 << 20
    a = torch.empty(n, dtype=torch.float32, device="cuda:0")
    b = torch.empty(n, dtype=torch.float32, device="cuda:0")
    pkg._lib.check(net.lib.pdec_randn(net.handle, pkg._lib.ptr(a), n, 0, 1234, 0))
    pkg._lib.check(net.lib.pdec_randn(net.handle, pkg._lib.ptr(b), n, 0, 1234, 0))
    assert torch.equal(a, b)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1.0) < 5e-3
    assert abs(float((a ** 4).mean()) - 3.0) < 0.05
    pkg._lib.check(net.lib.pdec_randn(net.handle, pkg._lib.ptr(b), n, 0, 1234, n // 4))
    assert not torch.equal(a, b)


@pytest.mark.parametrize("ns,dims_scale,drop", [(3, 1.6, False), (3, 1.2, False), (36, 2.0, True), (12, 2.0, True), (1, 0.6, True),
                                                (9, 1.8, True), (44, 3.0, True)])
def test_policy_act_rng_equals_randn_plus_act(pkg, ns, dims_scale, drop):
    """pdec_policy_act_rng (one launch, noise drawn in-kernel) == pdec_randn + pdec_policy_act: the same Philox
    stream element per column; forward within the fp32 tolerance (1e-5 rel) of the fp64 oracle"""
    from oracle import nn
    rng = np.random.default_rng(9)
    dims, acts = nn.layer_sizes(ns, 1, dims_scale, True, drop)     # 3->16->16->1 / 3->12->12->1 / 2-layer ns->h->1
    cols = 1000
    net, P = make_net(pkg, rng, dims, acts, torch.float32, cols)
    state = rng.standard_normal((ns, cols)).astype(np.float32)
    dstate = to_dev(state.T, torch.float32)
    noise = torch.empty((cols, 1), dtype=torch.float32, device="cuda:0")
    seed, off = 4321, 17
    L = pkg._lib
    L.check(net.lib.pdec_randn(net.handle, L.ptr(noise), cols, 0, seed, off))
    out = torch.empty((cols, 1), dtype=torch.float32, device="cuda:0")
    L.check(net.lib.pdec_policy_act_rng(net.handle, L.ptr(dstate), cols, 0.7, 1.0, 1, seed, off, L.ptr(out)))
    ref = nn.policy_act([p.astype(np.float64) for p in P], acts, state.astype(np.float64),
                        noise.cpu().numpy().T.astype(np.float64), 0.7, 1.0)
    assert np.abs(out.cpu().numpy().T - ref).max() <= 2e-5
    L.check(net.lib.pdec_policy_act_rng(net.handle, L.ptr(dstate), cols, 0.7, 1.0, 0, seed, off, L.ptr(out)))   # learning=false
    ref0 = nn.policy_act([p.astype(np.float64) for p in P], acts, state.astype(np.float64), None, 0.0, 1.0, learning=False)
    assert np.abs(out.cpu().numpy().T - ref0).max() <= 1e-5
    assert np.abs(ref - ref0).max() > 0.1


NETS = [(3, 1.6, 7.0, False, 1500),      # C2: 3->16->16->1 / 4->140->140->1 (3-layer fused MFMA passes)
        (36, 2.0, 17.0, True, 1500),     # 2-D Keller-Segel: 36->20->1 / 37->340->1 (2-layer fused MFMA passes)
        (12, 2.0, 17.0, True, 777),      # Keller-Segel10_16: 12->20->1 / 13->340->1
        (9, 1.8, 17.0, True, 300),       # Fluid: 9->18->1 / 10->340->1
        (1, 0.6, 7.0, True, 4100),       # KS22: 1->6->1 / 2->140->1
        (40, 1.0, 7.0, True, 200)]       # widest supported window: 40->10->1 / 41->140->1 (6 k-blocks)


@pytest.mark.parametrize("quirk", [1, 0])
@pytest.mark.parametrize("ns,sa,sc,drop,Bu", NETS)
def test_update_async_and_split_sequence_agree(pkg, quirk, ns, sa, sc, drop, Bu):
    """pdec_ddpg_update_async (4 launches: reduce+ADAM+Polyak fused) and the data-parallel split sequence
    critic_grads -> [all-reduce] -> adam_polyak_step -> actor_grads -> [all-reduce] -> adam_polyak_step
    leave bit-identical networks (replicas must not drift), and both match the oracle"""
    from oracle import nn
    rng = np.random.default_rng(21)
    na = 1
    da, aa = nn.layer_sizes(ns, na, sa, True, drop)
    dc, ac = nn.layer_sizes(ns, na, sc, False, drop)
    dtype, npdt = torch.float32, np.float32
    L = pkg._lib
    nets = []
    for rep in range(2):
        r2 = np.random.default_rng(5)
        nets.append([make_net(pkg, r2, d, a_, dtype, Bu) for d, a_ in ((da, aa), (dc, ac), (da, aa), (dc, ac))])
    PA, PC, PAt, PCt = (nets[0][i][1] for i in range(4))
    optA, optC = nn.Adam(PA, 5e-4), nn.Adam(PC, 1e-3)
    losses = torch.zeros(2, dtype=dtype, device="cuda:0")
    losses2 = torch.zeros(2, dtype=dtype, device="cuda:0")
    for it in range(3):
        s = rng.standard_normal((ns, Bu)).astype(npdt)
        sn = rng.standard_normal((ns, Bu)).astype(npdt)
        a = rng.uniform(-1, 1, (na, Bu)).astype(npdt)
        r = -rng.uniform(0, 1, Bu).astype(npdt)
        t = (rng.uniform(0, 1, Bu) < 0.1).astype(npdt)
        out = nn.ddpg_update(PA, PC, PAt, PCt, optA, optC, aa, ac, s, a, r, t, sn, npdt(np.float32(0.99)), np.float32(0.995), bool(quirk))
        ds, da_, dr, dt_, dsn = (to_dev(s.T, dtype), to_dev(a.T, dtype), to_dev(r, dtype), to_dev(t, dtype), to_dev(sn.T, dtype))
        A, Cn, At, Ct = (nets[0][i][0] for i in range(4))
        L.check(A.lib.pdec_ddpg_update_async(A.handle, Cn.handle, At.handle, Ct.handle, L.ptr(ds), L.ptr(da_), L.ptr(dr),
                                             L.ptr(dt_), L.ptr(dsn), Bu, 0.99, 0.995, quirk, 5e-4, 1e-3, L.ptr(losses)))
        A2, C2, At2, Ct2 = (nets[1][i][0] for i in range(4))
        import ctypes as C_
        L.check(A.lib.pdec_ddpg_critic_grads(A2.handle, C2.handle, At2.handle, Ct2.handle, L.ptr(ds), L.ptr(da_), L.ptr(dr),
                                             L.ptr(dt_), L.ptr(dsn), Bu, 0.99, quirk, 1.0, C_.c_void_p(losses2.data_ptr())))
        L.check(A.lib.pdec_adam_polyak_step(C2.handle, Ct2.handle, 1e-3, 0.9, 0.999, 1e-8, 0.995))
        L.check(A.lib.pdec_ddpg_actor_grads(A2.handle, C2.handle, L.ptr(ds), Bu, 1.0, C_.c_void_p(losses2.data_ptr() + 4)))
        L.check(A.lib.pdec_adam_polyak_step(A2.handle, At2.handle, 5e-4, 0.9, 0.999, 1e-8, 0.995))
        assert torch.equal(losses, losses2)
        lv = losses.cpu().numpy()
        assert abs(lv[0] - out["critic_loss"]) <= 2e-4 * max(1.0, abs(out["critic_loss"]))
        assert abs(lv[1] - out["actor_loss"]) <= 2e-4 * max(1.0, abs(out["actor_loss"]))
        for i, P in enumerate((PA, PC, PAt, PCt)):
            for x, y, z in zip(nets[0][i][0].params(), nets[1][i][0].params(), P):
                assert np.array_equal(x, y)
                assert relerr(x, z) <= 2e-4


@pytest.mark.parametrize("quirk", [1, 0])
@pytest.mark.parametrize("ns,na,sa,sc,drop,Bu", [(1, 1, 0.6, 7.0, True, 3), (12, 1, 2.0, 17.0, True, 3), (9, 1, 1.8, 17.0, True, 3),
                                                (15, 1, 1.0, 25.0, True, 3), (3, 1, 1.6, 7.0, False, 3), (8, 8, 4.8, 56.0, True, 3),
                                                (1, 1, 0.6, 7.0, True, 4), (12, 1, 2.0, 17.0, True, 1), (5, 1, 1.0, 10.0, True, 2),
                                                (9, 1, 1.8, 17.0, True, 7)])
def test_small_update_all_loops_in_one_launch(pkg, quirk, ns, na, sa, sc, drop, Bu):
    """pdec_ddpg_update_small: the reference's update shape (update_loops x minibatch of batch_size = 3 drawn from the
    replay traces, src/PDEagent.jl:317-418) in one launch == the oracle's update applied loop by loop on the same slots"""
    from oracle import nn
    rng = np.random.default_rng(31 + quirk)
    da, aa = nn.layer_sizes(ns, na, sa, True, drop)
    dc, ac = nn.layer_sizes(ns, na, sc, False, drop)
    dtype, npdt = torch.float32, np.float32
    A, PA = make_net(pkg, rng, da, aa, dtype, 16)
    Cn, PC = make_net(pkg, rng, dc, ac, dtype, 16)
    At, PAt = make_net(pkg, rng, da, aa, dtype, 16)
    Ct, PCt = make_net(pkg, rng, dc, ac, dtype, 16)
    optA, optC = nn.Adam(PA, 5e-4), nn.Adam(PC, 1e-3)
    slots_n, loops = 500, 4
    S = rng.standard_normal((slots_n, ns)).astype(npdt)
    Aa = rng.uniform(-1, 1, (slots_n, na)).astype(npdt)
    R = -rng.uniform(0, 1, slots_n).astype(npdt)
    T = (rng.uniform(0, 1, slots_n) < 0.1).astype(npdt)
    i_s = rng.integers(0, slots_n, (loops, Bu))
    i_rt = rng.integers(0, slots_n, (loops, Bu))
    i_sn = rng.integers(0, slots_n, (loops, Bu))
    out = None
    for it in range(loops):
        out = nn.ddpg_update(PA, PC, PAt, PCt, optA, optC, aa, ac, S[i_s[it]].T, Aa[i_s[it]].T, R[i_rt[it]], T[i_rt[it]],
                             S[i_sn[it]].T, npdt(np.float32(0.99)), np.float32(0.995), bool(quirk))
    L = pkg._lib
    dS, dA, dR, dT = to_dev(S, dtype), to_dev(Aa, dtype), to_dev(R, dtype), to_dev(T, dtype)
    idx = torch.as_tensor(np.stack([i_s, i_rt, i_sn]).astype(np.int32), device="cuda:0")
    losses = torch.zeros(2, dtype=dtype, device="cuda:0")
    import ctypes as C_
    L.check(A.lib.pdec_ddpg_update_small(A.handle, Cn.handle, At.handle, Ct.handle, L.ptr(dS), L.ptr(dA), L.ptr(dR), L.ptr(dT),
                                         C_.c_void_p(idx[0].data_ptr()), C_.c_void_p(idx[1].data_ptr()),
                                         C_.c_void_p(idx[2].data_ptr()), loops, Bu, 0.99, 0.995, quirk, 5e-4, 1e-3, L.ptr(losses)))
    lv = losses.cpu().numpy()
    assert abs(lv[0] - out["critic_loss"]) <= 2e-4 * max(1.0, abs(out["critic_loss"]))
    assert abs(lv[1] - out["actor_loss"]) <= 2e-4 * max(1.0, abs(out["actor_loss"]))
    # four chained fp32 updates: ADAM's m / (sqrt(v) + eps) amplifies rounding differences of near-zero gradients, so
    # the chained tolerance is looser than the single-update one used above (2e-4)
    for net, P in ((A, PA), (Cn, PC), (At, PAt), (Ct, PCt)):
        for x, y in zip(net.params(), P):
            assert relerr(x, y) <= 2e-3
    # the ADAM step counters advanced by `loops`: a following generic update continues the same optimiser state
    m = np.empty(A.num_params, dtype=np.float32); v = np.empty_like(m); bp = (C_.c_double * 2)()
    L.check(A.lib.pdec_adam_get_state(A.handle, m.ctypes.data_as(C_.c_void_p), v.ctypes.data_as(C_.c_void_p), bp))
    assert abs(bp[0] - 0.9 ** (loops + 1)) <= 1e-12


def test_library_refuses_a_split_request(pkg, monkeypatch):
    """PDEC_SPLIT != 0 (the deleted bf16-split experiment, HISTORY.md §3.2a) is an error of the fused passes, never a silent
    exact-f32 run"""
    import ctypes as C
    from oracle import nn
    rng = np.random.default_rng(2)
    da, aa = nn.layer_sizes(3, 1, 1.6, True, False)
    dc, ac = nn.layer_sizes(3, 1, 7.0, False, False)
    A, _ = make_net(pkg, rng, da, aa, torch.float32, 256)
    Cn, _ = make_net(pkg, rng, dc, ac, torch.float32, 256)
    s = to_dev(rng.standard_normal((256, 3)), torch.float32)
    L = torch.zeros(2, dtype=torch.float32, device="cuda:0")
    monkeypatch.setenv("PDEC_SPLIT", "a")
    with pytest.raises(pkg.PdecError, match="no longer exist"):
        pkg._lib.check(A.lib.pdec_ddpg_actor_grads(A.handle, Cn.handle, pkg._lib.ptr(s), 256, 1.0, C.c_void_p(L.data_ptr() + 4)))
    monkeypatch.setenv("PDEC_SPLIT", "0")
    pkg._lib.check(A.lib.pdec_ddpg_actor_grads(A.handle, Cn.handle, pkg._lib.ptr(s), 256, 1.0, C.c_void_p(L.data_ptr() + 4)))
    torch.cuda.synchronize()


