"""CPU tests (-m "not gpu"): the oracle against the reference's golden vectors, the C
restatement against the NumPy one, analytic known-answer tests for the unpinned parts."""
import numpy as np
import pytest

from util import ks_pair, load_golden


@pytest.mark.parametrize("name", ["ks22", "ks200", "ks22_global"])
def test_ks_numpy_oracle_matches_reference_golden(pkg, name):
    from oracle import ks
    setup, cfg, g = ks_pair(pkg, name)
    y, p, a, r = g["y"], g["p"], g["action"], g["reward"]
    assert max(np.abs(ks.do_step(cfg, y[t], p[t + 1]) - y[t + 1]).max() for t in range(50)) <= 1e-13
    assert max(np.abs(ks.do_step(cfg, y[t], p[t + 1], np.complex64) - y[t + 1]).max() for t in range(0, 50, 7)) <= 2e-5
    act = (lambda t: a[t]) if cfg.mono else (lambda t: a[t][None, :])
    assert max(np.abs(ks.prepare_action(cfg, act(t)) - p[t]).max() for t in range(51)) <= 1e-12
    assert max(np.abs(ks.reward_function(cfg, y[t], a[t][None, :], (a[t] - a[t - 1])[None, :]) - r[t]).max()
               for t in range(1, 51)) <= 1e-13


@pytest.mark.parametrize("name", ["ks22", "ks200"])
def test_ks_c_oracle_matches_reference_golden(pkg, name):
    from oracle import c_oracle
    setup, cfg, g = ks_pair(pkg, name)
    plan = c_oracle.KSPlan(cfg.nx, cfg.Lx)
    assert max(np.abs(plan.step(g["y"][t], g["p"][t + 1]) - g["y"][t + 1]).max() for t in range(50)) <= 1e-12


def test_c_batch_env_matches_numpy_oracle(pkg):
    from oracle import c_oracle, ks
    setup, cfg, g = ks_pair(pkg, "ks200")
    env = c_oracle.KSBatchEnv(cfg)
    y = g["y"][1:9].copy()
    a, ap = g["action"][2:10], g["action"][1:9]
    state, reward, done = env.step(y, a, ap)
    for b in range(8):
        o = ks.env_step(cfg, g["y"][1 + b], ap[b][None], a[b][None], 0.0)
        assert np.abs(y[b] - o["y"]).max() <= 1e-12
        assert np.abs(reward[b] - o["reward"]).max() <= 1e-13
        assert np.abs(state[b].T - o["state"]).max() <= 1e-13
        assert np.abs(reward[b] - g["reward"][2 + b]).max() <= 1e-12     # the reference's own log
    assert done.sum() == 0


def test_keller_segel_oracle_matches_reference_golden():
    from oracle import keller_segel as kg
    g = load_golden("kseg_hook.npz")
    cfg = kg.KSegConfig()
    idx = range(0, len(g["idx"]), 6)
    assert max(np.abs(kg.do_step(cfg, g["y_t"][i], g["p_t1"][i], 32) - g["y_t1"][i]).max() for i in idx) <= 5e-8
    assert max(np.abs(kg.prepare_action(cfg, g["action_t1"][i][None]) - g["p_t1"][i]).max() for i in idx) == 0
    assert max(np.abs(kg.reward_function(cfg, g["y_t1"][i], g["action_t1"][i][None],
                                         (g["action_t1"][i] - g["action_t"][i])[None]) - g["reward_t1"][i]).max()
               for i in idx) <= 1e-15


def test_product_setup_tables_match_oracle(pkg):
    """host logic of the product (prepare_gaussians incl. Julia's range length, rectangles)"""
    from oracle import ks, keller_segel as kg
    for name in ("ks22", "ks200", "ks22_global"):
        setup, cfg, g = ks_pair(pkg, name)
        assert np.array_equal(setup.gaussians, cfg.gaussians)
        assert np.array_equal(setup.gaussians_actuators, cfg.gaussians_actuators)
    s, c = pkg.KellerSegelSetup(), kg.KSegConfig()
    assert np.array_equal(s.gaussians, c.gaussians) and np.array_equal(s.gaussians_actuators, c.gaussians_actuators)
    assert s.state_shape == (12, 16) and pkg.KSSetup.bench_C2(256).state_shape == (3, 64)
    b = pkg.KSSetup.bench_C2(256)
    oc = ks.KSConfig(256, b.Lx, b.sensor_positions, window_size=3)
    assert np.array_equal(b.gaussians, oc.gaussians)


def test_julia_range_length_quirk(pkg):
    """KSSetup.jl:87 `dx-50dx:dx:Lx+50dx`: nx+99 points for KS22 (291-point support),
    the ideal nx+100 for KS200 -- both confirmed by the golden `p` rows (prepare_action parity 1e-13)"""
    jc = pkg.julia_compat if hasattr(pkg, "julia_compat") else __import__("importlib").import_module(
        "distributedconvrl-pde-control_amd.julia_compat")
    for nx, Lx, n in ((192, 22.0, 291), (240, 200.0, 340)):
        dx = Lx / nx
        assert len(jc.float_range(dx - 50 * dx, dx, Lx + 50 * dx)) == n
    assert len(jc.float_range(0.0, 0.5, 2.0)) == 5 and len(jc.float_range(1.0, 1.0, 10.0)) == 10


# ---- fluid: analytic known-answer tests (PARITY UNPINNED by the reference)
def test_fluid_pad_chop_and_dealiasing():
    from oracle import fluid
    cfg = fluid.FluidConfig(nx=32)
    rng = np.random.default_rng(0)
    f = rng.standard_normal((32, 32)) + 1j * rng.standard_normal((32, 32))
    assert np.abs(fluid.chop(cfg, fluid.pad(cfg, f)) - f).max() == 0
    lo = np.fft.fft2(rng.standard_normal((32, 32)))
    lo = lo * ((np.abs(cfg.kx_repeat) <= 2 * np.pi * 5) & (np.abs(cfg.ky_repeat) <= 2 * np.pi * 5))
    a1 = fluid.advection(cfg, lo)
    cfg.ifpad = 0
    a0 = fluid.advection(cfg, lo)
    assert np.abs(a1 - a0).max() <= 1e-12 * np.abs(a0).max()     # checks the 1.5*1.5 factor


def test_fluid_single_mode_and_viscous_decay():
    from oracle import fluid
    cfg = fluid.FluidConfig(nx=32, nu=1e-2)
    om = np.cos(2 * np.pi * 3 * cfg.xx + 2 * np.pi * 2 * cfg.yy)
    oh = np.fft.fft2(om)
    assert np.abs(fluid.advection(cfg, oh)).max() <= 1e-9 * np.abs(oh).max()   # one mode: zero Jacobian
    h = 0.01
    out = fluid.rk4(cfg, oh, 0 * oh, h)
    z = -cfg.nu * cfg.kx2ky2 * h
    taylor4 = 1 + z + z ** 2 / 2 + z ** 3 / 6 + z ** 4 / 24            # RK4 = degree-4 Taylor of exp
    assert np.abs(out - taylor4 * oh).max() <= 1e-9 * np.abs(oh).max()


# ---- NN: finite differences + committed torch-autograd golden
def test_nn_backward_finite_differences():
    from oracle import nn
    rng = np.random.default_rng(1)
    dims, acts = nn.layer_sizes(3, 1, 1.6, False, False)
    P = nn.glorot_uniform(rng, dims, np.float64)
    for i in range(1, len(P), 2):
        P[i] = rng.standard_normal(P[i].shape) * 0.1
    x = rng.standard_normal((dims[0], 9))
    y, zs, as_ = nn.forward(P, acts, x, keep=True)
    g, dx = nn.backward(P, acts, zs, as_, np.ones_like(y))
    eps = 1e-6
    for pi in (0, 1, 2, 4):
        W = P[pi]
        idx = tuple(rng.integers(0, s) for s in W.shape)
        W[idx] += eps; up = nn.forward(P, acts, x).sum()
        W[idx] -= 2 * eps; dn = nn.forward(P, acts, x).sum()
        W[idx] += eps
        assert abs((up - dn) / (2 * eps) - g[pi][idx]) <= 1e-6
    x2 = x.copy(); x2[1, 4] += eps
    assert abs((nn.forward(P, acts, x2).sum() - y.sum()) / eps - dx[1, 4]) <= 1e-5


def test_nn_ddpg_matches_torch_autograd_golden():
    """independent cross-check generated by tests/golden/make_nn_golden.py (torch autograd, CPU)"""
    from oracle import nn
    g = load_golden("nn_torch_golden.npz")
    for quirk in (1, 0):
        A = [g[f"A{i}"] for i in range(6)]; Cn = [g[f"C{i}"] for i in range(6)]
        At = [g[f"At{i}"] for i in range(6)]; Ct = [g[f"Ct{i}"] for i in range(6)]
        aa, ac = [1, 1, 2], [1, 1, 0]
        out = nn.ddpg_losses_and_grads(A, Cn, At, Ct, aa, ac, g["s"], g["a"], g["r"], g["t"], g["sn"], 0.99, bool(quirk))
        assert abs(out["critic_loss"] - g[f"closs_q{quirk}"]) <= 1e-12
        for i in range(6):
            assert np.abs(out["gC"][i] - g[f"gC{i}_q{quirk}"]).max() <= 1e-12
        out2 = nn.actor_grads(A, Cn, aa, ac, g["s"])
        assert abs(out2["actor_loss"] - g["aloss"]) <= 1e-12
        for i in range(6):
            assert np.abs(out2["gA"][i] - g[f"gA{i}"]).max() <= 1e-12


def test_quirk_loss_closed_form_equals_the_explicit_broadcast():
    """oracle/nn.py evaluates the reference's (1xBu).+(Bu) double mean (src/PDEagent.jl:388-393) without the Bu x Bu matrix
    above Bu = 2048: the closed form must be the explicit broadcast (checked at a size where both run), gradients untouched"""
    from oracle import nn
    rng = np.random.default_rng(4)
    da, aa = nn.layer_sizes(3, 1, 1.6, True, False)
    dc, ac = nn.layer_sizes(3, 1, 7.0, False, False)
    mk = lambda d: [p if i % 2 == 0 else rng.standard_normal(p.shape) * 0.1 for i, p in enumerate(nn.glorot_uniform(rng, d, np.float64))]
    PA, PC, PAt, PCt = mk(da), mk(dc), mk(da), mk(dc)
    Bu = 2500
    s, sn = rng.standard_normal((3, Bu)), rng.standard_normal((3, Bu))
    a, r, t = rng.uniform(-1, 1, (1, Bu)), -rng.uniform(0, 1, Bu), (rng.uniform(0, 1, Bu) < 0.1) * 1.0
    out = nn.ddpg_losses_and_grads(PA, PC, PAt, PCt, aa, ac, s, a, r, t, sn, 0.99, True)
    d = 0.99 * (1 - t) * out["qt"] - out["q"]
    explicit = np.mean((r[None, :] + d[:, None]) ** 2)
    assert abs(out["critic_loss"] - explicit) <= 1e-13 * explicit


def test_c_agent_matches_numpy_oracle():
    from oracle import c_oracle, nn
    rng = np.random.default_rng(2)
    da, aa = nn.layer_sizes(3, 1, 1.6, True, False)
    dc, ac = nn.layer_sizes(3, 1, 7.0, False, False)
    mk = lambda d: [p if i % 2 == 0 else rng.standard_normal(p.shape) * 0.1 for i, p in enumerate(nn.glorot_uniform(rng, d, np.float64))]
    PA, PC, PAt, PCt = mk(da), mk(dc), mk(da), mk(dc)
    ag = c_oracle.Agent(da, aa, dc, ac)
    for w, P in enumerate((PA, PC, PAt, PCt)):
        ag.set(w, P)
    optA, optC = nn.Adam(PA, 5e-4), nn.Adam(PC, 1e-3)
    Bu = 257
    for it in range(2):
        s, sn = rng.standard_normal((3, Bu)), rng.standard_normal((3, Bu))
        a, r, t = rng.uniform(-1, 1, (1, Bu)), -rng.uniform(0, 1, Bu), (rng.uniform(0, 1, Bu) < 0.1) * 1.0
        out = nn.ddpg_update(PA, PC, PAt, PCt, optA, optC, aa, ac, s, a, r, t, sn, 0.99, 0.995, True)
        al, cl = ag.ddpg_update(s.T, a.T, r, t, sn.T, 0.99, 0.995, 1, 5e-4, 1e-3)
        assert abs(al - out["actor_loss"]) <= 1e-12 and abs(cl - out["critic_loss"]) <= 1e-12
    for w, P in enumerate((PA, PC, PAt, PCt)):
        for x, y in zip(ag.get(w), P):
            assert np.abs(x - y).max() <= 1e-12
    st = rng.standard_normal((40, 3)); nz = rng.standard_normal((40, 1))
    assert np.abs(ag.act(st, nz, 1.2, 1.0) - nn.policy_act(PA, aa, st.T, nz.T, 1.2, 1.0).T).max() <= 1e-13


def test_adam_one_step_known_answer():
    """first ADAM step moves every parameter by eta*sign(g) (up to eps): constants from agent.jld2"""
    from oracle import nn
    g = load_golden("ks22_agent.npz")
    assert np.allclose(sorted(set(np.round(g["adam_eta"], 6))), [5e-4, 1e-3])
    assert np.allclose(g["adam_beta_eps"], [0.9, 0.999, 1e-8])
    P = [np.ones((2, 2)), np.zeros(2)]
    opt = nn.Adam(P, 1e-3)
    G = [np.array([[1.0, -2.0], [0.5, -0.1]]), np.array([3.0, -4.0])]
    P2 = opt.step([p.copy() for p in P], G)
    assert np.allclose(P2[0], P[0] - 1e-3 * np.sign(G[0]), atol=1e-9)
    assert np.allclose(P2[1], P[1] - 1e-3 * np.sign(G[1]), atol=1e-9)


def test_ks_fd_rk4_variant_converges_to_the_spectral_step():
    """The RK4 + 5-point-FD KS variant (oracle/ks.py: rhs_fd, do_step_rk4_fd; stencils of KSSetup.jl:55-59) is a different
    discretisation from the reference's CNAB2 step, so it gets its own known answers: (i) the stencils are exact on
    low-order trigonometric data up to O(dx^2); (ii) one short control step approaches the spectral CNAB2 step at
    second order in dx (error ratio ~4 per grid doubling)."""
    from oracle import ks
    errs = []
    for nx in (64, 128, 256):
        cfg = ks.KSConfig(nx, 22.0, np.arange(1, nx + 1, nx // 8), dt=2e-4, oversampling=20)
        x = cfg.xx
        y = 2.0 * np.sin(2 * np.pi * x / 22.0) + 0.5 * np.cos(4 * np.pi * x / 22.0)
        p = 0.3 * np.sin(6 * np.pi * x / 22.0)
        k = 2 * np.pi / 22.0
        exact = (-(y * (2.0 * k * np.cos(k * x) - 0.5 * 2 * k * np.sin(2 * k * x)))
                 - (-(k ** 2) * 2.0 * np.sin(k * x) - 0.5 * (2 * k) ** 2 * np.cos(2 * k * x))
                 - ((k ** 4) * 2.0 * np.sin(k * x) + 0.5 * (2 * k) ** 4 * np.cos(2 * k * x)) + p)
        assert np.abs(ks.rhs_fd(cfg, y, p) - exact).max() <= 3.0 * cfg.dx ** 2
        a = ks.do_step_rk4_fd(cfg, y, p)
        b = ks.do_step(cfg, y, p)
        errs.append(np.abs(a - b).max())
    assert errs[0] / errs[1] > 3.0 and errs[1] / errs[2] > 3.0, errs
    assert errs[2] < 1e-5


def test_kseg2d_oracle_reduces_to_pinned_1d():
    """oracle/keller_segel2d.py on a field that does not depend on y == the 1-D oracle, which the reference's
    golden pins (test above); transposing the grid commutes with the step"""
    from oracle import keller_segel as k1, keller_segel2d as k2
    g = load_golden("kseg_hook.npz")
    c1, c2 = k1.KSegConfig(), k2.KSeg2DConfig()
    for b in (0, 11):
        y1, a, ap = g["y_t"][b], g["action_t1"][b][None, :], g["action_t"][b][None, :]
        y2 = np.repeat(y1[:, None, :], c2.ny, axis=1)
        p2 = k2.prepare_action(c2, a)
        assert np.abs(p2 - g["p_t1"][b][None, :]).max() <= 1e-12
        out = k2.do_step(c2, y2, p2)
        assert np.abs(out - g["y_t1"][b][:, None, :]).max() <= 1e-7
        assert np.abs(out - k1.do_step(c1, y1, g["p_t1"][b])[:, None, :]).max() <= 1e-13
        assert np.abs(k2.reward_function(c2, out, a, a - ap) - g["reward_t1"][b]).max() <= 1e-9
        s1, s2 = k1.featurize(c1, y1, None), k2.featurize(c2, y2, None)
        idx = [j for _ in range(3) for j in range(3)]
        exp = np.concatenate([s1[0:3][idx], s1[3:6][idx]])
        assert np.abs(s2 - np.concatenate([exp, exp])).max() <= 1e-14
    rng = np.random.default_rng(0)
    c = k2.KSeg2DConfig(nx=20, ny=20, substeps=2)
    y, p = 1 + 0.2 * rng.standard_normal((2, 20, 20)), rng.standard_normal((20, 20))
    out = k2.do_step(c, y, p)
    outT = k2.do_step(c, np.swapaxes(y, 1, 2), p.T)
    assert np.abs(np.swapaxes(outT, 1, 2) - out).max() <= 1e-13


def test_philox_stream_known_answer_and_derived_draws():
    """oracle/rng.py (the counter stream that replaces Julia's RNGs on the device): Philox4x32-10 known answer of the
    Random123 distribution (counter 0, key 0), and the derived draws -- uniform slot indices stay inside
    [0, n_valid - stride) with s' one stride further, normals have unit variance"""
    from oracle import rng
    w = rng.philox4x32(np.array([0], dtype=np.uint64), 0)[0]
    assert [int(x) for x in w] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert np.array_equal(rng.words(3, 10, 7), rng.philox4x32(np.array([10, 11], dtype=np.uint64), 3).reshape(-1)[:7])
    sl = rng.sample_slots(5, 0, 4000, n_valid=800, n_rt=2000, capacity=800, stride=8)
    lg = (sl[1] - (2000 - 800) % 800) % 800          # position relative to the oldest entry
    assert lg.min() >= 0 and lg.max() < 800 - 8 and np.array_equal(sl[2], (sl[0] + 8) % 808)
    assert len(np.unique(lg)) > 700
    z = rng.randn(1, 0, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01


def test_c_oracle_is_clean_under_asan_and_ubsan():
    """SURVEY.md §5 (sanitizers; CPU build only -- GPU ASAN is unavailable on the pool): the oracle's C restatement rebuilt
    with -fsanitize=address,undefined behind oracle/c/sanitize_main.c, which drives every exported entry point at ragged
    sizes with exactly-sized buffers; any out-of-bounds access / UB aborts the run"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle", "c"), "sanitize"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="3")
    r = subprocess.run([os.path.join(root, "oracle", "_build", "oracle_sanitize")], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "sanitize: OK" in r.stdout, r.stdout + r.stderr


def test_global_agent_last_logged_action_is_the_saved_best_actor_on_the_last_state():
    """The mono / global path (KSglobalSetup.jl:211-249 featurize -> [8, 1] state; actor 8 -> 48 -> 8, src/PDEagent.jl:18-30) against
    reference-held data: scripts/KS/KS22_global-agent/saves/hook.jld2 logs the best episode (noise 1.2 * 0.2^k, small by then) and
    keeps a copy of the actor taken at that episode's end (`bestNNA`, src/PDEhook.jl:68-75).  The action of the last logged row is that
    actor -- 20 minibatch updates earlier -- on the state of the row before: the oracle's featurize + forward give it to 7e-2 on all
    eight actuators; the sensors in reversed order miss by 0.70, a state scaled by 2 by 0.50."""
    from oracle import ks, nn
    g = load_golden("ks22_global_hook.npz")
    nx, Lx, stride, sigma = int(g["nx"]), float(g["Lx"]), int(g["sensor_stride"]), float(g["sigma"])
    cfg = ks.KSConfig(nx, Lx, np.arange(1, nx + 1, stride), sigma_sensors=sigma, sigma_actuators=sigma, mono=True,
                      disturbance_in_step=False)
    P = [g["best_W1"], g["best_b1"], g["best_W2"], g["best_b2"]]
    assert [p.shape for p in P] == [(48, 8), (48,), (8, 48), (8,)]
    acts = [nn.RELU, nn.TANH]
    st = ks.featurize(cfg, g["y"][49])
    assert st.shape == (8, 1)
    a = g["action"][50].reshape(-1)
    f = lambda s: nn.policy_act(P, acts, s, None, 0.0, 1.0, learning=False).reshape(-1)
    assert np.abs(f(st) - a).max() <= 0.09
    assert np.abs(f(st[::-1]) - a).max() > 0.4 and np.abs(f(2 * st) - a).max() > 0.3
