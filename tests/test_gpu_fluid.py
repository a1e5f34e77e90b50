"""GPU parity of the 2-D fluid path (src/fluid_rk4.jl + scripts/Fluid/setup/FluidSetup.jl) against the
NumPy oracle (oracle/fluid.py; PARITY UNPINNED by reference artifacts -- the oracle itself is pinned by the
analytic KATs in test_oracle.py) plus size-independent properties at the reference's grid sizes.
fp64 throughout; tolerances are relative to max|reference| and written at each assert."""
import numpy as np
import pytest

from util import to_dev

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F64 = torch.float64


def _mem(z):    # Julia complex [.., ny, nx] -> memory [.., nx, ny, 2]
    z = np.swapaxes(np.asarray(z, dtype=np.complex128), -1, -2)
    return np.ascontiguousarray(np.stack([z.real, z.imag], axis=-1))


def _jul(t):    # memory [.., nx, ny, 2] -> Julia complex [.., ny, nx]
    a = t.detach().cpu().numpy()
    return np.swapaxes(a[..., 0] + 1j * a[..., 1], -1, -2)


_PAIRS = {}


def _pair(pkg, n, ifpad=1, spa=4, K=None, variance=0.08, **kw):
    """(product setup, oracle config); memoised -- the sensor tables of a 512 x 512 grid take ~35 s each to build on the host"""
    from oracle import fluid
    key = (n, ifpad, spa, K, variance, tuple(sorted(kw.items())))
    if key not in _PAIRS:
        setup = pkg.FluidSetup(nx=n, ifpad=ifpad, sensors_per_axis=spa, variance=variance, oversampling=K, **kw)
        cfg = fluid.FluidConfig(nx=n, ifpad=ifpad, sensors_per_axis=spa, variance=variance, oversampling=K)
        _PAIRS[key] = (setup, cfg)
    return _PAIRS[key]


def _fields(cfg, B, seed, hermitian=True):
    from oracle import fluid
    rng = np.random.default_rng(seed)
    y = np.stack([fluid.ic(cfg, 4, rng) for _ in range(B)])
    if not hermitian:     # a generic (non-real-field) spectrum exercises the Herm() packing in full
        y = y + 0.05 * np.abs(y).max() * (rng.standard_normal(y.shape) + 1j * rng.standard_normal(y.shape))
    p = np.stack([np.fft.fft2(rng.standard_normal((cfg.ny, cfg.nx))) for _ in range(B)])
    return y, p


@pytest.mark.parametrize("n,ifpad,herm", [(16, 1, True), (16, 1, False), (16, 0, False), (32, 1, False), (24, 1, True),
                                          (64, 1, True), (64, 0, True), (128, 1, True),
                                          # padded / un-padded lengths served by the one-line-per-wave transforms
                                          (128, 0, False), (256, 0, False), (256, 1, False), (256, 1, True), (512, 0, True),
                                          (512, 1, False)])
def test_rhs_matches_oracle(pkg, n, ifpad, herm):
    from oracle import fluid
    setup, cfg = _pair(pkg, n, ifpad, spa=8 if n >= 256 else 4, variance=0.04 if n >= 256 else 0.08)
    B = 3 if n < 256 else 2
    y, p = _fields(cfg, B, seed=n + ifpad, hermitian=herm)
    env = pkg.PDEenv(setup, B=B, dtype=F64)
    out = _jul(env.rhs(to_dev(_mem(y), F64), to_dev(_mem(p), F64)))
    for b in range(B):
        ref = fluid.rhs(cfg, y[b].copy(), p[b])
        assert np.abs(out[b] - ref).max() <= 1e-11 * np.abs(ref).max(), (b, np.abs(out[b] - ref).max(), np.abs(ref).max())


@pytest.mark.parametrize("n,B", [(256, 14), (512, 5)])
def test_rhs_with_several_tiles_per_workgroup(pkg, n, B):
    """round 4: the x-pass of the padded grids is a persistent kernel (fluid_k2p_kernel) whose workgroups WALK their share of
    the (trajectory, 8-column) tiles with the next tile prefetched behind the current transforms -- 14 x 48 = 672 and 5 x 96 =
    480 tiles on 256 CUs: two and three tiles per workgroup, ragged (the B = 2 cases above give every workgroup one tile and
    never enter the loop's steady state).  Right-hand side of every trajectory against the oracle (src/fluid_rk4.jl:134-190)."""
    from oracle import fluid
    setup, cfg = _pair(pkg, n, 1, spa=8, variance=0.04)
    y, p = _fields(cfg, B, seed=7 * n + B, hermitian=True)
    env = pkg.PDEenv(setup, B=B, dtype=F64)
    out = _jul(env.rhs(to_dev(_mem(y), F64), to_dev(_mem(p), F64)))
    for b in range(B):
        ref = fluid.rhs(cfg, y[b].copy(), p[b])
        assert np.abs(out[b] - ref).max() <= 1e-11 * np.abs(ref).max(), (b, np.abs(out[b] - ref).max(), np.abs(ref).max())


@pytest.mark.parametrize("n,ifpad", [(16, 1), (32, 1), (32, 0), (256, 1), (128, 0)])
def test_do_step_rk4_matches_oracle(pkg, n, ifpad):
    from oracle import fluid
    K = 3 if n < 128 else 2
    setup, cfg = _pair(pkg, n, ifpad, K=K, spa=8 if n >= 128 else 4, variance=0.04 if n >= 128 else 0.08)
    B = 2
    y, p = _fields(cfg, B, seed=7)
    env = pkg.PDEenv(setup, B=B, dtype=F64)
    yin = to_dev(_mem(y), F64)
    out, flags = env.do_step(yin, to_dev(_mem(p), F64))
    assert np.abs(_jul(yin) - y).max() == 0.0          # do_step must not touch its input (src/PDEenv.jl:216-218)
    for b in range(B):
        ref = fluid.do_step(cfg, y[b], p[b], K)
        assert np.abs(_jul(out)[b] - ref).max() <= 1e-11 * np.abs(ref).max()
    assert int(flags.sum()) == 0


def test_closures_and_env_step_match_oracle(pkg):
    from oracle import fluid
    n, spa, K = 32, 4, 2
    setup, cfg = _pair(pkg, n, 1, spa=spa, K=K)
    B = 3
    rng = np.random.default_rng(3)
    y, _ = _fields(cfg, B, seed=11)
    a0 = rng.uniform(-1, 1, (B, 1, spa * spa))
    a1 = rng.uniform(-1, 1, (B, 1, spa * spa))
    env = pkg.PDEenv(setup, B=B, dtype=F64, y0=y)
    # featurize at reset (FluidSetup.jl:204-245)
    for b in range(B):
        st = fluid.featurize(cfg, y[b])
        assert np.abs(env.state[b].cpu().numpy().T - st).max() <= 1e-12 * max(1.0, np.abs(st).max())
    # prepare_action (:247-261)
    pa = _jul(env.prepare_action(to_dev(a1.reshape(B, -1, 1), F64)))
    for b in range(B):
        ref = fluid.prepare_action(cfg, a1[b])
        assert np.abs(pa[b] - ref).max() <= 1e-12 * np.abs(ref).max()
    # whole (env)(action): src/PDEenv.jl:195-241
    env.action.copy_(to_dev(a0.reshape(B, -1, 1), F64))
    env(to_dev(a1.reshape(B, -1, 1), F64))
    for b in range(B):
        p = fluid.prepare_action(cfg, a1[b])
        yn = fluid.do_step(cfg, y[b], p, K)
        assert np.abs(_jul(env.p)[b] - p).max() <= 1e-12 * np.abs(p).max()
        assert np.abs(_jul(env.y)[b] - yn).max() <= 1e-11 * np.abs(yn).max()
        r = fluid.reward_function(cfg, yn, a1[b], a1[b] - a0[b])
        assert np.abs(env.reward[b].cpu().numpy() - r).max() <= 1e-11 * max(1.0, np.abs(r).max())
        st = fluid.featurize(cfg, yn)
        assert np.abs(env.state[b].cpu().numpy().T - st).max() <= 1e-11 * max(1.0, np.abs(st).max())
    # stand-alone reward closure
    rr = env.reward_function().cpu().numpy()
    assert np.abs(rr - env.reward.cpu().numpy()).max() <= 1e-14


def test_reward_blowup_flag(pkg):
    """check_max_value = "reward": the episode ends when max|reward| > max_value (src/PDEenv.jl:232-237)"""
    setup, cfg = _pair(pkg, 16, 1, spa=4, K=1, max_value=1e-9)
    y, _ = _fields(cfg, 2, seed=5)
    env = pkg.PDEenv(setup, B=2, dtype=F64, y0=y)
    env(torch.zeros(env._ashape, dtype=F64, device="cuda:0"))
    assert bool(env.done.all().item())


@pytest.mark.parametrize("n", [128, 256])
def test_single_fourier_mode_has_zero_advection_and_exact_viscous_rk4(pkg, n):
    """Size-independent properties at the reference's own grid sizes: a single Fourier mode
    has zero Jacobian, so rhs = -nu k^2 w and one RK4 sub-step multiplies it by the degree-4 Taylor
    polynomial of exp(-nu k^2 h)."""
    setup, cfg = _pair(pkg, n, 1, spa=8, K=1)
    ky, kx = 3, 5
    w = np.zeros((n, n), dtype=complex)
    w[ky, kx] = n * n * 0.5
    w[-ky, -kx] = n * n * 0.5
    env = pkg.PDEenv(setup, B=1, dtype=F64)
    z = torch.zeros_like(env.y)
    out = _jul(env.rhs(to_dev(_mem(w[None]), F64), z))[0]
    lin = -cfg.nu * cfg.kx2ky2 * w
    assert np.abs(out - lin).max() <= 1e-12 * np.abs(w).max()
    yn, _ = env.do_step(to_dev(_mem(w[None]), F64), z)
    x = -cfg.nu * cfg.kx2ky2[ky, kx] * (cfg.dt / 1)
    poly = 1 + x + x * x / 2 + x ** 3 / 6 + x ** 4 / 24
    assert np.abs(_jul(yn)[0] - poly * w).max() <= 1e-13 * np.abs(w).max()


def test_padded_equals_unpadded_for_band_limited_data(pkg):
    """3/2-rule property (checks the 1.5*1.5 factor, src/fluid_rk4.jl:176): for spectra confined to |k| < n/3 the
    de-aliased and the plain product agree."""
    n = 96
    rng = np.random.default_rng(2)
    f = rng.standard_normal((n, n))
    fh = np.fft.fft2(f)
    kk = np.abs(np.fft.fftfreq(n, 1.0 / n))
    fh[kk > n // 6, :] = 0
    fh[:, kk > n // 6] = 0
    outs = []
    for ifpad in (1, 0):
        setup, _ = _pair(pkg, n, ifpad, spa=4)
        env = pkg.PDEenv(setup, B=1, dtype=F64)
        outs.append(_jul(env.rhs(to_dev(_mem(fh[None]), F64), torch.zeros_like(env.y)))[0])
    assert np.abs(outs[0] - outs[1]).max() <= 1e-11 * np.abs(outs[0]).max()


@pytest.mark.parametrize("n", [64, 128, 192, 256, 384, 512, 768])
def test_wave_fft_engine_matches_numpy(pkg, n):
    """the register-resident one-line-per-wave FFT (csrc/wave_fft.hpp) the fluid kernels are built on: forward and
    unnormalised inverse against numpy.fft (= FFTW's conventions), fp64 <= 1e-13 relative"""
    rng = np.random.default_rng(n)
    L = pkg._lib
    lib = L.init(0)
    nl = 9
    x = rng.standard_normal((nl, n)) + 1j * rng.standard_normal((nl, n))
    xin = to_dev(np.stack([x.real, x.imag], axis=-1), F64)
    out = torch.empty_like(xin)
    L.check(lib.pdec_debug_wave_fft(L.ptr(xin), L.ptr(out), n, nl, -1))
    got = out.cpu().numpy()
    ref = np.fft.fft(x, axis=1)
    assert np.abs(got[..., 0] + 1j * got[..., 1] - ref).max() <= 1e-13 * np.abs(ref).max()
    L.check(lib.pdec_debug_wave_fft(L.ptr(xin), L.ptr(out), n, nl, +1))
    got = out.cpu().numpy()
    ref = np.fft.ifft(x, axis=1) * n
    assert np.abs(got[..., 0] + 1j * got[..., 1] - ref).max() <= 1e-13 * np.abs(ref).max()


def test_device_initialiser_matches_oracle_ic(pkg):
    """pdec_fluid_ic (taylorvtx sum + fft2 on the GPU, src/fluid_rk4.jl:54-120) against the ORACLE's ic(caseno)
    (oracle/fluid.py) with shared draws: the vortex table handed to the device is drawn in the reference's order, so
    the oracle consuming the same generator produces the same B fields.  fp64, <= 1e-11 relative."""
    import ctypes as C
    from oracle import fluid
    setup, cfg = _pair(pkg, 64, 1, spa=4, K=2)
    B = 3
    env = pkg.PDEenv(setup, B=B, dtype=torch.float64)
    for case in (3, 4, 2, 1):
        v = setup.ic_vortices(case, np.random.default_rng(7), B)
        rng = np.random.default_rng(7)
        refs = [fluid.ic(cfg, case, rng) for _ in range(B)]          # same stream, trajectory after trajectory
        out = torch.empty_like(env.y)
        pkg._lib.check(env.lib.pdec_fluid_ic(env.handle, v.ctypes.data_as(C.POINTER(C.c_double)), v.shape[1], pkg._lib.ptr(out)))
        got = _jul(out)
        for b in range(B):
            assert np.abs(got[b] - refs[b]).max() <= 1e-11 * np.abs(refs[b]).max(), (case, b)
        if case in (3, 4):
            assert np.abs(refs[0] - refs[1]).max() > 1e-3 * np.abs(refs[0]).max()      # the draws differ per trajectory
    setup.evaluation = False
    y0 = setup.random_init_device(env, np.random.default_rng(3))
    assert y0.shape == env.y.shape and bool(torch.isfinite(y0).all())
    # Hermitian spectrum of a real field: the inverse transform has no imaginary part
    z = y0[1].cpu().numpy()
    f = np.fft.ifft2((z[..., 0] + 1j * z[..., 1]).T)
    assert np.abs(f.imag).max() <= 1e-12 * np.abs(f.real).max()


@pytest.mark.slow
def test_config_c5_full_grid_vs_oracle(pkg):
    """BASELINE.json configs[4] at the FULL 512 x 512 grid (3/2-rule padding -> 768 x 768 transforms), 16 x 16 sensors,
    fp64: do_step with K = 2 RK4 sub-steps and one fused (env)(action) against the oracle, B = 2 (the oracle needs
    ~40 padded 2-D FFTs per trajectory here).  do_step <= 1e-11 relative; the fused env step (actuator synthesis +
    fft2 of the forcing + 2 RK4 sub-steps through 768-point lines) <= 3e-11 (measured 1.3e-11)."""
    from oracle import fluid
    n, spa, K, B = 512, 16, 2, 2
    setup, cfg = _pair(pkg, n, 1, spa=spa, K=K, variance=0.022)
    y, p = _fields(cfg, B, seed=21)
    env = pkg.PDEenv(setup, B=B, dtype=F64, y0=y)
    out, flags = env.do_step(to_dev(_mem(y), F64), to_dev(_mem(p), F64))
    for b in range(B):
        ref = fluid.do_step(cfg, y[b], p[b], K)
        assert np.abs(_jul(out)[b] - ref).max() <= 1e-11 * np.abs(ref).max()
    assert int(flags.sum()) == 0
    rng = np.random.default_rng(5)
    a0, a1 = rng.uniform(-1, 1, (B, 1, spa * spa)), rng.uniform(-1, 1, (B, 1, spa * spa))
    env.action.copy_(to_dev(a0.reshape(B, -1, 1), F64))
    env(to_dev(a1.reshape(B, -1, 1), F64))
    for b in range(B):
        pb = fluid.prepare_action(cfg, a1[b])
        yn = fluid.do_step(cfg, y[b], pb, K)
        assert np.abs(_jul(env.p)[b] - pb).max() <= 1e-12 * np.abs(pb).max()
        assert np.abs(_jul(env.y)[b] - yn).max() <= 3e-11 * np.abs(yn).max()
        r = fluid.reward_function(cfg, yn, a1[b], a1[b] - a0[b])
        assert np.abs(env.reward[b].cpu().numpy() - r).max() <= 1e-10 * max(1.0, np.abs(r).max())
        st = fluid.featurize(cfg, yn)
        assert np.abs(env.state[b].cpu().numpy().T - st).max() <= 1e-10 * max(1.0, np.abs(st).max())


@pytest.mark.slow
def test_config_c5_full_size_shard_properties(pkg):
    """configs[4] per-GPU shard at full size -- 512 x 512, B = 16, the reference's K = floor(16 nx dt) = 163 RK4 sub-steps
    of one control step (scripts/Fluid/setup/FluidSetup.jl:47) -- through size-independent properties: a single Fourier
    mode per trajectory has zero Jacobian (src/fluid_rk4.jl:145-190), so 163 sub-steps multiply it by the 163rd power of
    the degree-4 Taylor polynomial of exp(-nu k^2 h) and leave every other mode at zero; and the fused env step on random
    vortex fields stays finite with the mean vorticity (mode (0,0), where rhs = p(0,0) = 0 for zero actions) unchanged."""
    n, B = 512, 16
    setup, cfg = _pair(pkg, n, 1, spa=16, variance=0.022)
    K = setup.oversampling
    assert K == 163
    env = pkg.PDEenv(setup, B=B, dtype=F64)
    w = np.zeros((B, n, n), dtype=complex)
    modes = [(1 + 3 * b, 2 + 5 * b) for b in range(B)]
    for b, (ky, kx) in enumerate(modes):
        w[b, ky, kx] = n * n * 0.5
        w[b, -ky, -kx] = n * n * 0.5
    yn, flags = env.do_step(to_dev(_mem(w), F64), torch.zeros_like(env.y))
    got = _jul(yn)
    for b, (ky, kx) in enumerate(modes):
        x = -cfg.nu * cfg.kx2ky2[ky, kx] * (cfg.dt / K)
        poly = (1 + x + x * x / 2 + x ** 3 / 6 + x ** 4 / 24) ** K
        assert np.abs(got[b] - poly * w[b]).max() <= 1e-11 * np.abs(w[b]).max(), b
    assert int(flags.sum()) == 0
    from oracle import fluid
    rng = np.random.default_rng(9)
    base = [fluid.ic(cfg, 3, rng) for _ in range(2)]                 # 2 x 30 random vortices (src/fluid_rk4.jl:101-117)
    y = np.stack([(0.5 + 0.05 * b) * base[b % 2] for b in range(B)])
    env = pkg.PDEenv(setup, B=B, dtype=F64, y0=y)
    env(torch.zeros(env._ashape, dtype=F64, device="cuda:0"))
    assert bool(torch.isfinite(env.y).all()) and bool(torch.isfinite(env.state).all()) and bool(torch.isfinite(env.reward).all())
    m0 = _jul(env.y)[:, 0, 0]
    assert np.abs(m0 - y[:, 0, 0]).max() <= 1e-9 * max(1.0, np.abs(y).max())
    assert not bool(env.done.any())


def test_reference_trained_fluid_actor_closed_loop(pkg):
    """row F3: the actor the reference trained for Fluid_8 (hook.bestNNA 9 -> 18 -> 1; fixture from
    scripts/Fluid/Fluid_8/saves/hook.jld2 -- the hook holds no trajectory, collect_bestDF = false) driven closed-loop on this
    path at the reference's own training grid (128 x 128, 8 x 8 sensors, variance 0.08, K = floor(16 nx dt) = 40 RK4
    sub-steps per control step): three control steps of policy -> prepare_action -> do_step -> reward -> featurize from a
    random-vortex initial condition follow the oracle's loop (fp64; actions <= 1e-9, spectra <= 1e-10 relative)."""
    from oracle import fluid, nn
    from util import load_golden
    g = load_golden("fluid8_hook.npz")
    best = [g["best_W1"], g["best_b1"], g["best_W2"], g["best_b2"]]
    setup, cfg = _pair(pkg, 128, 1, spa=8, variance=0.08)
    assert setup.oversampling == 40 and setup.state_shape == (9, 64)
    y0 = fluid.ic(cfg, 3, np.random.default_rng(5))
    env = pkg.PDEenv(setup, B=1, dtype=F64, y0=y0[None])
    agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(0), dtype=torch.float32)
    assert agent.policy.behavior_actor.model.dims == [9, 18, 1]
    pkg.checkpoint.load_actor(agent.policy.behavior_actor, best)
    agent.policy.start_steps = -1
    P = [b.astype(np.float64) for b in best]
    y, a_prev = y0.copy(), np.zeros((1, 64))
    state = fluid.featurize(cfg, y)
    assert np.abs(env.state[0].cpu().numpy().T - state).max() <= 1e-11 * max(1.0, np.abs(state).max())
    for k in range(3):
        a = np.clip(nn.forward(P, [nn.RELU, nn.TANH], state), -1, 1)
        p = fluid.prepare_action(cfg, a)
        y = fluid.do_step(cfg, y, p, 40)
        r = fluid.reward_function(cfg, y, a, a - a_prev)
        state, a_prev = fluid.featurize(cfg, y), a
        act = agent.policy(env, learning=False)
        env(act)
        assert np.abs(env.action_julia() - a).max() <= 1e-9
        assert np.abs(env.y_julia() - y).max() <= 1e-10 * np.abs(y).max()
        assert np.abs(env.reward[0].cpu().numpy() - r).max() <= 1e-9 * max(1.0, np.abs(r).max())
        assert np.abs(env.state[0].cpu().numpy().T - state).max() <= 1e-9 * max(1.0, np.abs(state).max())
    assert not bool(env.done.any())


def test_temporal_stack_of_the_fluid_featurize(pkg):
    """featurize with temporal_steps = 2 (scripts/Fluid/setup/FluidSetup.jl:229-237; optional branch): 18 state rows per
    actuator, reset repeats the 3 x 3 window, a step stacks the fresh window on the previous one -- two steps vs the oracle"""
    from oracle import fluid
    n, spa, K = 32, 4, 2
    setup, cfg0 = _pair(pkg, n, 1, spa=spa, K=K, temporal_steps=2)
    import copy
    cfg = copy.copy(cfg0)
    cfg.temporal_steps = 2
    assert setup.state_shape == (18, spa * spa)
    B = 2
    rng = np.random.default_rng(8)
    y, _ = _fields(cfg, B, seed=13)
    env = pkg.PDEenv(setup, B=B, dtype=F64, y0=y)
    st = [fluid.featurize(cfg, y[b]) for b in range(B)]
    for b in range(B):
        assert st[b].shape == (18, spa * spa)
        assert np.abs(env.state[b].cpu().numpy().T - st[b]).max() <= 1e-12 * max(1.0, np.abs(st[b]).max())
    for step in range(2):
        a = rng.uniform(-1, 1, (B, 1, spa * spa))
        yj = _jul(env.y)
        env(to_dev(a.reshape(B, -1, 1), F64))
        for b in range(B):
            yn = fluid.do_step(cfg, yj[b], fluid.prepare_action(cfg, a[b]), K)
            ref = fluid.featurize(cfg, yn, st[b])
            assert np.abs(env.state[b].cpu().numpy().T - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
            assert np.array_equal(ref[9:], st[b][:9])
            st[b] = ref
