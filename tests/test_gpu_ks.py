"""GPU parity of the KS environment path against the reference's golden trajectories and the
oracle (through the C ABI).  Tolerances: fp64 <= 1e-12 abs per teacher-forced
control step vs the golden; fp32 <= 2e-5 abs (a complex64 CPU run measures 3.5e-6..7.6e-6)."""
import numpy as np
import pytest

from util import ks_pair, to_dev

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = {"f64": 1e-12, "f32": 2e-5}


@pytest.mark.parametrize("name", ["ks22", "ks200", "ks22_global"])
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_do_step_matches_reference_golden(pkg, name, prec):
    """y[t+1] = do_step(y[t], p[t+1]) for all 50 logged pairs, batched as B = 50 trajectories."""
    setup, cfg, g = ks_pair(pkg, name)
    dt = torch.float64 if prec == "f64" else torch.float32
    y, p = g["y"], g["p"]
    env = pkg.PDEenv(setup, B=50, dtype=dt)
    out, flags = env.do_step(to_dev(y[:50], dt), to_dev(p[1:51], dt))
    err = np.abs(out.cpu().numpy().astype(np.float64) - y[1:51]).max()
    assert err <= TOL[prec], err
    assert int(flags.sum()) == 0


@pytest.mark.parametrize("name", ["ks22", "ks200", "ks22_global"])
def test_env_step_fused_matches_golden_and_oracle(pkg, name):
    """Fused (env)(action): p, y, reward against the golden rows; state against the oracle."""
    from oracle import ks
    setup, cfg, g = ks_pair(pkg, name)
    dt = torch.float64
    y, p, a, r = g["y"], g["p"], g["action"], g["reward"]
    B = 49
    env = pkg.PDEenv(setup, B=B, dtype=dt)
    env.y.copy_(to_dev(y[1:50], dt))
    prev = to_dev(a[1:50], dt).reshape(env._ashape)
    env.action.copy_(prev)
    env(to_dev(a[2:51], dt).reshape(env._ashape))
    assert np.abs(env.p.cpu().numpy() - p[2:51]).max() <= 1e-12
    assert np.abs(env.y.cpu().numpy() - y[2:51]).max() <= 1e-12
    assert np.abs(env.reward.cpu().numpy() - r[2:51]).max() <= 1e-12
    for b in (0, 17, 48):
        st = ks.featurize(cfg, y[2 + b])
        got = env.state[b].cpu().numpy().T if not setup.mono else env.state[b].cpu().numpy().reshape(-1, 1)
        assert np.abs(got - st).max() <= 1e-13
    assert not bool(env.done.any())


def test_simd_sharing_form_of_the_fused_step(pkg):
    """pdec_env_set_simd_sharing: the 64-VGPR form of the fused KS step (per-mode constants, constant term and previous
    nonlinear term in LDS) against the oracle (KSSetup.jl:130-245, fp32 tolerance of the register form) and against the
    register form itself over several control steps, odd B; other environments report that they have no such form."""
    from oracle import ks
    setup = pkg.KSSetup.bench_C2(256)
    cfg = ks.KSConfig(256, setup.Lx, setup.sensor_positions, sigma_sensors=1.0, sigma_actuators=1.0, window_size=3)
    rng = np.random.default_rng(3)
    B, T = 5, 4
    y0 = setup.generate_random_init(rng, B) * 0.15
    acts = rng.uniform(-1, 1, (T + 1, B, 64))
    envs = [pkg.PDEenv(setup, B=B, dtype=torch.float32, autoreset=False) for _ in range(2)]
    assert envs[1].set_simd_sharing(True)
    for e in envs:
        e.y.copy_(to_dev(y0, torch.float32))
        e.action.copy_(to_dev(acts[0], torch.float32).reshape(e._ashape))
    yo = [y0[b].copy() for b in range(B)]
    for t in range(T):
        for e in envs:
            e(to_dev(acts[t + 1], torch.float32).reshape(e._ashape))
        for b in range(B):
            o = ks.env_step(cfg, yo[b], acts[t][b][None], acts[t + 1][b][None], 0.0)
            yo[b] = o["y"]
            assert np.abs(envs[1].y[b].cpu().numpy() - o["y"]).max() <= 2e-5 * (t + 1)
            assert np.abs(envs[1].p[b].cpu().numpy() - o["p"]).max() <= 1e-5
            assert np.abs(envs[1].reward[b].cpu().numpy() - o["reward"]).max() <= 1e-5 * (t + 1)
        for name in ("y", "p", "state", "reward"):
            a, b_ = getattr(envs[0], name), getattr(envs[1], name)
            assert float((a - b_).abs().max()) <= 2e-6 * (t + 1), name
    assert not bool(envs[1].done.any())
    assert not envs[1].set_simd_sharing(False)
    assert not pkg.PDEenv(pkg.KSSetup.KS22(), B=2, dtype=torch.float64).set_simd_sharing(True)     # fp64, N = 192: no such form


def test_fp32_env_step_and_odd_batch(pkg):
    """fp32, odd B (the last workgroup integrates a single trajectory), bench geometry C2."""
    from oracle import ks
    setup = pkg.KSSetup.bench_C2(256)
    cfg = ks.KSConfig(256, setup.Lx, setup.sensor_positions, sigma_sensors=1.0, sigma_actuators=1.0, window_size=3)
    rng = np.random.default_rng(0)
    B = 5
    y0 = setup.generate_random_init(rng, B) * 0.15
    act_prev = rng.uniform(-1, 1, (B, 64))
    act = rng.uniform(-1, 1, (B, 64))
    env = pkg.PDEenv(setup, B=B, dtype=torch.float32)
    env.y.copy_(to_dev(y0, torch.float32))
    env.action.copy_(to_dev(act_prev, torch.float32).reshape(env._ashape))
    env(to_dev(act, torch.float32).reshape(env._ashape))
    for b in range(B):
        o = ks.env_step(cfg, y0[b], act_prev[b][None], act[b][None], 0.0)
        assert np.abs(env.y[b].cpu().numpy() - o["y"]).max() <= 2e-5
        assert np.abs(env.p[b].cpu().numpy() - o["p"]).max() <= 1e-5
        assert np.abs(env.reward[b].cpu().numpy() - o["reward"]).max() <= 1e-5
        assert np.abs(env.state[b].cpu().numpy().T - o["state"]).max() <= 1e-5


def test_standalone_closures_match_oracle(pkg):
    from oracle import ks
    setup, cfg, g = ks_pair(pkg, "ks200")
    dt = torch.float64
    env = pkg.PDEenv(setup, B=3, dtype=dt)
    y, a = g["y"][5:8], g["action"]
    st = env.featurize(to_dev(y, dt)).cpu().numpy()
    pp = env.prepare_action(to_dev(a[5:8], dt).reshape(env._ashape)).cpu().numpy()
    rr = env.reward_function(to_dev(y, dt), to_dev(a[5:8], dt).reshape(env._ashape),
                             to_dev(a[4:7], dt).reshape(env._ashape)).cpu().numpy()
    for b in range(3):
        assert np.abs(st[b].T - ks.featurize(cfg, y[b])).max() <= 1e-13
        assert np.abs(pp[b] - ks.prepare_action(cfg, a[5 + b][None])).max() <= 1e-12
        assert np.abs(rr[b] - ks.reward_function(cfg, y[b], a[5 + b][None], (a[5 + b] - a[4 + b])[None])).max() <= 1e-13


def test_mass_conservation_and_blowup_flag(pkg):
    """Size-independent properties at the bench size: with p = 0 the CNAB2 step conserves
    sum(y) (mode 0 has L = G = 0); a state beyond max_value raises the done flag."""
    setup = pkg.KSSetup.bench_C2(256)
    B = 512
    rng = np.random.default_rng(1)
    y0 = setup.generate_random_init(rng, B) * 0.1
    env = pkg.PDEenv(setup, B=B, dtype=torch.float64)
    yd = to_dev(y0, torch.float64)
    out, flags = env.do_step(yd, torch.zeros_like(yd))
    assert torch.allclose(out.sum(1), yd.sum(1), atol=1e-10)
    assert int(flags.sum()) == 0
    yd[3] += 40.0   # mean mode is conserved, so |y| stays > max_value = 30 but finite
    out, flags = env.do_step(yd, torch.zeros_like(yd))
    f = flags.cpu().numpy()
    assert f[3] == 1 and f.sum() == 1


def test_disturbance_term(pkg):
    """mu != 0 (KS200_disturbed.jl:16): the inhomogeneous term added outside A_inv (KSSetup.jl:155)."""
    from oracle import ks
    g_setup, cfg, g = ks_pair(pkg, "ks200")
    pos = np.arange(1, 241, 3)
    setup = pkg.KSSetup(240, 200.0, pos, mu=0.02)
    cfg = ks.KSConfig(240, 200.0, pos, mu=0.02)
    env = pkg.PDEenv(setup, B=2, dtype=torch.float64)
    y, p = g["y"][10:12], g["p"][11:13]
    out, _ = env.do_step(to_dev(y, torch.float64), to_dev(p, torch.float64))
    for b in range(2):
        assert np.abs(out[b].cpu().numpy() - ks.do_step(cfg, y[b], p[b])).max() <= 1e-12


def test_host_pointer_wrapper(pkg):
    """pdec_pde_step_host: the form a Julia do_step(env) closure binds (host arrays in/out)."""
    import ctypes as C
    setup, cfg, g = ks_pair(pkg, "ks22")
    env = pkg.PDEenv(setup, B=1, dtype=torch.float64)
    y = np.ascontiguousarray(g["y"][3]); p = np.ascontiguousarray(g["p"][4])
    out = np.empty_like(y); done = np.zeros(1, dtype=np.int32)
    lib = pkg._lib.load()
    pkg._lib.check(lib.pdec_pde_step_host(env.handle, y.ctypes.data, p.ctypes.data, out.ctypes.data, done.ctypes.data))
    assert np.abs(out - g["y"][4]).max() <= 1e-12 and done[0] == 0


@pytest.mark.parametrize("nx", [240, 256])
@pytest.mark.parametrize("integ", ["rk4_fd", "midpoint_fd"])
@pytest.mark.parametrize("prec,tol", [("f64", 1e-11), ("f32", 2e-4)])
def test_rk4_fd_variant_matches_its_oracle(pkg, prec, tol, integ, nx, monkeypatch):
    """north-star variant: RK4 + periodic 5-point FD (stencils of KSSetup.jl:55-59); rhs, do_step and the fused
    (env)(action) against oracle/ks.py rhs_fd / do_step_rk4_fd (relative to max|value|).  nx = 256: the fused step runs in
    its one-wave-per-trajectory form (four cells per lane, neighbours by lane exchange, csrc/env.hip: ksfd_wave_step_kernel);
    it must also agree with the general one-cell-per-thread form (PDEC_KSFD_LDS=1) to round-off."""
    from oracle import ks
    Lx, K, dtc = 200.0, 30, 0.1
    pos = np.arange(1, nx + 1, 3)
    setup = pkg.KSSetup(nx, Lx, pos, integrator=integ, mu=0.02, dt=dtc, oversampling=K, window_size=3)
    step = ks.do_step_rk4_fd if integ == "rk4_fd" else ks.do_step_midpoint_fd     # midpoint: src/PDEenv.jl:208-214
    cfg = ks.KSConfig(nx, Lx, pos, mu=0.02, dt=dtc, oversampling=K, window_size=3)
    dt = torch.float64 if prec == "f64" else torch.float32
    rng = np.random.default_rng(4)
    B = 7
    y = setup.generate_random_init(rng, B) * 0.1
    a0 = rng.uniform(-1, 1, (B, len(pos)))
    a1 = rng.uniform(-1, 1, (B, len(pos)))
    env = pkg.PDEenv(setup, B=B, dtype=dt, y0=y)
    p = np.stack([ks.prepare_action(cfg, a1[b][None]) for b in range(B)])
    f = env.rhs(to_dev(y, dt), to_dev(p, dt)).cpu().numpy()
    for b in range(B):
        ref = ks.rhs_fd(cfg, y[b], p[b])
        assert np.abs(f[b] - ref).max() <= tol * np.abs(ref).max()
    out, flags = env.do_step(to_dev(y, dt), to_dev(p, dt))
    env.action.copy_(to_dev(a0, dt).reshape(env._ashape))
    env(to_dev(a1, dt).reshape(env._ashape))
    for b in range(B):
        ref = step(cfg, y[b], p[b])
        assert np.abs(out[b].cpu().numpy() - ref).max() <= tol * np.abs(ref).max()
        assert np.abs(env.y[b].cpu().numpy() - ref).max() <= tol * np.abs(ref).max()
        st = ks.featurize(cfg, ref)
        assert np.abs(env.state[b].cpu().numpy().T - st).max() <= 10 * tol
        r = ks.reward_function(cfg, ref, a1[b][None], (a1[b] - a0[b])[None])
        assert np.abs(env.reward[b].cpu().numpy() - r).max() <= 10 * tol
    assert int(flags.sum()) == 0
    if nx == 256:
        y_wave, st_wave, r_wave = env.y.clone(), env.state.clone(), env.reward.clone()
        monkeypatch.setenv("PDEC_KSFD_LDS", "1")
        env2 = pkg.PDEenv(setup, B=B, dtype=dt, y0=y)
        env2.action.copy_(to_dev(a0, dt).reshape(env2._ashape))
        env2(to_dev(a1, dt).reshape(env2._ashape))
        rt = 1e-13 if prec == "f64" else 1e-5
        assert float((env2.y - y_wave).abs().max()) <= rt * float(y_wave.abs().max())
        assert float((env2.state - st_wave).abs().max()) <= 10 * rt and float((env2.reward - r_wave).abs().max()) <= 10 * rt


@pytest.mark.parametrize("nx", [1024, 4096, 600, 250, 60])
@pytest.mark.parametrize("prec,tol", [("f64", 1e-11), ("f32", 3e-5)])
def test_every_fft_engine_and_size_limit(pkg, nx, prec, tol):
    """the fused step at the other engine sizes: 1024 (config C3: four waves, one cross-wave radix-4 stage + the single-wave 256-point engine), 4096 / 2048 (the largest fp32 / fp64 N the
    in-LDS kernel accepts), 600 (KS500's grid: factors 2,3,5), 250 (2 * 5^3), 60 (fewer points than a wave has lanes; the reference's kernels need nx >= 50); odd batch"""
    from oracle import ks
    if nx == 4096 and prec == "f64":
        nx = 2048       # complex fp64: the two 64 KiB transform buffers of 4096 points would exceed the 160 KiB LDS
    setup = pkg.KSSetup.bench_C2(nx)
    if nx >= 2048:      # sensors every 16 cells: the sensing scratch of 1024 sensors would not fit beside the 4096-point FFT
        setup = pkg.KSSetup(nx, setup.Lx, np.arange(1, nx + 1, 16), sigma_sensors=1.0, sigma_actuators=1.0, window_size=3)
    cfg = ks.KSConfig(nx, setup.Lx, setup.sensor_positions, sigma_sensors=1.0, sigma_actuators=1.0, window_size=3)
    dt = torch.float64 if prec == "f64" else torch.float32
    rng = np.random.default_rng(nx)
    B, A = 3, setup.n_actuators
    y0 = setup.generate_random_init(rng, B) * 0.15
    act_prev, act = rng.uniform(-1, 1, (B, A)), rng.uniform(-1, 1, (B, A))
    env = pkg.PDEenv(setup, B=B, dtype=dt)
    env.y.copy_(to_dev(y0, dt))
    env.action.copy_(to_dev(act_prev, dt).reshape(env._ashape))
    env(to_dev(act, dt).reshape(env._ashape))
    for b in range(B):
        o = ks.env_step(cfg, y0[b], act_prev[b][None], act[b][None], 0.0)
        assert np.abs(env.y[b].cpu().numpy() - o["y"]).max() <= tol * max(1.0, np.abs(o["y"]).max())
        assert np.abs(env.reward[b].cpu().numpy() - o["reward"]).max() <= tol * 10
        assert np.abs(env.state[b].cpu().numpy().T - o["state"]).max() <= tol * 10
    assert int(env.done.sum()) == 0


def test_size_limits_are_reported_not_crashed(pkg):
    """beyond the limits the create call fails with a message (no launch with shapes the kernels do not cover)"""
    for nx in (8192, 254 * 7):          # too large for LDS; a prime factor other than 2, 3, 5
        with pytest.raises(pkg.PdecError):
            pkg.PDEenv(pkg.KSSetup.bench_C2(nx), B=1, dtype=torch.float32)


def test_config_c3_per_gpu_shard_full_size(pkg):
    """BASELINE.json configs[2], one GPU's shard at FULL size: KS N = 1024 (Lx = 853.33), A = 256 actuators, B = 512
    trajectories, fp32, fused (env)(action).  Sampled trajectories (first / odd / last of the batch, both members of a
    packed pair) against the fp64 oracle at the fp32 tolerance; all 512: finite, no blow-up flag, and with zero
    forcing the CNAB2 step conserves sum(y) (mode 0 has L = G = 0) -- the size-independent property."""
    from oracle import ks
    nx, B = 1024, 512
    setup = pkg.KSSetup.bench_C2(nx)
    cfg = ks.KSConfig(nx, setup.Lx, setup.sensor_positions, sigma_sensors=1.0, sigma_actuators=1.0, window_size=3)
    A = setup.n_actuators
    assert A == 256 and abs(setup.Lx - 853.3333333333334) < 1e-9
    rng = np.random.default_rng(3)
    y0 = setup.generate_random_init(rng, B) * 0.15
    act_prev, act = rng.uniform(-1, 1, (B, A)), rng.uniform(-1, 1, (B, A))
    dt = torch.float32
    env = pkg.PDEenv(setup, B=B, dtype=dt)
    env.y.copy_(to_dev(y0, dt))
    env.action.copy_(to_dev(act_prev, dt).reshape(env._ashape))
    env(to_dev(act, dt).reshape(env._ashape))
    assert bool(torch.isfinite(env.y).all()) and bool(torch.isfinite(env.state).all()) and int(env.done.sum()) == 0
    for b in (0, 1, 255, 510, 511):
        o = ks.env_step(cfg, y0[b], act_prev[b][None], act[b][None], 0.0)
        assert np.abs(env.y[b].cpu().numpy() - o["y"]).max() <= 3e-5 * max(1.0, np.abs(o["y"]).max())
        assert np.abs(env.p[b].cpu().numpy() - o["p"]).max() <= 1e-5 * max(1.0, np.abs(o["p"]).max())
        assert np.abs(env.reward[b].cpu().numpy() - o["reward"]).max() <= 3e-4
        assert np.abs(env.state[b].cpu().numpy().T - o["state"]).max() <= 3e-4
    yd = to_dev(y0, dt)
    out, flags = env.do_step(yd, torch.zeros_like(yd))
    s0, s1 = yd.double().sum(1), out.double().sum(1)
    assert float((s1 - s0).abs().max()) <= 2e-3 * float(yd.abs().sum(1).max()) / nx * 30      # fp32 sums of 1024 cells
    assert int(flags.sum()) == 0


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_temporal_stack_of_the_ks_featurize(pkg, prec):
    """featurize with temporal_steps > 1 (scripts/KS/setup/KSSetup.jl:209-218; optional branch, 1 in the shipped scripts):
    reset repeats the fresh window rows, every step stacks the fresh rows on the newest rows of the previous state.
    Three control steps, temporal_steps = 3, window 3, odd B, against the oracle (teacher-forced on y)."""
    from oracle import ks
    dt = torch.float64 if prec == "f64" else torch.float32
    tol = 1e-12 if prec == "f64" else 2e-5
    setup = pkg.KSSetup.KS22(window_size=3, temporal_steps=3)
    cfg = ks.KSConfig(192, 22.0, np.arange(1, 193, 24), sigma_sensors=0.7, sigma_actuators=0.7, window_size=3, temporal_steps=3)
    assert setup.state_shape == (9, 8)
    rng = np.random.default_rng(5)
    B = 3
    y0 = setup.generate_random_init(rng, B) * 0.2
    env = pkg.PDEenv(setup, B=B, dtype=dt, y0=y0, autoreset=False)
    st = [ks.featurize(cfg, y0[b]) for b in range(B)]                       # reset: isnothing(env) branch
    for b in range(B):
        assert st[b].shape == (9, 8) and np.abs(env.state[b].cpu().numpy().T - st[b]).max() <= tol
    a_prev = np.zeros((B, 1, 8))
    for step in range(3):
        a = rng.uniform(-1, 1, (B, 1, 8))
        y = env.y.cpu().numpy().astype(np.float64)
        env(to_dev(a.reshape(B, 8, 1), dt).reshape(env._ashape))
        for b in range(B):
            o = ks.env_step(cfg, y[b], a_prev[b], a[b], 0.0, prev_state=st[b])
            assert np.abs(env.y[b].cpu().numpy() - o["y"]).max() <= tol
            assert np.abs(env.state[b].cpu().numpy().T - o["state"]).max() <= tol
            assert np.array_equal(o["state"][3:], st[b][:6])                  # the stack really shifts
            st[b] = o["state"]
        a_prev = a
