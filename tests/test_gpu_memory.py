"""Action memory (`memory_size > 0`): the reference's optional branch in which the actor has 1 + memory_size outputs per
actuator, row 1 drives the PDE and the reward, and the other rows come back as the last rows of the next state
(scripts/KS/setup/KSSetup.jl:39,48,216-226, scripts/Keller-Segel/setup/KellerSegelSetup.jl:303-313, src/PDEagent.jl:201).
No shipped script sets it, so there is no reference artifact; the oracle restates the branch line by line and the GPU path
(the composed env step, csrc/env.hip env_step_composed; pdec_mlp_set_noise_rows) is compared with it here."""
import numpy as np
import pytest

from util import to_dev

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F64 = torch.float64


def _col(t):      # device [A, rows] of one trajectory -> Julia [rows, A]
    return t.cpu().numpy().astype(np.float64).T


@pytest.mark.parametrize("T", [1, 2])
@pytest.mark.parametrize("m", [1, 3])
def test_ks_closed_loop_with_action_memory_matches_oracle(pkg, T, m):
    """KS (CNAB2), window 3, temporal stack T, memory m: reset form (zeros in the memory rows), then four control steps with
    given actions: forcing, field, reward (row 1 of the action only) and state (memory rows = rows 2.. of the action, the
    temporal block dropping the previous state's memory rows) per trajectory, fp64 <= 1e-12; terminal flags follow the blow-up"""
    from oracle import ks
    nx, B, steps = 256, 3, 4
    setup = pkg.KSSetup.bench_C2(nx, temporal_steps=T, memory_size=m)
    cfg = ks.KSConfig(nx, setup.Lx, setup.sensor_positions, sigma_sensors=1.0, sigma_actuators=1.0, window_size=3, temporal_steps=T)
    cfg.memory_size = m
    A, na = setup.n_actuators, 1 + m
    assert setup.state_shape == (3 * T + m, A) and setup.action_shape == (na, A)
    rng = np.random.default_rng(11)
    y0 = setup.generate_random_init(rng, B) * 0.2
    env = pkg.PDEenv(setup, B=B, dtype=F64, y0=y0, autoreset=False)
    assert tuple(env.state.shape) == (B, A, 3 * T + m) and tuple(env.action.shape) == (B, A, na)
    state = []
    for b in range(B):
        st = ks.featurize(cfg, y0[b], None, None)
        assert st.shape == (3 * T + m, A) and np.all(st[-m:] == 0)
        assert np.abs(_col(env.state[b]) - st).max() <= 1e-13
        state.append(st)
    yo = [y0[b].copy() for b in range(B)]
    aprev = [np.zeros((na, A)) for _ in range(B)]
    for t in range(steps):
        act = rng.uniform(-1, 1, (B, na, A))
        env(to_dev(np.swapaxes(act, 1, 2), F64))
        for b in range(B):
            o = ks.env_step(cfg, yo[b], aprev[b], act[b], 0.0, prev_state=state[b])
            assert np.abs(env.p[b].cpu().numpy() - o["p"]).max() <= 1e-12
            assert np.abs(env.y[b].cpu().numpy() - o["y"]).max() <= 1e-12
            assert np.abs(env.reward[b].cpu().numpy() - o["reward"]).max() <= 1e-12
            assert np.abs(_col(env.state[b]) - o["state"]).max() <= 1e-12
            assert np.array_equal(o["state"][-m:], act[b][1:])
            yo[b], state[b], aprev[b] = o["y"], o["state"], act[b]
    assert not bool(env.done.any())
    # the stand-alone featurize closure with and without the action
    got = env.featurize(env.y, env.prev_state, env.action)
    assert torch.equal(got, env.state)
    assert bool((env.featurize(env.y, env.prev_state)[:, :, -m:] == 0).all())


def test_ks_fp32_and_blow_up_flags_with_action_memory(pkg):
    """fp32 (2e-5) and the blow-up flag of the composed step: a trajectory that leaves |y| <= max_value raises `done` (the last
    one of an odd batch: the CNAB2 kernel packs two trajectories into one complex sequence, so an overflow to inf reaches the
    partner too -- with or without action memory)"""
    from oracle import ks
    nx, B, m = 256, 3, 2
    setup = pkg.KSSetup.bench_C2(nx, memory_size=m)
    cfg = ks.KSConfig(nx, setup.Lx, setup.sensor_positions, sigma_sensors=1.0, sigma_actuators=1.0, window_size=3)
    cfg.memory_size = m
    rng = np.random.default_rng(12)
    y0 = setup.generate_random_init(rng, B) * 0.2
    y0[2] *= 400.0                                    # far beyond max_value = 30
    env = pkg.PDEenv(setup, B=B, dtype=torch.float32, y0=y0, autoreset=False)
    act = rng.uniform(-1, 1, (B, 1 + m, setup.n_actuators))
    env(to_dev(np.swapaxes(act, 1, 2), torch.float32))
    done = env.done.cpu().numpy()
    assert done.tolist() == [False, False, True]
    for b in (0, 1):
        o = ks.env_step(cfg, y0[b], np.zeros_like(act[b]), act[b], 0.0, prev_state=None)
        assert np.abs(env.y[b].cpu().numpy() - o["y"]).max() <= 2e-5
        assert np.abs(_col(env.state[b]) - o["state"]).max() <= 2e-5
        assert np.abs(env.reward[b].cpu().numpy() - o["reward"]).max() <= 2e-5


def test_keller_segel_env_step_with_action_memory_matches_oracle(pkg):
    """1-D Keller-Segel (window 3 x 2 species x 2 temporal steps + memory 2): the same composed step on the RK4 kernel"""
    from oracle import keller_segel as kg
    m, B = 2, 3
    setup, cfg = pkg.KellerSegelSetup(memory_size=m), kg.KSegConfig()
    cfg.memory_size = m
    A, na = setup.n_actuators, 1 + m
    assert setup.state_shape == (12 + m, A)
    rng = np.random.default_rng(13)
    y0 = np.stack([setup.y0_standard() * (1.0 + 0.02 * b) for b in range(B)])          # [B, 2, nx]
    env = pkg.PDEenv(setup, B=B, dtype=F64, y0=np.ascontiguousarray(np.swapaxes(y0, 1, 2)), autoreset=False)
    state = [kg.featurize(cfg, y0[b], None, None) for b in range(B)]
    for b in range(B):
        assert np.abs(_col(env.state[b]) - state[b]).max() <= 1e-13
    yo, aprev = [y0[b] for b in range(B)], [np.zeros((na, A)) for _ in range(B)]
    for t in range(3):
        act = rng.uniform(-1, 1, (B, na, A))
        env(to_dev(np.swapaxes(act, 1, 2), F64))
        for b in range(B):
            p = kg.prepare_action(cfg, act[b])
            yn = kg.do_step(cfg, yo[b], p)
            r = kg.reward_function(cfg, yn, act[b], act[b] - aprev[b])
            st = kg.featurize(cfg, yn, state[b], act[b])
            assert np.abs(np.swapaxes(env.y[b].cpu().numpy(), 0, 1) - yn).max() <= 1e-11
            assert np.abs(env.reward[b].cpu().numpy() - r).max() <= 1e-11
            assert np.abs(_col(env.state[b]) - st).max() <= 1e-11
            yo[b], state[b], aprev[b] = yn, st, act[b]


@pytest.mark.parametrize("n,B,T", [(32, 3, 1), (32, 3, 2), (512, 8, 1)])
def test_fluid_env_step_with_action_memory_matches_oracle(pkg, n, B, T):
    """fluid (FluidSetup.jl:229-244): 3 x 3 window x temporal stack + 2 memory rows; two control steps against the oracle's
    closures on a 32 x 32 grid; on the 512 x 512 grid (B = 8: two half-batch children, the action rows sliced per child) the
    state is checked through its structure -- memory rows = rows 2.. of the action, window rows = the memory-free environment's"""
    from oracle import fluid
    m = 2
    if n == 32:
        spa, K = 4, 2
        setup = pkg.FluidSetup(nx=n, ifpad=1, sensors_per_axis=spa, variance=0.08, oversampling=K, temporal_steps=T, memory_size=m)
        cfg = fluid.FluidConfig(nx=n, ifpad=1, sensors_per_axis=spa, variance=0.08, oversampling=K)
        cfg.temporal_steps, cfg.memory_size = T, m
        rng = np.random.default_rng(21)
        y = np.stack([fluid.ic(cfg, 4, rng) for _ in range(B)])
        A, na = spa * spa, 1 + m
        assert setup.state_shape == (9 * T + m, A) and setup.action_shape == (na, A)
        ymem = np.ascontiguousarray(np.stack([np.stack([y[b].T.real, y[b].T.imag], axis=-1) for b in range(B)]))
        env = pkg.PDEenv(setup, B=B, dtype=F64, y0=ymem, autoreset=False)
        state = [fluid.featurize(cfg, y[b], None, None) for b in range(B)]
        for b in range(B):
            assert np.abs(_col(env.state[b]) - state[b]).max() <= 1e-12 and not state[b][-m:].any()
        yo, aprev = [y[b] for b in range(B)], [np.zeros((na, A)) for _ in range(B)]
        for t in range(2):
            act = rng.uniform(-1, 1, (B, na, A))
            env(to_dev(np.swapaxes(act, 1, 2), F64))
            for b in range(B):
                p = fluid.prepare_action(cfg, act[b])
                yn = fluid.do_step(cfg, yo[b], p, K)
                r = fluid.reward_function(cfg, yn, act[b], act[b] - aprev[b])
                st = fluid.featurize(cfg, yn, state[b], act[b])
                got = env.y[b].cpu().numpy()
                assert np.abs((got[..., 0] + 1j * got[..., 1]).T - yn).max() <= 1e-11 * np.abs(yn).max()
                assert np.abs(env.reward[b].cpu().numpy() - r).max() <= 1e-11 * max(1.0, np.abs(r).max())
                assert np.abs(_col(env.state[b]) - st).max() <= 1e-11 * max(1.0, np.abs(st).max())
                yo[b], state[b], aprev[b] = yn, st, act[b]
        return
    setup_m = pkg.FluidSetup(nx=n, sensors_per_axis=16, variance=0.04, memory_size=m, oversampling=3)
    setup_0 = pkg.FluidSetup(nx=n, sensors_per_axis=16, variance=0.04, oversampling=3)
    em = pkg.PDEenv(setup_m, B=B, dtype=F64, autoreset=False)
    e0 = pkg.PDEenv(setup_0, B=B, dtype=F64, autoreset=False)
    assert em.n_part_streams == 1
    y0 = setup_0.random_init_device(e0, np.random.default_rng(4))
    em.set_y0(y0); e0.set_y0(y0)
    A = setup_m.n_actuators
    act = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (B, A, 1 + m))).cuda()
    em(act)
    e0(act[..., :1].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(em.y, e0.y) and torch.equal(em.reward, e0.reward)
    assert torch.equal(em.state[..., :9], e0.state) and torch.equal(em.state[..., 9:], act[..., 1:])


def test_policy_noise_spares_the_memory_rows(pkg):
    """(policy)(env) with memory_size = 2 (src/PDEagent.jl:201): noise on row 1 only, clamp on every row; against the oracle
    with the device's own Philox draws, and the rows without noise equal the actor's clamped outputs exactly"""
    from oracle import nn
    m = 2
    setup = pkg.KSSetup.bench_C2(256, memory_size=m, act_noise=0.4)
    B = 2
    env = pkg.PDEenv(setup, B=B, dtype=F64, autoreset=False)
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(5), dtype=torch.float32, start_steps=-1, noise_seed=77)
    pol = agent.policy
    assert pol.memory_size == m and pol.behavior_actor.model.dims[-1] == 1 + m
    assert pol.behavior_critic.model.dims[0] == setup.state_shape[0] + 1 + m
    a = pol(env, learning=True).clone()                                   # [B, A, 1 + m]
    quiet = pol(env, learning=False).clone()
    A_, acts = pol.behavior_actor.model.params(), pol.behavior_actor.model.acts
    cols = B * setup.n_actuators
    st = env.state.reshape(cols, -1).cpu().numpy().T
    noise = torch.empty((cols, 1 + m), dtype=F64, device="cuda:0")
    pkg._lib.check(pol.lib.pdec_randn(env.handle, pkg._lib.ptr(noise), noise.numel(), pkg._lib.PDEC_F64, 77, 0))
    want = nn.policy_act(A_, acts, st, noise.cpu().numpy().T, 0.4, setup.act_limit, learning=True, memory_size=m)
    got = a.reshape(cols, 1 + m).cpu().numpy().T
    assert np.abs(got - want).max() <= 1e-12
    assert torch.equal(a[..., 1:], quiet[..., 1:]) and not torch.equal(a[..., 0], quiet[..., 0])
    # the mask is a property of the network object (set at agent creation), so callers that enqueue the acting kernel
    # themselves -- TrainPipeline: pdec_policy_act_rng_dev on the behaviour actor -- get it as well
    m32 = pol.behavior_actor.model
    s32 = env.state.reshape(cols, -1).float().contiguous()
    out = [torch.empty((cols, 1 + m), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    for o, learning in zip(out, (1, 0)):
        pkg._lib.check(pol.lib.pdec_policy_act_rng_dev(m32.handle, pkg._lib.ptr(s32), cols, 0.4, float(setup.act_limit), learning, 123,
                                                       pkg._lib.ptr(o)))
    torch.cuda.synchronize()
    assert torch.equal(out[0][:, 1:], out[1][:, 1:]) and not torch.equal(out[0][:, 0], out[1][:, 0])


def test_ddpg_update_with_memory_shaped_nets_matches_oracle(pkg):
    """one DDPG update with na = 3 actor outputs / ns + 3 critic inputs (the generic passes): the four networks after ADAM and
    Polyak against the oracle (src/PDEagent.jl:363-418)"""
    from oracle import nn
    m = 2
    setup = pkg.KSSetup.bench_C2(256, memory_size=m)
    ns, na = setup.state_shape[0], 1 + m
    agent = pkg.create_agent(setup=setup, B=2, rng=np.random.default_rng(6), dtype=torch.float32, start_steps=-1)
    pol = agent.policy
    nets = [pol.behavior_actor, pol.behavior_critic, pol.target_actor, pol.target_critic]
    P = [[w.copy() for w in n.model.params()] for n in nets]
    acts_a, acts_c = nets[0].model.acts, nets[1].model.acts
    rng = np.random.default_rng(7)
    Bu = 96
    batch = dict(state=rng.standard_normal((Bu, ns)), action=rng.uniform(-1, 1, (Bu, na)), reward=rng.standard_normal(Bu),
                 terminal=(rng.uniform(size=Bu) < 0.1).astype(np.float64), next_state=rng.standard_normal((Bu, ns)))
    pol.update({k: to_dev(v, torch.float32) for k, v in batch.items()})
    torch.cuda.synchronize()
    optA = nn.Adam(P[0], nets[0].optimizer.eta)
    optC = nn.Adam(P[1], nets[1].optimizer.eta)
    f32 = np.float32
    nn.ddpg_update(P[0], P[1], P[2], P[3], optA, optC, acts_a, acts_c, batch["state"].T.astype(f32), batch["action"].T.astype(f32),
                   batch["reward"].astype(f32), batch["terminal"].astype(f32), batch["next_state"].T.astype(f32),
                   f32(pol.y), f32(pol.rho_effective), quirk=pol.quirk)
    for n, want in zip(nets, P):
        for w, ref in zip(n.model.params(), want):
            assert np.abs(w - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


def test_memory_is_refused_where_it_is_not_built(pkg):
    for mk in (lambda: pkg.KSSetup.bench_C2(256, memory_size=1, mono=True),
               lambda: pkg.KellerSegel2DSetup(nx=32, ny=32, memory_size=1),
               lambda: pkg.KSSetup.bench_C2(256, memory_size=1, check_max_value="reward")):
        with pytest.raises(pkg.PdecError, match="memory_size"):
            mk()


@pytest.mark.parametrize("which,B", [("ks22", 1), ("ks22", 3), ("kseg", 1)])
def test_run_loop_with_action_memory(pkg, which, B):
    """RL.jl's run loop end to end with memory_size = 2 (start policy, replay pushes of [1 + m]-wide actions, sampled DDPG
    updates on the memory-shaped nets, hook rows): finite returns, the actor trained, trace widths and the hook's action rows
    carry the memory rows; a rollout of the trained actor (step loop: the one-launch rollouts serve single-output actors) agrees
    with stepping the environment by hand"""
    m = 2
    setup = (pkg.KSSetup.KS22(te=1.0, update_loops=3, start_steps=2, update_after=2, memory_size=m, window_size=3) if which == "ks22"
             else pkg.KellerSegelSetup(te=0.06, update_loops=3, start_steps=2, update_after=1, memory_size=m))
    env = pkg.PDEenv(setup, B=B, dtype=F64)
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), trajectory_length=2000)
    hook = pkg.PDEhook(min_best_episode=1, use_random_init=True)
    before = [w.copy() for w in agent.policy.behavior_actor.params()]
    pkg.run(agent, env, pkg.StopAfterEpisodeWithMinSteps(30), hook)
    torch.cuda.synchronize()
    assert len(hook.rewards) >= 2 and np.isfinite(hook.rewards).all()
    tr = agent.trajectory
    A = setup.n_actuators
    assert tr.action.shape[1] == 1 + m and tr.state.shape[1] == setup.state_shape[0]
    assert any(np.abs(a - b).max() > 0 for a, b in zip(agent.policy.behavior_actor.params(), before))
    assert np.isfinite(agent.policy.losses()).all()
    row = hook.bestDF[-1]
    assert np.asarray(row["action"]).size == (1 + m) * A
    # the stored transitions: the memory rows of state t + 1 are rows 2.. of action t (per actuator column)
    n = min(tr.n_rt, 5 * A * B)
    s, a = tr.state[:n + A * B].cpu().numpy(), tr.action[:n].cpu().numpy()
    stride = A * B
    ok = 0
    for i in range(n - stride):
        if float(tr.terminal[i]) == 0.0 and np.array_equal(s[i + stride][-m:], a[i][1:]):
            ok += 1
    assert ok >= (n - stride) // 2          # (episode boundaries and resets break the chain; every interior row keeps it)
    # rollout of the trained actor (greedy) == stepping by hand
    env2 = pkg.PDEenv(setup, B=B, dtype=F64, autoreset=False)
    env3 = pkg.PDEenv(setup, B=B, dtype=F64, autoreset=False)
    actor = agent.policy._actor_for(F64, B * A)
    agent.policy.start_steps = -1            # (the run's end reset the step counter: no start policy for the hand loop)
    out = env2.rollout(actor, 3, act_limit=setup.act_limit, learning=False, log=True)
    for t in range(3):
        a = agent.policy(env3, learning=False)
        env3(a)
        assert torch.equal(out["action"][t], env3.action) and torch.equal(out["y"][t], env3.y)
    assert torch.equal(env2.state, env3.state) and torch.equal(env2.y, env3.y)
