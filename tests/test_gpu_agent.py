"""GPU tests of the host mirror driving the HIP path: policy act, fused update through
CustomDDPGPolicy, and the RL.jl-style run loop with PDEhook (B = 1 like the reference, and a
batched run)."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_policy_update_matches_oracle_and_losses(pkg):
    from oracle import nn
    setup = pkg.KSSetup.bench_C2(256)
    B = 4
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(3), quirk_frozen_targets=False)   # Polyak as written
    pol = agent.policy
    assert pol.rho_effective == 0.995
    rng = np.random.default_rng(4)
    Bu = 256
    batch_np = dict(state=rng.standard_normal((Bu, 3)), action=rng.uniform(-1, 1, (Bu, 1)), reward=-rng.uniform(0, 1, Bu),
                    terminal=(rng.uniform(0, 1, Bu) < 0.1) * 1.0, next_state=rng.standard_normal((Bu, 3)))
    batch = {k: torch.as_tensor(v, dtype=torch.float32, device="cuda:0") for k, v in batch_np.items()}
    P = [[p.astype(np.float32) for p in m.params()] for m in (pol.behavior_actor.model, pol.behavior_critic.model,
                                                                pol.target_actor.model, pol.target_critic.model)]
    for a, b in zip(P[0], P[2]):
        assert np.array_equal(a, b)                      # create_agent force-syncs targets (PDEagent.jl:76-77)
    da, aa = nn.layer_sizes(3, 1, 1.6, True, False)
    dc, ac = nn.layer_sizes(3, 1, 7.0, False, False)
    optA, optC = nn.Adam(P[0], 5e-4), nn.Adam(P[1], 1e-3)
    f32 = lambda k: batch_np[k].astype(np.float32)
    out = nn.ddpg_update(P[0], P[1], P[2], P[3], optA, optC, aa, ac, f32("state").T, f32("action").T, f32("reward"),
                         f32("terminal"), f32("next_state").T, np.float32(0.99), np.float32(0.995), True)
    pol.update(batch)
    al, cl = pol.losses()
    assert abs(al - out["actor_loss"]) <= 2e-4 and abs(cl - out["critic_loss"]) <= 2e-4
    for m, Pr in zip((pol.behavior_actor.model, pol.behavior_critic.model, pol.target_actor.model, pol.target_critic.model), P):
        for x, y in zip(m.params(), Pr):
            assert np.abs(x - y).max() <= 2e-4 * max(1.0, np.abs(y).max())


def test_acting_path_is_fp64_with_fp32_weights(pkg):
    """src/PDEagent.jl:184-189: Float64 state promotes the Float32 weights"""
    from oracle import nn
    setup = pkg.KSSetup.KS22()
    env = pkg.PDEenv(setup, B=1, dtype=torch.float64)
    agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(0), start_steps=-1)
    a = agent.policy(env, learning=False)
    P = agent.policy.behavior_actor.model.params()
    _, acts = nn.layer_sizes(1, 1, 0.6, True, True)
    ref = nn.policy_act(P, acts, env.state_julia(), None, 0.0, 1.0, learning=False)
    assert a.dtype == torch.float64 and np.abs(a.cpu().numpy().reshape(-1) - ref.reshape(-1)).max() <= 1e-13


@pytest.mark.parametrize("which", ["ks22", "ks200", "keller_segel", "ks22_three_layer", "fluid"])
def test_few_column_acting_kernel_equals_the_generic_launch_sequence(pkg, which):
    """Round 6: agent(env) for the reference's own shape (ONE trajectory: A columns; fp64 fields, Float32 networks,
    src/PDEagent.jl:183-207) is ONE launch -- parameters promoted on the fly, no promoted copy of the actor -- instead of the
    seven of the generic path (parameter copy, pack, a GEMM per layer, randn, noise + clamp).  Same arithmetic: the actions
    equal, BIT FOR BIT, what the generic sequence gives through a promoted clone (pdec_mlp_copy -> pdec_randn -> pdec_policy_act),
    with and without exploration noise, in fp64 and for fp32 states with the actor's own type."""
    import ctypes as C
    setup = {"ks22": lambda: pkg.KSSetup.KS22(), "ks200": lambda: pkg.KSSetup.KS200(), "keller_segel": lambda: pkg.KellerSegelSetup(),
             "ks22_three_layer": lambda: pkg.KSSetup.KS22(drop_middle_layer=False), "fluid": lambda: pkg.FluidSetup(nx=64)}[which]()
    ns, A = setup.state_shape
    agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(3), start_steps=-1)
    pol, L = agent.policy, pkg._lib
    m = pol.behavior_actor.model
    na = m.dims[-1]
    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float64, torch.float32):
        state = torch.randn(A, ns, generator=g, dtype=torch.float64).to(dtype).cuda()
        for learning in (1, 0):
            pol._noise_seed, pol._noise_off, pol.act_noise = 11, 40, 0.7
            out = torch.empty(A, na, dtype=dtype, device="cuda")
            pol.act_into(state, A, dtype, out, bool(learning))
            assert pol._noise_off == 40 + (learning and (A * na + 3) // 4)
            # the generic launch sequence on a clone of the actor in the state's type
            clone = m.clone(dtype=dtype, max_cols=A)
            noise = torch.zeros(A, na, dtype=dtype, device="cuda")
            L.check(pol.lib.pdec_randn(clone.handle, L.ptr(noise), A * na, L.dtype_code(dtype), 11, 40))
            want = torch.empty_like(out)
            L.check(pol.lib.pdec_policy_act(clone.handle, L.ptr(state), L.ptr(noise) if learning else None, A, 0.7, float(pol.act_limit),
                                            L.ptr(want)))
            torch.cuda.synchronize()
            mfma = C.c_int(0)
            L.check(pol.lib.pdec_mlp_acts_on_published_copy(m.handle, C.byref(mfma)))
            if mfma.value and dtype == torch.float32:
                # fp32 states of a 3-layer fp32 actor keep the fused MFMA acting kernel (another summation order: ~1e-7)
                assert float((out - want).abs().max()) <= 1e-6
                continue
            assert torch.equal(out, want), (which, dtype, learning, float((out - want).abs().max()))
            assert bool(torch.isfinite(out).all()) and float(out.abs().max()) <= pol.act_limit


@pytest.mark.parametrize("B", [1, 8])
def test_run_loop_trains_and_logs(pkg, B):
    """run(agent, env, stop, hook) with the reference's stage order; KS22 constants, short run"""
    setup = pkg.KSSetup.KS22(te=1.0, update_loops=2, start_steps=2, update_after=2)
    env = pkg.PDEenv(setup, B=B, dtype=torch.float32)
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), trajectory_length=2000)
    hook = pkg.PDEhook(min_best_episode=1, use_random_init=True)
    before = copy.deepcopy(agent.policy.behavior_actor).params()
    pkg.run(agent, env, pkg.StopAfterEpisodeWithMinSteps(25), hook)
    assert len(hook.rewards) == 3 and all(np.isfinite(hook.rewards))
    # 11 steps per episode: ten additions of dt=0.1 give 0.9999999999999999 < te (same in Julia;
    # the reference's te=5.0 episodes log 51 rows for the same reason)
    assert hook.bestepisode >= 1 and len(hook.bestDF) == 11
    row = hook.bestDF[-1]
    assert row["y"].shape == (192,) and row["p"].shape == (192,) and row["action"].shape == (8,) and row["reward"].shape == (8,)
    after = agent.policy.behavior_actor.params()
    assert any(np.abs(a - b).max() > 0 for a, b in zip(after, before))      # the actor was trained
    assert len(agent.trajectory) == 33 * 8 * B
    al, cl = agent.policy.losses()
    assert np.isfinite(al) and np.isfinite(cl)


@pytest.mark.parametrize("which,B,dtype", [("ks22", 1, "f64"), ("ks22", 4, "f32"), ("kseg", 1, "f64")])
def test_overlapped_run_loop_is_bit_identical_to_the_plain_one(pkg, which, B, dtype):
    """round 4: run(agent, env, stop, hook) with the environment and the networks on two streams runs the update of a control
    step beside its env step (two events per step carry the true dependencies; RL.jl's stage order, src/PDEagent.jl:211-361,
    src/PDEenv.jl:195-241, is untouched).  Same kernels, arguments and per-stream order as the one-stream loop: episode rewards,
    all four networks, the replay traces and the logged best trajectory are bit-identical -- in the overlapped form, in the
    two-stream form with every stage joined, and on the default stream."""
    dt = torch.float64 if dtype == "f64" else torch.float32
    out = []
    for mode in ("default_stream", "two_streams_overlap", "two_streams_joined"):
        setup = (pkg.KSSetup.KS22(te=1.0, update_loops=3, start_steps=2, update_after=2) if which == "ks22"
                 else pkg.KellerSegelSetup(te=0.06, update_loops=3, start_steps=2, update_after=1))
        if mode == "default_stream":
            s_env = s_upd = None
        else:
            s_env, s_upd = torch.cuda.Stream(), torch.cuda.Stream()
        env = pkg.PDEenv(setup, B=B, dtype=dt, stream=s_env)
        agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), trajectory_length=2000, stream=s_upd)
        hook = pkg.PDEhook(min_best_episode=1, use_random_init=True)
        pkg.run(agent, env, pkg.StopAfterEpisodeWithMinSteps(30), hook, overlap=None if mode != "two_streams_joined" else False)
        torch.cuda.synchronize()
        pol, tr = agent.policy, agent.trajectory
        snap = [np.asarray(hook.rewards)]
        for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
            snap += [x.copy() for x in getattr(pol, n).params()]
        snap += [t.cpu().numpy().copy() for t in (tr.state, tr.action, tr.reward, tr.terminal)]
        snap += [np.asarray(r["y"]) for r in hook.bestDF] + [np.asarray(r["reward"]) for r in hook.bestDF]
        snap.append(np.asarray([tr.n_sa, tr.n_rt, len(hook.bestDF), hook.bestepisode]))
        out.append(snap)
    assert len(out[0][0]) >= 2 and np.isfinite(out[0][0]).all()
    for other in out[1:]:
        assert len(other) == len(out[0])
        for x, y in zip(out[0], other):
            assert x.shape == y.shape and np.array_equal(x, y)


def test_rollout_with_an_actor_that_lives_on_another_stream(pkg):
    """env.rollout() enqueues on the environment's stream; an actor created on the agent's own stream (the two-stream run
    loop) is moved over for the call and ordered behind the last writer of its parameters: same rows as the same actor living
    on the environment's stream (README example: run(...) on two streams, then a rollout)"""
    import ctypes as C
    setup = pkg.KSSetup.KS22(te=1.0, update_loops=2, start_steps=2, update_after=2)
    s_env, s_upd = torch.cuda.Stream(), torch.cuda.Stream()
    env = pkg.PDEenv(setup, B=1, dtype=torch.float64, stream=s_env)
    agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(0), stream=s_upd)
    hook = pkg.PDEhook(min_best_episode=1, use_random_init=True)
    pkg.run(agent, env, pkg.StopAfterEpisodeWithMinSteps(25), hook)
    a = agent.policy._actor_for(env.dtype, 8)
    assert a.stream is s_upd
    env.reset()
    o1 = env.rollout(a, 9, log=True)
    a2 = a.clone()
    s_env.wait_stream(s_upd)
    pkg._lib.check(a2.lib.pdec_set_stream(a2.handle, C.c_void_p(s_env.cuda_stream)))
    a2.stream = s_env
    env.reset()
    o2 = env.rollout(a2, 9, log=True)
    torch.cuda.synchronize()
    assert torch.equal(o1["y"], o2["y"]) and torch.equal(o1["action"], o2["action"]) and bool(torch.isfinite(o1["y"]).all())
    assert float(o1["action"].abs().max()) > 0


def test_native_rccl_comm_single_rank(pkg):
    """pdec_comm_* with nranks = 1: all-reduce of the gradient buffer is the identity"""
    import ctypes as C
    lib = pkg._lib.init(0)
    uid = (C.c_char * 128)()
    pkg._lib.check(lib.pdec_comm_unique_id(uid))
    h = pkg._lib.Handle()
    pkg._lib.check(lib.pdec_comm_create(C.byref(h), 1, 0, uid))
    net = pkg.HipMLP([3, 16, 1], ["relu", "tanh"], pkg.glorot_uniform(np.random.default_rng(0), [3, 16, 1]), max_cols=32)
    x = torch.randn(32, 3, device="cuda:0"); dy = torch.randn(32, 1, device="cuda:0")
    g0, _ = net.backward(x, dy)
    pkg._lib.check(lib.pdec_allreduce_grads(h, net.handle))
    torch.cuda.synchronize()
    red = pkg.distributed.GradReducer()._view(net).cpu().numpy()
    flat = np.concatenate([g0[0].ravel(), g0[1], g0[2].ravel(), g0[3]])      # internal layout: W row-major
    assert np.allclose(red, flat, atol=1e-6)
    pkg._lib.check(lib.pdec_destroy(h))


def test_reference_trained_actor_controls_ks22_on_this_path(pkg):
    """End-to-end control quality: the actor the reference trained (hook.bestNNA, fixture
    extracted from scripts/KS/KS22/saves/hook.jld2), run noise-free through PDEenv + the policy on this path from the
    golden initial state, reproduces the oracle's closed-loop rollout step for step and suppresses the KS
    instability (return -0.125 vs -4.35 uncontrolled; the reference's best training episode logged -0.73 with
    exploration noise)."""
    import numpy as np
    import torch
    from oracle import ks, nn
    from util import ks_pair
    setup, cfg, g = ks_pair(pkg, "ks22")
    env = pkg.PDEenv(setup, B=1, dtype=torch.float64, y0=g["y"][0])
    agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(0), dtype=torch.float32)
    best = [g["best_W1"], g["best_b1"], g["best_W2"], g["best_b2"]]
    pkg.checkpoint.load_actor(agent.policy.behavior_actor, best)
    agent.policy.start_steps = -1
    P = [b.astype(np.float64) for b in best]
    y, a_prev, t, ret_o, ret_g = g["y"][0].copy(), np.zeros((1, 8)), 0.0, 0.0, 0.0
    for k in range(50):
        a = np.clip(nn.forward(P, [nn.RELU, nn.TANH], ks.featurize(cfg, y)), -1, 1)
        o = ks.env_step(cfg, y, a_prev, a, t)
        y, a_prev, t = o["y"], a, o["time"]
        ret_o += o["reward"].mean()
        act = agent.policy(env, learning=False)
        env(act)
        ret_g += float(env.reward.mean().item())
        assert np.abs(env.action_julia() - a).max() <= 1e-9
        assert np.abs(env.y_julia() - y).max() <= 1e-8
    assert abs(ret_g - ret_o) <= 1e-8 and -0.3 < ret_g < -0.05
    # the uncontrolled run from the same state is an order of magnitude worse
    env0 = pkg.PDEenv(setup, B=1, dtype=torch.float64, y0=g["y"][0])
    ret0 = 0.0
    for k in range(50):
        env0(torch.zeros(env0._ashape, dtype=torch.float64, device="cuda:0"))
        ret0 += float(env0.reward.mean().item())
    assert ret0 < 10 * ret_g


def test_agent_checkpoint_roundtrip(pkg, tmp_path):
    """save_agent / load_agent (.npz): weights, ADAM moments and beta powers survive; a resumed agent takes the
    same next update step bit for bit"""
    import numpy as np
    import torch
    setup = pkg.KSSetup.bench_C2(256)
    mk = lambda: pkg.create_agent(setup=setup, B=2, rng=np.random.default_rng(3), dtype=torch.float32, max_update_cols=512)
    a1, a2 = mk(), pkg.create_agent(setup=setup, B=2, rng=np.random.default_rng(99), dtype=torch.float32, max_update_cols=512)
    rng = np.random.default_rng(0)
    def batch():
        f = lambda *s: torch.as_tensor(rng.standard_normal(s).astype(np.float32), device="cuda:0")
        return dict(state=f(512, 3), action=f(512, 1).clamp(-1, 1), reward=-f(512).abs(), terminal=torch.zeros(512, device="cuda:0"),
                    next_state=f(512, 3))
    a1.policy.update(batch())
    path = str(tmp_path / "agent.npz")
    pkg.checkpoint.save_agent(path, a1)
    pkg.checkpoint.load_agent(path, a2)
    b = batch()
    a1.policy.update(b)
    a2.policy.update(b)
    for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        for x, y in zip(getattr(a1.policy, n).model.params(), getattr(a2.policy, n).model.params()):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("case", ["ks_c2_f32", "ks_c2_f32_persistent", "kseg_f64", "kseg_f64_persistent", "kseg_f32_persistent", "kseg2d_f32"])
def test_rollout_equals_step_by_step_loop(pkg, case, monkeypatch):
    """pdec_rollout (T control steps in one call, row F2) against the per-step loop policy_act_rng -> env(action), including
    the accumulated reward, the logged rows and the step at which a trajectory blows up.  The host-enqueued form
    (Keller-Segel, 2-D; KS with PDEC_ROLLOUT_PERSISTENT=0) issues the same kernels and is bit-identical; the persistent
    KS kernel (one launch for all T steps, actor evaluated on the vector unit inside it) draws the same noise but sums the
    actor's layers in a different order than the MFMA acting kernel: fp32 actions <= 2e-6, fields <= 2e-5 over 6 steps.
    Round 3: the 1-D Keller-Segel environment has the same one-launch form (kseg_rollout_kernel; KellerSegelSetup.jl:213-332):
    fp64 to 1e-11, fp32 to the KS tolerances."""
    import ctypes as C
    L = pkg._lib
    persistent = case.endswith("_persistent")
    monkeypatch.setenv("PDEC_ROLLOUT_PERSISTENT", "1" if persistent else "0")
    if case.startswith("ks_c2_f32"):
        setup, dt, B = pkg.KSSetup.bench_C2(256), torch.float32, 5
        y0 = setup.generate_random_init(np.random.default_rng(0), B) * 0.15
    elif case.startswith("kseg_f"):
        setup, dt, B = pkg.KellerSegelSetup(), (torch.float32 if "f32" in case else torch.float64), 3
        y0 = np.swapaxes(setup.generate_random_init(np.random.default_rng(0), B), 1, 2)
    else:
        setup, dt, B = pkg.KellerSegel2DSetup(nx=64, ny=32, substeps=4), torch.float32, 2
        y0 = np.moveaxis(setup.generate_random_init(np.random.default_rng(0), B), 1, -1)
    T, noise, lim, seed = 6, 0.3, 1.0, 99
    envs = [pkg.PDEenv(setup, B=B, dtype=dt, y0=np.ascontiguousarray(y0), autoreset=False) for _ in range(2)]
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=dt, start_steps=-1)
    actor = agent.policy.behavior_actor.model
    ns, A = setup.state_shape
    cols = B * A
    # reference loop
    e = envs[0]
    rsum = torch.zeros_like(e.reward)
    rows = []
    off = 0
    for t in range(T):
        a = torch.empty(e._ashape, dtype=dt, device="cuda:0")
        L.check(e.lib.pdec_policy_act_rng(actor.handle, L.ptr(e.state), cols, noise, lim, 1, seed, off, L.ptr(a)))
        off += (cols + 3) // 4
        e(a)
        rsum += e.reward
        rows.append((e.y.clone(), e.p.clone(), e.action.clone(), e.reward.clone()))
    out = envs[1].rollout(actor, T, act_noise=noise, act_limit=lim, learning=True, seed=seed, offset=0, log=True)
    torch.cuda.synchronize()
    if persistent:
        sc = 1e-6 if dt == torch.float64 else 1.0          # fp64: 2e-12 / 2e-11
        close = lambda x, y, tol: float((x - y).abs().max()) <= tol * sc
        assert close(out["action"][0], rows[0][2], 2e-6)          # same state, same noise element for element
        assert close(envs[1].y, e.y, 2e-5) and close(envs[1].state, e.state, 2e-5) and close(envs[1].action, e.action, 2e-5)
        assert close(out["reward_sum"], rsum, 2e-5)
        for t in range(T):
            assert close(out["y"][t], rows[t][0], 2e-5) and close(out["p"][t], rows[t][1], 2e-4)
            assert close(out["action"][t], rows[t][2], 2e-5) and close(out["reward"][t], rows[t][3], 2e-5)
    else:
        assert torch.equal(envs[1].y, e.y) and torch.equal(envs[1].state, e.state) and torch.equal(envs[1].action, e.action)
        assert torch.equal(out["reward_sum"], rsum)
        for t in range(T):
            assert torch.equal(out["y"][t], rows[t][0]) and torch.equal(out["p"][t], rows[t][1])
            assert torch.equal(out["action"][t], rows[t][2]) and torch.equal(out["reward"][t], rows[t][3])
    assert envs[1].steps == T and int(out["done_any"].sum()) == 0 and out["done_step"].tolist() == [-1] * B
    # blow-up bookkeeping: a trajectory started beyond max_value is flagged at step 0
    envs[1].y[B - 1].fill_(1e3)
    out2 = envs[1].rollout(actor, 2, learning=False)
    assert out2["done_step"].tolist()[B - 1] == 0 and bool(envs[1].done[B - 1])


def test_testrun_helper_equals_policy_rollout(pkg):
    """pkg.testrun (evaluation episode as one device-side rollout) == stepping the noise-free policy by hand"""
    setup = pkg.KSSetup.KS22()
    env = pkg.PDEenv(setup, B=2, dtype=torch.float64)
    agent = pkg.create_agent(setup=setup, B=2, rng=np.random.default_rng(4), start_steps=-1)   # no warm-up policy
    out = pkg.testrun(agent, env, steps=10)
    ref = pkg.PDEenv(setup, B=2, dtype=torch.float64)
    ref.reset()
    total = torch.zeros_like(ref.reward)
    for _ in range(10):
        a = agent.policy(ref, learning=False)
        ref(a)
        total += ref.reward
    assert torch.allclose(out["reward_sum"], total, rtol=0, atol=1e-12)
    assert torch.allclose(env.y, ref.y, rtol=0, atol=1e-12)
    assert out["episode_reward"].shape == (2,) and out["y"].shape[0] == 10


# ---------------------------------------------------------------------------------------------------------------------
# round 2: per-trajectory reset, device-side replay (row F1), initialiser kernels (F4), graph replay (F2), closed loops
def test_blown_up_trajectory_does_not_poison_the_learner(pkg):
    """B > 1: one trajectory is pushed past max_value.  The step raises its flag, pushes ONE terminal transition and
    restarts that trajectory from its initial condition in the same step (pdec_env_autoreset); every state that
    reaches the replay stays finite and so do all four networks after the following updates."""
    setup = pkg.KSSetup.KS22()
    B = 4
    env = pkg.PDEenv(setup, B=B, dtype=torch.float64)
    assert env.autoreset
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), trajectory_length=4000, start_steps=-1,
                             update_after=1)
    hook = pkg.PDEhook(collect_bestDF=False)
    stages = pkg.agent
    env.reset()
    agent(stages.PRE_EPISODE_STAGE, env)
    y0, s0 = env.y0.clone(), env.state.clone()
    for k in range(8):
        if k == 2:
            env.y[2] += 40.0                     # beyond max_value = 30 (the mean mode is conserved): this step raises its flag
        a = agent(env)
        agent(stages.PRE_ACT_STAGE, env, a)
        env(a)
        agent(stages.POST_ACT_STAGE, env)
        hook(stages.POST_ACT_STAGE, agent, env)
        flags = env._done_flags.cpu().numpy()
        if k == 2:
            assert flags.tolist() == [0, 0, 1, 0]
            assert torch.equal(env.y[2], y0[2]) and torch.equal(env.state[2], s0[2])       # restarted in the same step
            assert not torch.equal(env.y[1], y0[1])
        else:
            assert flags.sum() == 0
        assert bool(torch.isfinite(env.y).all()) and bool(torch.isfinite(env.state).all()) and bool(torch.isfinite(env.reward).all())
        assert not env.is_terminated()
    tr = agent.trajectory
    n = len(tr)
    assert bool(torch.isfinite(tr.state[:n + tr.stride]).all()) and bool(torch.isfinite(tr.reward[:n]).all())
    term = tr.terminal[:n].view(-1, B, 8)
    assert term[2, 2].eq(1).all() and int(term.sum()) == 8          # exactly one terminal transition per actuator
    for nna in (agent.policy.behavior_actor, agent.policy.behavior_critic, agent.policy.target_actor, agent.policy.target_critic):
        assert all(np.isfinite(p).all() for p in nna.params())
    hook._flush(env)
    assert np.isfinite(hook.reward)


def test_device_replay_rebuilds_the_reference_buffer_and_rewards(pkg):
    """GPU half of tests/test_replay_golden.py (A17 + A22 + F1): the stage kernels (pdec_replay_push_sa / _push_rt) fed with
    the reference's own stream of states / actions / rewards rebuild the head of its saved replay buffer bit for bit,
    pdec_replay_sample fetches s' at +A, and pdec_reward on a field with the stored next-state sensors returns the
    stored reward (Float32 traces: <= 1e-6)."""
    from test_replay_golden import replay_head, ks22_cfg, field_with_sensors, transitions, A, EP
    s, a, r, t = replay_head()
    n_ep = len(r) // EP
    dev = torch.device("cuda:0")
    setup = pkg.KSSetup.KS22()
    agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(0), trajectory_length=150000)
    agent._maybe_update = lambda: None
    tr = agent.trajectory
    assert tr._h is not None and tr.capacity == 150000 and tr.stride == A

    class Env:
        B, te, _ashape = 1, 5.0, (1, A, 1)
    env = Env()
    st = pkg.agent
    for e in range(n_ep):
        agent(st.PRE_EPISODE_STAGE, env)
        for k in range(51):
            i = e * EP + k * A
            env.state = torch.as_tensor(s[i:i + A], device=dev).reshape(1, A, 1).double()      # fp64 env state -> fp32 trace
            agent(st.PRE_ACT_STAGE, env, torch.as_tensor(a[i:i + A], device=dev).reshape(1, A, 1).double())
            env.reward = torch.as_tensor(r[i:i + A], device=dev).reshape(1, A).double()
            env._done_flags = torch.zeros(1, dtype=torch.int32, device=dev)
            env.time = 5.1 if k == 50 else 0.0                                                 # time-out on step 51
            agent(st.POST_ACT_STAGE, env)
        env.state = torch.full((1, A, 1), 123.0, device=dev, dtype=torch.float64)
        agent(st.POST_EPISODE_STAGE, env)
    torch.cuda.synchronize()
    n = n_ep * EP
    assert len(tr) == n and tr.n_sa == n + A
    assert np.array_equal(tr.state[:n, 0].cpu().numpy(), s[:n]) and np.array_equal(tr.action[:n, 0].cpu().numpy(), a[:n])
    assert np.array_equal(tr.reward[:n].cpu().numpy(), r[:n]) and np.array_equal(tr.terminal[:n].cpu().numpy(), t[:n])
    # pde_sample / pde_fetch! on the device against the oracle's restatement of the same counter stream
    from oracle import rng as orng
    b = tr.sample_device(77, 5, 512)
    sl = orng.sample_slots(77, 5, 512, len(tr), tr.n_rt, tr.capacity, tr.stride)
    assert np.array_equal(b["state"][:, 0].cpu().numpy(), s[sl[0]]) and np.array_equal(b["next_state"][:, 0].cpu().numpy(), s[sl[0] + A])
    assert np.array_equal(b["reward"].cpu().numpy(), r[sl[1]]) and np.array_equal(b["terminal"].cpu().numpy(), t[sl[1]])
    assert np.array_equal(b["action"][:, 0].cpu().numpy(), a[sl[0]]) and sl[0].max() < n - A
    # reward_function / featurize kernels against the stored rewards
    cfg = ks22_cfg()
    rows = transitions()
    Bq = len(rows)
    penv = pkg.PDEenv(setup, B=Bq, dtype=torch.float64, autoreset=False)
    y = np.stack([field_with_sensors(cfg, sn) for _, sn, _, _, _ in rows])
    act = np.stack([x[2] for x in rows]).astype(np.float64)
    actp = np.stack([x[3] for x in rows]).astype(np.float64)
    yd = torch.as_tensor(y, device=dev)
    got_s = penv.featurize(yd).cpu().numpy()[:, :, 0]
    assert np.abs(got_s - np.stack([x[1] for x in rows])).max() <= 1e-12
    got_r = penv.reward_function(yd, torch.as_tensor(act, device=dev).reshape(penv._ashape),
                                 torch.as_tensor(actp, device=dev).reshape(penv._ashape)).cpu().numpy()
    assert np.abs(got_r - np.stack([x[4] for x in rows])).max() <= 1e-6


def test_in_kernel_sampling_equals_host_slots_from_the_same_stream(pkg):
    """pdec_ddpg_update_small_rng (pde_sample inside the kernel, the host passes seed + offset) == pdec_ddpg_update_small
    fed with the slots the oracle derives from the same Philox stream: identical parameters after 20 x 3 updates, for
    the register-resident 2-layer kernel (KS22 shapes) and the generic one (3-layer nets), incl. a wrapped buffer"""
    from oracle import rng as orng
    for setup, kw in ((pkg.KSSetup.KS22(), {}), (pkg.KSSetup.bench_C2(256), {})):
        ns, A = setup.state_shape
        agents = [pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(5), trajectory_length=320) for _ in range(2)]
        g = torch.Generator().manual_seed(3)
        for ag in agents:
            tr = ag.trajectory
            g.manual_seed(3)
            for step in range(55):                       # 55 steps into a 320-column buffer (40 / 5 steps): wraps
                tr.push_sa(torch.randn(A, ns, generator=g).cuda(), (torch.rand(A, 1, generator=g) * 2 - 1).cuda())
                tr.push_rt(-torch.rand(A, generator=g).cuda(), (torch.rand(A, generator=g) < 0.1).float().cuda())
            tr.push_sa(torch.randn(A, ns, generator=g).cuda(), None)
        p0, p1 = agents[0].policy, agents[1].policy
        tr = agents[0].trajectory
        assert tr.n_rt > tr.capacity
        p0._sample_seed, p0._sample_off = 4242, 17
        p0.update_small_rng(tr)
        slots = orng.sample_slots(4242, 17, p1.update_loops * p1.batch_size, len(tr), tr.n_rt, tr.capacity, tr.stride)
        p1.update_small(agents[1].trajectory, slots.reshape(3, p1.update_loops, p1.batch_size))
        torch.cuda.synchronize()
        assert p0._sample_off == 17 + (p1.update_loops * p1.batch_size + 3) // 4
        for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
            for x, y in zip(getattr(p0, n).model.params(), getattr(p1, n).model.params()):
                assert np.array_equal(x, y), n
        assert p0.losses() == p1.losses()


@pytest.mark.parametrize("which", ["two_layer_register_kernel", "three_layer_generic_kernel"])
def test_small_update_leaves_frozen_targets_untouched(pkg, which):
    """with the reference's frozen target networks (rho = 1; src/PDEagent.jl:415-417 iterates over an empty list)
    the small-batch update -- the path of the reference-shaped B = 1, 20 x 3 training -- must not TOUCH the targets, like the
    large-batch finish kernels and pdec_polyak: dest = 1 * dest + 0 * src would rewrite them, and turn a non-finite behaviour
    parameter into a NaN target.  An Inf planted in a behaviour bias stays out of the targets (bit-identical to before the
    update); with moving targets (rho = 0.995) the same update does move them."""
    import warnings
    setup = pkg.KSSetup.KS22() if which.startswith("two") else pkg.KSSetup.bench_C2(256)
    ns, A = setup.state_shape
    for frozen in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", pkg.TargetNetworkWarning)
            agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(5), trajectory_length=320, quirk_frozen_targets=frozen)
        pol, tr = agent.policy, agent.trajectory
        g = torch.Generator().manual_seed(3)
        for step in range(30):
            tr.push_sa(torch.randn(A, ns, generator=g).cuda(), (torch.rand(A, 1, generator=g) * 2 - 1).cuda())
            tr.push_rt(-torch.rand(A, generator=g).cuda(), torch.zeros(A).cuda())
        tr.push_sa(torch.randn(A, ns, generator=g).cuda(), None)
        assert pol.small_update_ok()
        if frozen:                                   # an Inf in the behaviour critic's last bias
            prm = [x.copy() for x in pol.behavior_critic.model.params()]
            prm[-1][...] = np.inf
            pol.behavior_critic.model.set_params(prm)
        before = [x.copy() for n in (pol.target_actor, pol.target_critic) for x in n.model.params()]
        pol.update_small_rng(tr)
        torch.cuda.synchronize()
        after = [x for n in (pol.target_actor, pol.target_critic) for x in n.model.params()]
        same = all(np.array_equal(a, b) for a, b in zip(before, after))
        assert same == frozen
        assert all(np.isfinite(x).all() for x in after)


@pytest.mark.parametrize("which", ["KS22", "KS200"])
def test_split_small_update_is_bit_identical(pkg, which, monkeypatch):
    """Frozen targets (the KS experiments' regime): the 20 x 3 update with the critic and the actor updates as two chains side
    by side (ddpg_small2f_kernel: TD targets of all minibatches up front, actor update i - 1 on one more wave beside critic update
    i) == the one-chain kernel (PDEC_SMALL_SPLIT=0), bit for bit: parameters, ADAM moments, beta powers, losses, after three
    launches on a wrapped buffer."""
    setup, kw = (pkg.KSSetup.KS22() if which == "KS22" else pkg.KSSetup.KS200()), {}
    ns, A = setup.state_shape
    agents = [pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(5), trajectory_length=320, **kw) for _ in range(2)]
    g = torch.Generator()
    for ag in agents:
        tr = ag.trajectory
        g.manual_seed(3)
        for step in range(55):
            tr.push_sa(torch.randn(A, ns, generator=g).cuda(), (torch.rand(A, 1, generator=g) * 2 - 1).cuda())
            tr.push_rt(-torch.rand(A, generator=g).cuda(), (torch.rand(A, generator=g) < 0.1).float().cuda())
        tr.push_sa(torch.randn(A, ns, generator=g).cuda(), None)
    for ag, split in zip(agents, ("1", "0")):
        monkeypatch.setenv("PDEC_SMALL_SPLIT", split)
        pol = ag.policy
        assert pol.small_update_ok() and pol.rho_effective == 1.0
        pol._sample_seed, pol._sample_off = 4242, 17
        for _ in range(3):
            pol.update_small_rng(ag.trajectory)
        torch.cuda.synchronize()
    p0, p1 = agents[0].policy, agents[1].policy
    for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        m0, m1 = getattr(p0, n).model, getattr(p1, n).model
        for x, y in zip(m0.params(), m1.params()):
            assert np.array_equal(x, y), n
        if n.startswith("behavior"):
            from importlib import import_module
            ck = import_module(pkg.__name__ + ".checkpoint")
            for x, y in zip(ck._adam_state(m0), ck._adam_state(m1)):          # m, v, beta powers
                assert np.array_equal(x, y), n
    assert p0.losses() == p1.losses()
    # and the update did something
    fresh = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(5), trajectory_length=320, **kw).policy
    assert not np.array_equal(fresh.behavior_critic.model.params()[0], p0.behavior_critic.model.params()[0])


@pytest.mark.parametrize("which", ["keller_segel", "fluid"])
def test_owner_lane_adam_is_bit_identical(pkg, which, monkeypatch):
    """Moving targets (the Keller-Segel / fluid experiments' regime), the 20 x 3 update in one launch: the actor's ADAM state one
    parameter per lane of wave 0 (gradients and new weights / targets through LDS; 5 ADAM chains per update for the Keller-Segel
    actor's 280 per-unit parameters instead of 14 on 20 unit threads) == the per-unit form (PDEC_SMALL_OWN=0), bit for bit:
    parameters, targets, ADAM moments, beta powers, losses after three launches on a wrapped buffer."""
    from importlib import import_module
    ck = import_module(pkg.__name__ + ".checkpoint")
    setup = pkg.KellerSegelSetup() if which == "keller_segel" else pkg.FluidSetup(nx=128)
    ns, A = setup.state_shape
    agents = [pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(5), trajectory_length=40 * A) for _ in range(2)]
    g = torch.Generator()
    for ag in agents:
        tr = ag.trajectory
        g.manual_seed(3)
        for step in range(55):
            tr.push_sa(torch.randn(A, ns, generator=g).cuda(), (torch.rand(A, 1, generator=g) * 2 - 1).cuda())
            tr.push_rt(-torch.rand(A, generator=g).cuda(), (torch.rand(A, generator=g) < 0.1).float().cuda())
        tr.push_sa(torch.randn(A, ns, generator=g).cuda(), None)
    for ag, own in zip(agents, ("1", "0")):
        monkeypatch.setenv("PDEC_SMALL_OWN", own)
        pol = ag.policy
        assert pol.small_update_ok() and pol.rho_effective < 1.0
        pol._sample_seed, pol._sample_off = 4242, 17
        for _ in range(3):
            pol.update_small_rng(ag.trajectory)
        torch.cuda.synchronize()
    p0, p1 = agents[0].policy, agents[1].policy
    for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        m0, m1 = getattr(p0, n).model, getattr(p1, n).model
        for x, y in zip(m0.params(), m1.params()):
            assert np.array_equal(x, y), n
        if n.startswith("behavior"):
            for x, y in zip(ck._adam_state(m0), ck._adam_state(m1)):
                assert np.array_equal(x, y), n
    assert p0.losses() == p1.losses()
    fresh = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(5), trajectory_length=40 * A).policy
    assert not np.array_equal(fresh.target_actor.model.params()[0], p0.target_actor.model.params()[0])


@pytest.mark.parametrize("case", ["mid_episode", "zero_policy", "episode_ends_here", "episode_ended_before"])
def test_step_glue_equals_the_three_launches(pkg, case):
    """pdec_step_glue (POST_ACT push of step t - 1 + agent(env) + PRE_ACT push of step t in ONE launch) == the three calls the
    stage loop makes (replay_push_rt, the acting kernel, replay_push_sa): same traces, same action, same halt flag -- in the
    middle of an episode, under the start policy's zero action, when step t - 1 ended the episode (its terminal transition is
    pushed, the flag raised, the PRE_ACT push skipped) and when the episode had ended before (nothing pushed)."""
    import ctypes as C
    from importlib import import_module
    from types import SimpleNamespace
    runmod, L = import_module(pkg.__name__ + ".run"), pkg._lib
    setup = pkg.KSSetup.KS22()
    ns, A = setup.state_shape
    out = []
    for glue in (True, False):
        agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(5), trajectory_length=320)
        pol, tr = agent.policy, agent.trajectory
        g = torch.Generator().manual_seed(11)
        for _ in range(7):
            tr.push_sa(torch.randn(A, ns, generator=g).cuda(), (torch.rand(A, 1, generator=g) * 2 - 1).cuda())
            tr.push_rt(-torch.rand(A, generator=g).cuda(), torch.zeros(A).cuda())
        tr.push_sa(torch.randn(A, ns, generator=g).cuda(), (torch.rand(A, 1, generator=g) * 2 - 1).cuda())     # step t - 1's (s, a)
        dt = torch.float64
        logs = SimpleNamespace(state=torch.randn(3, A, ns, generator=g).to(dt).cuda(), action=torch.zeros(3, A, 1, dtype=dt).cuda(),
                               reward=-torch.rand(3, 1, A, generator=g).to(dt).cuda(), done=torch.zeros(3, dtype=torch.int32).cuda())
        halt = torch.zeros(1, dtype=torch.int32).cuda()
        if case == "episode_ends_here":
            logs.done[0] = 1
        if case == "episode_ended_before":
            halt[0] = 1
        pol._noise_seed, pol._noise_off = 99, 5
        env = SimpleNamespace(dtype=dt)
        acting = case != "zero_policy"
        t = 1
        lib = pol.behavior_actor.model.lib
        L.check(lib.pdec_set_episode_halt(tr._h, L.ptr(halt)))
        try:
            if glue:
                pol._glue_off = False
                assert runmod._step_glue(lib, pol, tr, env, logs, t, A, A, acting, True)
            else:
                tr.push_rt_flags(logs.reward[t - 1].view(-1), logs.done[t - 1:t], A, False)
                a_t = logs.action[t + 1]
                if acting:
                    pol.act_into(logs.state[t], A, dt, a_t.view(A, 1))
                else:
                    a_t.zero_()
                tr.push_sa(logs.state[t].view(A, ns), a_t.view(A, 1))
        finally:
            torch.cuda.synchronize()
            L.check(lib.pdec_set_episode_halt(tr._h, None))
        out.append(dict(state=tr.state.cpu().numpy().copy(), action=tr.action.cpu().numpy().copy(), reward=tr.reward.cpu().numpy().copy(),
                        terminal=tr.terminal.cpu().numpy().copy(), a=logs.action.cpu().numpy().copy(), halt=int(halt.item()),
                        noise=pol._noise_off, n_sa=tr.n_sa, n_rt=tr.n_rt))
    a, b = out
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert a["halt"] == (1 if case.startswith("episode_end") else 0)
    if acting:
        assert np.abs(a["a"][2]).max() > 0 and a["noise"] == 5 + (A + 3) // 4


def test_launch_sync_hand_over_refusal_and_timeout(pkg):
    """pdec_set_launch_sync (device-side hand-over between the glue launch and the fused fp64 KS step of one trajectory):
    (a) producer and consumer on two streams: the env step waits INSIDE the kernel for the flag the glue launch raises and
    raises its own; (b) a launch that cannot honour a pending sync refuses (PDEC_E_INVALID) instead of ignoring it; (c) a wait
    whose hand-over never comes gives up after 0.3 s, counts a timeout, and the launch still completes.  First of all the two
    streams have to run side by side (pdec_streams_run_side_by_side, what run() asks before it uses the sync)."""
    import ctypes as C
    import time
    L = pkg._lib
    lib = L.init(0)
    # HIP maps streams onto a few hardware queues: take streams from torch's pool until the pair sits on two of them
    s_env, s_upd, side, same = torch.cuda.Stream(), torch.cuda.Stream(), C.c_int(0), C.c_int(1)
    for _ in range(8):
        L.check(lib.pdec_streams_run_side_by_side(C.c_void_p(s_env.cuda_stream), C.c_void_p(s_upd.cuda_stream), C.byref(side)))
        if side.value:
            break
        s_upd = torch.cuda.Stream()
    if not side.value:        # (an environment condition, not a defect: run() keeps its stream-level events in that case)
        pytest.skip("no pair of streams on different hardware queues among nine picks")
    L.check(lib.pdec_streams_run_side_by_side(C.c_void_p(s_env.cuda_stream), C.c_void_p(s_env.cuda_stream), C.byref(same)))
    assert same.value == 0
    setup = pkg.KSSetup.KS22()
    ns, A = setup.state_shape
    env = pkg.PDEenv(setup, B=1, dtype=torch.float64, stream=s_env)
    agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(5), trajectory_length=320, stream=s_upd)
    pol, tr = agent.policy, agent.trajectory
    m = pol.behavior_actor.model
    flags = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    n0 = C.c_int(0)
    L.check(lib.pdec_launch_sync_timeouts(C.byref(n0)))
    # (a) env step first in issue order, on its own stream: it can only proceed once the glue launch (issued later) has signalled
    y1, st1 = torch.empty_like(env.y), torch.empty_like(env.state)
    p1, r1 = torch.empty_like(env.p), torch.empty_like(env.reward)
    done = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    a_t = torch.zeros(1, A, 1, dtype=torch.float64, device="cuda:0")
    a_prev = torch.zeros_like(a_t)
    P = L.ptr
    with torch.cuda.stream(s_env):
        L.check(lib.pdec_set_launch_sync(env.handle, P(flags[0:1]), 1, P(flags[1:2]), 1))
        L.check(lib.pdec_env_step(env.handle, P(env.y), P(a_t), P(a_prev), P(env.state), P(y1), P(p1), P(st1), P(r1), P(done)))
    time.sleep(0.02)
    with torch.cuda.stream(s_upd):            # (read through the OTHER stream: the null stream may share s_env's hardware queue)
        seen = flags.to("cpu", non_blocking=True)
    s_upd.synchronize()
    assert seen.tolist() == [0, 0]            # the env step is still waiting
    served = C.c_int(0)
    pol._noise_seed, pol._noise_off = 3, 0
    with torch.cuda.stream(s_upd):
        L.check(lib.pdec_set_launch_sync(m.handle, None, 0, P(flags[0:1]), 1))
        cap1 = tr.capacity + tr.stride
        L.check(lib.pdec_step_glue(m.handle, tr._h, L.dtype_code(torch.float64), None, None, A, 0, P(tr.reward), P(tr.terminal),
                                   tr.capacity, 0, 0, 1, P(env.state), A, float(pol.act_noise), float(pol.act_limit), 3, 0, P(a_t),
                                   P(tr.state), P(tr.action), cap1, 0, A, L.Handle(0), C.byref(served)))
    assert served.value == 1
    torch.cuda.synchronize()
    assert flags.tolist() == [1, 1]
    # the step saw the action the glue launch wrote: same result as an ordinary step on that action
    y2, st2, p2, r2 = torch.empty_like(env.y), torch.empty_like(env.state), torch.empty_like(env.p), torch.empty_like(env.reward)
    with torch.cuda.stream(s_env):
        L.check(lib.pdec_env_step(env.handle, P(env.y), P(a_t), P(a_prev), P(env.state), P(y2), P(p2), P(st2), P(r2), P(done)))
    torch.cuda.synchronize()
    assert float(a_t.abs().max()) > 0 and torch.equal(y1, y2) and torch.equal(st1, st2) and torch.equal(r1, r2)
    # (b) a batched fp32 step is not the launch that honours it
    env32 = pkg.PDEenv(pkg.KSSetup.bench_C2(256), B=4, dtype=torch.float32)
    L.check(lib.pdec_set_launch_sync(env32.handle, P(flags[0:1]), 1, None, 0))
    with pytest.raises(pkg.PdecError, match="launch sync"):
        env32(torch.zeros(env32._ashape, dtype=torch.float32, device="cuda:0"))
    env32(torch.zeros(env32._ashape, dtype=torch.float32, device="cuda:0"))          # (cleared: the next step is ordinary)
    # (c) nobody raises the flag to 7: the wait gives up, the launch completes
    t0 = time.perf_counter()
    with torch.cuda.stream(s_upd):
        L.check(lib.pdec_set_launch_sync(m.handle, P(flags[0:1]), 7, None, 0))
        L.check(lib.pdec_step_glue(m.handle, tr._h, L.dtype_code(torch.float64), None, None, A, 0, P(tr.reward), P(tr.terminal),
                                   tr.capacity, 0, 0, 2, P(env.state), A, 0.0, 1.0, 3, 0, P(a_t), P(tr.state), P(tr.action), cap1, 0, A,
                                   L.Handle(0), C.byref(served)))
    torch.cuda.synchronize()
    assert 0.25 < time.perf_counter() - t0 < 2.0
    n1 = C.c_int(0)
    L.check(lib.pdec_launch_sync_timeouts(C.byref(n1)))
    assert n1.value == n0.value + 1 and float(a_t.abs().max()) == 0.0


def test_random_init_kernels_match_the_oracle_stream(pkg):
    """pdec_env_random_init (generate_random_init of KSSetup.jl:288-298 / KellerSegelSetup.jl:373-384 as a kernel) against
    the oracle's formulas evaluated with the coefficients of the same Philox stream (oracle/rng.py)"""
    from oracle import rng as orng
    B = 5
    setup = pkg.KSSetup.KS22()
    env = pkg.PDEenv(setup, B=B, dtype=torch.float64)
    y = torch.empty_like(env.y)
    used = env.random_init(11, 3, out=y)
    assert used == B * 2
    a = orng.random_init_coefficients(11, 3, B, 8)
    xx = setup.dx * np.arange(1, setup.nx + 1)
    ref = sum(a[:, i - 1:i] * np.sin(i * xx / (2 * np.pi))[None] for i in range(1, 9))
    ref = ref * 30 / np.linalg.norm(ref, axis=1, keepdims=True)
    assert np.abs(y.cpu().numpy() - ref).max() <= 1e-12 and abs(np.linalg.norm(ref[0]) - 30) < 1e-12
    ks = pkg.KellerSegelSetup()
    env2 = pkg.PDEenv(ks, B=B, dtype=torch.float32)
    y2 = torch.empty_like(env2.y)
    nsin = int(np.ceil(ks.Lx / 3))
    assert env2.random_init(9, 0, out=y2) == B * ((2 * nsin + 3) // 4)
    a = orng.random_init_coefficients(9, 0, B, 2 * nsin)
    xx = ks.dx * np.arange(1, ks.nx + 1)
    ref = np.ones((B, ks.nx, 2))
    for i in range(1, nsin + 1):
        sn = np.sin(i * xx / (2 * np.pi * (ks.Lx / 22)))
        ref[:, :, 0] += a[:, i - 1:i] * sn[None]
        ref[:, :, 1] += a[:, nsin + i - 1:nsin + i] * sn[None]
    assert np.abs(y2.cpu().numpy() - ref).max() <= 3e-7
    # PDEhook's PRE_EPISODE re-initialisation uses the kernel and keeps env.y0 / y / state consistent
    hook = pkg.PDEhook(use_random_init=True, init_seed=5)
    hook(pkg.agent.PRE_EPISODE_STAGE, None, env)
    from oracle import ks as oks
    cfg = oks.KSConfig(192, 22.0, np.arange(1, 193, 24), sigma_sensors=0.7, sigma_actuators=0.7)
    assert torch.equal(env.y, env.y0) and abs(float(env.y[0].norm()) - 30) < 1e-9
    assert np.abs(env.state[1].cpu().numpy().T - oks.featurize(cfg, env.y[1].cpu().numpy())).max() <= 1e-12


def _make_pipeline(pkg, use_graphs, B=64, E=17, lag=2, two_layer=False):
    setup = pkg.KSSetup.bench_C2(256, drop_middle_layer=True) if two_layer else pkg.KSSetup.bench_C2(256)
    s_env, s_upd = torch.cuda.Stream(), torch.cuda.Stream()
    y0 = setup.generate_random_init(np.random.default_rng(0), B) * 0.15
    env = pkg.PDEenv(setup, B=B, dtype=torch.float32, y0=y0, stream=s_env, autoreset=False)
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, stream=s_upd, start_steps=-1,
                             noise_seed=7, trajectory_length=1)
    agent.policy.act_noise = 0.3
    torch.cuda.synchronize()
    return pkg.TrainPipeline(env, agent, lag=lag, episode_steps=E, stream_env=s_env, stream_upd=s_upd, use_graphs=use_graphs,
                             chunks=(6, 1), noise_seed=99)


def test_pipeline_with_the_simd_sharing_step_tracks_the_register_form(pkg, monkeypatch):
    """The training pipeline asks for the 64-VGPR form of the fused KS step (pdec_env_set_simd_sharing).  The same 30
    control steps (acting, PDE step, DDPG update each step, one episode boundary) with the register form instead
    (PDEC_SHARE=0): same arithmetic in the same order, so fields, actions and the trained networks agree to fp32 round-off
    amplified over the steps -- a wrong constant slot or sub-step would show up at once at this size."""
    monkeypatch.setenv("PDEC_SHARE", "1")
    pa = _make_pipeline(pkg, False)
    monkeypatch.setenv("PDEC_SHARE", "0")
    pb = _make_pipeline(pkg, False)
    assert pa.simd_sharing and not pb.simd_sharing
    pa.run(30); pb.run(30)
    pa.sync(); pb.sync()
    assert bool(torch.isfinite(pa.y).all())
    assert float((pa.y - pb.y).abs().max()) <= 2e-3 * float(pb.y.abs().max())
    assert float((pa.aring[(pa.tick - 1) % 3] - pb.aring[(pb.tick - 1) % 3]).abs().max()) <= 2e-3
    for n in ("behavior_actor", "behavior_critic"):
        for x, y in zip(getattr(pa.policy, n).model.params(), getattr(pb.policy, n).model.params()):
            assert np.abs(x - y).max() <= 2e-4 * max(1.0, np.abs(y).max()), n
    pa.close()
    assert not pa.simd_sharing and not pa.env.set_simd_sharing(False)       # close() hands the env back in its register form


def test_events_riding_on_the_reduction_launches_change_nothing(pkg, monkeypatch):
    """pdec_mlp_set_stop_event: the two events the env stream waits for are attached to the reduction launches' own dispatch
    packets instead of being recorded behind them (PDEC_STOP_EVENTS=0).  Pure synchronisation: 40 control steps, across an
    episode boundary, give bit-identical fields, actions, rewards and networks either way; a stop event nobody consumes is
    recorded by pdec_mlp_flush_stop_event."""
    monkeypatch.setenv("PDEC_STOP_EVENTS", "1")
    pa = _make_pipeline(pkg, False)
    monkeypatch.setenv("PDEC_STOP_EVENTS", "0")
    pb = _make_pipeline(pkg, False)
    assert pa.stop_events and not pb.stop_events
    pa.run(40); pb.run(40)
    pa.sync(); pb.sync()
    assert torch.equal(pa.y, pb.y) and torch.equal(pa.state, pb.state) and bool(torch.isfinite(pa.y).all())
    for k in range(3):
        assert torch.equal(pa.aring[k], pb.aring[k]) and torch.equal(pa.rring[k], pb.rring[k])
    for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        for x, y in zip(getattr(pa.policy, n).model.params(), getattr(pb.policy, n).model.params()):
            assert np.array_equal(x, y), n
    # an armed event that no launch consumes: flushed as a plain record, after which a waiter is released
    import ctypes as C
    L = pkg._lib
    ev = L.Handle()
    L.check(pa.lib.pdec_event_create(C.byref(ev)))
    h = pa.policy.behavior_actor.model.handle
    L.check(pa.lib.pdec_mlp_set_stop_event(h, ev))
    L.check(pa.lib.pdec_mlp_flush_stop_event(h))
    L.check(pa.lib.pdec_mlp_flush_stop_event(h))           # second call: nothing pending, no-op
    L.check(pa.lib.pdec_stream_wait_event(C.c_void_p(pa.s_env.cuda_stream), ev))
    pa.sync()
    pa.lib.pdec_destroy(ev)


@pytest.mark.parametrize("lag,two_layer", [(2, False), (1, False), (2, True)])
def test_graph_replay_is_bit_identical_to_the_eager_pipeline(pkg, lag, two_layer):
    """row F2: the two-stream control step replayed from captured HIP graphs (chunks of 6 and 1 steps, first / last step
    of each 17-step episode eager) against the same pipeline issued call by call: identical PDE states, actions and
    all four networks after 60 steps, i.e. across episode boundaries, partial chunks and both slot parities"""
    pe = _make_pipeline(pkg, False, lag=lag, two_layer=two_layer)
    pg = _make_pipeline(pkg, True, lag=lag, two_layer=two_layer)
    pg.run(5)
    pg.capture()
    n0 = pg.tick
    pe.run(n0)
    for n in (1, 7, 20, 32):
        pe.run(n)
        pg.run(n)
    pe.sync(); pg.sync()
    assert pe.tick == pg.tick and pg.n_graph_launches > 0 and len(pg.graphs) == 12
    assert torch.equal(pe.y, pg.y) and torch.equal(pe.state, pg.state)
    for k in range(3):
        assert torch.equal(pe.aring[k], pg.aring[k]) and torch.equal(pe.rring[k], pg.rring[k]) and torch.equal(pe.tring[k], pg.tring[k])
    for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        for x, y in zip(getattr(pe.policy, n).model.params(), getattr(pg.policy, n).model.params()):
            assert np.array_equal(x, y), n
    assert bool(torch.isfinite(pg.y).all()) and pe.policy.losses() == pg.policy.losses()
    # the eager pipeline replays recorded library calls for interior steps (pipeline._eager); issued through the Python
    # layers every step it must give the same numbers
    ps = _make_pipeline(pkg, False, lag=lag, two_layer=two_layer)
    ps.fast_eager = False
    ps.run(pe.tick)
    ps.sync()
    assert len(pe._progs) == 6 and not ps._progs
    assert torch.equal(ps.y, pe.y) and torch.equal(ps.state, pe.state)
    for n in ("behavior_actor", "behavior_critic"):
        for x, y in zip(getattr(ps.policy, n).model.params(), getattr(pe.policy, n).model.params()):
            assert np.array_equal(x, y), n
    import ctypes as C
    nn_ = C.c_int()
    pkg._lib.check(pg.lib.pdec_graph_num_nodes(pg.graphs[(6, 0)], C.byref(nn_)))
    assert nn_.value >= 6 * 6            # act, env step and the four update launches of each of the six steps
    pg.close()


def test_pipeline_step_matches_the_oracle_rollout(pkg):
    """the pipelined step against the ORACLE (src/PDEagent.jl:175-209 + src/PDEenv.jl:195-241): with exploration noise
    and learning rates at zero the networks stay fixed, and 12 replayed control steps from the oracle's initial states
    must follow the oracle's closed loop (actor forward -> clamp -> prepare_action -> CNAB2 -> reward -> featurize),
    fp32 <= 2e-5 per step on y (teacher-forced comparison: the oracle restarts every step from the device state)"""
    from oracle import ks, nn
    pg = _make_pipeline(pkg, True, B=6, E=0)
    pol = pg.policy
    pol.act_noise = 0.0
    for nna in (pol.behavior_actor, pol.behavior_critic):
        nna.optimizer.eta = 0.0
    setup = pg.env.setup
    cfg = ks.KSConfig(256, setup.Lx, setup.sensor_positions, sigma_sensors=1.0, sigma_actuators=1.0, window_size=3)
    P = [p.astype(np.float64) for p in pol.behavior_actor.params()]
    acts = [nn.RELU, nn.RELU, nn.TANH]
    pg.run(4)
    pg.capture()
    for _ in range(12):
        pg.sync()
        k = pg.tick
        y = pg.y.cpu().numpy().astype(np.float64)
        a_prev = pg.aring[(k - 1) % 3].cpu().numpy().astype(np.float64)[:, :, 0]
        pg.run(1)
        pg.sync()
        y1 = pg.y.cpu().numpy()
        a1 = pg.aring[k % 3].cpu().numpy()[:, :, 0]
        for b in range(6):
            a = np.clip(nn.forward(P, acts, ks.featurize(cfg, y[b])), -1, 1)
            assert np.abs(a1[b] - a[0]).max() <= 2e-5
            o = ks.env_step(cfg, y[b], a_prev[b][None], a1[b][None].astype(np.float64), 0.0)
            assert np.abs(y1[b] - o["y"]).max() <= 2e-5
            assert np.abs(pg.rring[k % 3][b].cpu().numpy() - o["reward"]).max() <= 1e-5
            assert np.abs(pg.state[b].cpu().numpy().T - o["state"]).max() <= 1e-5
    assert pg.n_graph_launches >= 12
    pg.close()


def test_rollout_follows_the_oracle_closed_loop(pkg):
    """pdec_rollout (T steps in one call) against an ORACLE rollout, not against the per-step HIP loop: the reference-trained
    KS22 actor (hook.bestNNA), noise-free, 50 control steps from the golden initial state in fp64 -- every logged row
    (y, p, action, reward) equals the oracle's closed loop to 1e-8, and so does the return"""
    from oracle import ks, nn
    from util import ks_pair
    setup, cfg, g = ks_pair(pkg, "ks22")
    env = pkg.PDEenv(setup, B=2, dtype=torch.float64, y0=np.stack([g["y"][0], g["y"][7]]), autoreset=False)
    best = [g["best_W1"], g["best_b1"], g["best_W2"], g["best_b2"]]
    actor = pkg.HipMLP([1, 6, 1], ["relu", "tanh"], best, dtype=torch.float64, max_cols=16)
    out = env.rollout(actor, 50, learning=False, log=True)
    torch.cuda.synchronize()
    P = [b.astype(np.float64) for b in best]
    for b, row in enumerate((0, 7)):
        y, a_prev, ret = g["y"][row].copy(), np.zeros((1, 8)), 0.0
        for k in range(50):
            a = np.clip(nn.forward(P, [nn.RELU, nn.TANH], ks.featurize(cfg, y)), -1, 1)
            o = ks.env_step(cfg, y, a_prev, a, 0.0)
            y, a_prev = o["y"], a
            ret += o["reward"].mean()
            assert np.abs(out["action"][k, b, :, 0].cpu().numpy() - a[0]).max() <= 1e-9
            assert np.abs(out["y"][k, b].cpu().numpy() - y).max() <= 1e-8
            assert np.abs(out["p"][k, b].cpu().numpy() - o["p"]).max() <= 1e-9
            assert np.abs(out["reward"][k, b].cpu().numpy() - o["reward"]).max() <= 1e-9
        assert abs(float(out["reward_sum"][b].mean()) - ret) <= 1e-8


def test_reference_trained_keller_segel_actor_closed_loop(pkg):
    """row F3: the actor the reference trained for Keller-Segel10_16 (hook.bestNNA 12 -> 20 -> 1, fixture from
    scripts/Keller-Segel/Keller-Segel10_16/saves/hook.jld2) driven closed-loop on this path from a golden state: 30
    control steps of policy -> (env)(action) with the temporal state stack (KellerSegelSetup.jl:265-316) follow the
    oracle's loop (fixed 32 RK4 sub-steps on both sides) to 1e-9, per-step and as a device-side rollout"""
    from oracle import keller_segel as kg, nn
    from util import load_golden
    g = load_golden("kseg_hook.npz")
    setup, cfg = pkg.KellerSegelSetup(), kg.KSegConfig()
    best = [g["best_W1"], g["best_b1"], g["best_W2"], g["best_b2"]]
    y0 = g["y_t"][3]
    mem = np.ascontiguousarray(np.swapaxes(y0, 0, 1))[None]
    env = pkg.PDEenv(setup, B=1, dtype=torch.float64, y0=mem)
    env2 = pkg.PDEenv(setup, B=1, dtype=torch.float64, y0=mem)
    agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(0), dtype=torch.float32)
    pkg.checkpoint.load_actor(agent.policy.behavior_actor, best)
    agent.policy.start_steps = -1
    P = [b.astype(np.float64) for b in best]
    y, state, a_prev, rets = y0.copy(), kg.featurize(cfg, y0, None), np.zeros((1, 16)), []
    assert np.abs(env.state[0].cpu().numpy().T - state).max() <= 1e-13
    T = 30
    for k in range(T):
        a = np.clip(nn.forward(P, [nn.RELU, nn.TANH], state), -1, 1)
        p = kg.prepare_action(cfg, a)
        y = kg.do_step(cfg, y, p, 32)
        r = kg.reward_function(cfg, y, a, a - a_prev)
        state, a_prev = kg.featurize(cfg, y, state), a
        rets.append(r)
        act = agent.policy(env, learning=False)
        env(act)
        assert np.abs(env.action_julia() - a).max() <= 1e-9
        assert np.abs(env.y_julia() - y).max() <= 1e-9
        assert np.abs(env.reward[0].cpu().numpy() - r).max() <= 1e-9
        assert np.abs(env.state[0].cpu().numpy().T - state).max() <= 1e-9
    actor64 = agent.policy._actor_for(torch.float64, 16)
    out = env2.rollout(actor64, T, learning=False, log=True)
    torch.cuda.synchronize()
    assert np.abs(out["reward"][:, 0].cpu().numpy() - np.stack(rets)).max() <= 1e-9
    assert np.abs(np.swapaxes(env2.y[0].cpu().numpy(), 0, 1) - y).max() <= 1e-9


def test_agent_written_as_jld2_arrays_reads_back(pkg, tmp_path):
    """row F3 write side on a live agent: save_agent_jld2 / save_actor_jld2 -> the library's JLD2 reader -> the same
    parameters, ADAM moments and beta powers; load_actor puts them onto a fresh network which then acts identically"""
    import importlib
    jl = importlib.import_module("distributedconvrl-pde-control_amd.jld2")
    setup = pkg.KSSetup.KS22()
    agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(2), max_update_cols=64)
    g = torch.Generator().manual_seed(0)
    batch = dict(state=torch.randn(64, 1, generator=g).cuda(), action=torch.rand(64, 1, generator=g).cuda(),
                 reward=-torch.rand(64, generator=g).cuda(), terminal=torch.zeros(64).cuda(), next_state=torch.randn(64, 1, generator=g).cuda())
    agent.policy.update(batch)
    path = str(tmp_path / "agent.jld2")
    pkg.checkpoint.save_agent_jld2(path, agent)
    back = jl.read_arrays(path)
    for name in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        model = getattr(agent.policy, name).model
        params = model.params()
        for li in range(len(params) // 2):
            assert np.array_equal(back[f"{name}_W{li + 1}"], params[2 * li]) and np.array_equal(back[f"{name}_b{li + 1}"], params[2 * li + 1])
        m, v, bp = pkg.checkpoint._adam_state(model)
        assert np.array_equal(back[f"{name}_adam_m"], m) and np.array_equal(back[f"{name}_adam_beta_pow"], bp)
        assert back[f"{name}_dims"].tolist() == list(model.dims)
    assert back["behavior_critic_adam_beta_pow"].tolist() == [0.9 * 0.9, 0.999 * 0.999]       # one ADAM step taken
    apath = str(tmp_path / "best.jld2")
    pkg.checkpoint.save_actor_jld2(apath, agent.policy.behavior_actor)
    b = jl.read_arrays(apath)
    fresh = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(77)).policy.behavior_actor
    pkg.checkpoint.load_actor(fresh, [b["bestNNA_W1"], b["bestNNA_b1"], b["bestNNA_W2"], b["bestNNA_b2"]])
    x = torch.randn(8, 1, generator=g).cuda()
    assert torch.equal(fresh(x), agent.policy.behavior_actor(x))
