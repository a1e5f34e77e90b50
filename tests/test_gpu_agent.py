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
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(3))
    pol = agent.policy
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


@pytest.mark.parametrize("case", ["ks_c2_f32", "kseg_f64", "kseg2d_f32"])
def test_rollout_equals_step_by_step_loop(pkg, case):
    """pdec_rollout (T control steps enqueued in one call, row F2) == the per-step loop policy_act_rng -> env(action),
    bit for bit, including the accumulated reward, the logged rows and the step at which a trajectory blows up"""
    import ctypes as C
    L = pkg._lib
    if case == "ks_c2_f32":
        setup, dt, B = pkg.KSSetup.bench_C2(256), torch.float32, 5
        y0 = setup.generate_random_init(np.random.default_rng(0), B) * 0.15
    elif case == "kseg_f64":
        setup, dt, B = pkg.KellerSegelSetup(), torch.float64, 3
        y0 = np.swapaxes(setup.generate_random_init(np.random.default_rng(0), B), 1, 2)
    else:
        setup, dt, B = pkg.KellerSegel2DSetup(nx=64, ny=32, substeps=4), torch.float32, 2
        y0 = np.moveaxis(setup.generate_random_init(np.random.default_rng(0), B), 1, -1)
    T, noise, lim, seed = 6, 0.3, 1.0, 99
    envs = [pkg.PDEenv(setup, B=B, dtype=dt, y0=np.ascontiguousarray(y0)) for _ in range(2)]
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
