"""GPU tests of pipeline.TrainPipeline beyond the eager / graph identity of test_gpu_agent.py: the device-replay route
(ordering of the stage pushes against the env step, src/PDEagent.jl:237-340), graph capture with episode lengths whose
ring phases are not all reachable, frozen per-step scalars, mid-episode restarts, the full-size C2 instance of the
headline."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(pkg, use_graphs=False, B=64, E=17, lag=2, chunks=(6, 1), use_replay=False, replay_steps=8, nx=256, **setup_kw):
    setup = pkg.KSSetup.bench_C2(nx, **setup_kw)
    s_env, s_upd = torch.cuda.Stream(), torch.cuda.Stream()
    y0 = setup.generate_random_init(np.random.default_rng(0), B) * 0.15
    env = pkg.PDEenv(setup, B=B, dtype=torch.float32, y0=y0, stream=s_env, autoreset=False)
    cols = B * setup.n_actuators
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, stream=s_upd, start_steps=-1,
                             noise_seed=7, trajectory_length=(replay_steps * cols // B if use_replay else 1))
    agent.policy.act_noise = 0.3
    torch.cuda.synchronize()
    return pkg.TrainPipeline(env, agent, lag=lag, episode_steps=E, stream_env=s_env, stream_upd=s_upd, use_graphs=use_graphs,
                             chunks=chunks, noise_seed=99, use_replay=use_replay)


def _same_networks(pa, pb, names=("behavior_actor", "behavior_critic", "target_actor", "target_critic")):
    for n in names:
        for x, y in zip(getattr(pa.policy, n).model.params(), getattr(pb.policy, n).model.params()):
            assert np.array_equal(x, y), n


def test_replay_pushes_are_ordered_behind_the_env_step(pkg):
    """the stage pushes of step k copy the reward / terminal flags / next state that env_k writes.
    Pipeline A runs 30 steps (two episode boundaries, a replay of 8 steps that wraps three times) WITHOUT any host
    synchronisation; pipeline B -- same seeds -- drains the device after every step, which hides any missing
    stream order.  Both must leave the same four traces, counters and networks, bit for bit; and the newest rows
    of the reward / terminal / state traces must be the ring contents of the last step."""
    pa = _make(pkg, use_replay=True)
    pb = _make(pkg, use_replay=True)
    n = 30
    pa.run(n)
    for _ in range(n):
        pb.run(1)
        torch.cuda.synchronize()
    pa.sync(); pb.sync()
    ta, tb = pa.agent.trajectory, pb.agent.trajectory
    assert (ta.n_sa, ta.n_rt) == (tb.n_sa, tb.n_rt) and ta.n_rt == n * pa.cols
    for name in ("state", "action", "reward", "terminal"):
        assert torch.equal(getattr(ta, name), getattr(tb, name)), name
    _same_networks(pa, pb)
    assert torch.equal(pa.y, pb.y) and bool(torch.isfinite(pa.y).all())
    # newest rows = what the last env step produced (step n-1 is not the last of an episode: 30 = 17 + 13)
    k = n - 1
    lo = (ta.n_rt - pa.cols) % ta.capacity
    assert torch.equal(ta.reward[lo:lo + pa.cols], pa.rring[k % 3].view(-1))
    assert torch.equal(ta.terminal[lo:lo + pa.cols], pa.tring[k % 3].view(-1))
    # ... and the terminal row block of a time-out step is all ones: step 16 (last of episode 0) lies 13 steps back,
    # outside the 8-step replay, so check step 33 after four more steps
    pa.run(4); pa.sync()
    lo = (ta.n_rt - pa.cols) % ta.capacity
    assert float(ta.terminal[lo:lo + pa.cols].min()) == 1.0          # step 33 = 2 * 17 - 1
    pa.close(); pb.close()


def test_capture_skips_unreachable_ring_phases(pkg):
    """chunk 24 with 26-step episodes -- the interior offsets 1 .. E-1-c combined with E mod 6 never meet
    some ring phases, and the capture loop used to issue eager steps for ever looking for them.  Now those (chunk, phase)
    pairs are skipped, run() falls back to smaller chunks, and the run stays bit-identical to the eager pipeline."""
    pg = _make(pkg, True, E=26, chunks=(24, 6, 1))
    pe = _make(pkg, False, E=26)
    pg.run(5)
    pg.capture()                                       # returns
    assert pg._captured and 0 < len([1 for (c, _p) in pg.graphs if c == 24]) < 6
    assert len([1 for (c, _p) in pg.graphs if c == 6]) == 6
    pe.run(pg.tick)
    for n in (3, 26, 31):
        pg.run(n); pe.run(n)
    pg.sync(); pe.sync()
    assert pg.tick == pe.tick and pg.n_graph_launches > 0
    assert torch.equal(pg.y, pe.y)
    _same_networks(pg, pe)
    pg.close()


def test_changed_scalars_invalidate_recorded_steps_and_graphs(pkg):
    """act_noise, act_limit, the learning rates, gamma and rho are frozen into the recorded interior
    steps and the graphs.  A noise schedule (the reference decays act_noise between training loops,
    scripts/KS/setup/KSSetup.jl:304-319) must reach the kernels: after the change the pipeline with recorded steps /
    graphs equals the one issued through the Python layers every step."""
    pa = _make(pkg, True)
    pb = _make(pkg, False)
    pb.fast_eager = False
    pa.run(5); pa.capture()
    pb.run(pa.tick)
    pa.run(14); pb.run(14)
    for p in (pa, pb):
        p.policy.act_noise = 0.05
        p.policy.behavior_actor.optimizer.eta = 1e-4
    pa.run(20); pb.run(20)
    pa.sync(); pb.sync()
    assert not pa._captured and not pa.graphs and pa._progs       # dropped, interior steps re-recorded with the new values
    assert torch.equal(pa.y, pb.y)
    for k in range(3):
        assert torch.equal(pa.aring[k], pb.aring[k])
    _same_networks(pa, pb)
    pa.capture()
    pa.run(13); pb.run(pa.tick - pb.tick)
    pa.sync(); pb.sync()
    assert pa._captured and torch.equal(pa.y, pb.y)
    _same_networks(pa, pb)
    pa.close()


def test_torch_events_disable_the_recorded_step_replay(pkg, monkeypatch):
    """PDEC_TORCH_EVENTS=1 (diagnostic): the cross-stream records / waits are torch calls the recorded-call replay would
    not re-issue, so fast_eager is off for that event type and the run equals the default one"""
    monkeypatch.setenv("PDEC_TORCH_EVENTS", "1")
    pt = _make(pkg, False)
    monkeypatch.delenv("PDEC_TORCH_EVENTS")
    pe = _make(pkg, False)
    assert not pt.fast_eager and pe.fast_eager
    pt.run(25); pe.run(25)
    pt.sync(); pe.sync()
    assert not pt._progs and torch.equal(pt.y, pe.y)
    _same_networks(pt, pe)


def test_restart_in_the_middle_of_an_episode(pkg):
    """reset_from() at tick > 0 that does not follow a time-out step.  The transitions produced before
    the restart must not be trained on (the first step of the new episode overwrites the next-state slot of the last old
    transition): no update is issued until LAG steps of the new episode exist, recorded steps are dropped, and the
    recorded-step pipeline equals the one issued through the Python layers."""
    pa = _make(pkg, False)
    pb = _make(pkg, False)
    pb.fast_eager = False
    for p in (pa, pb):
        p.run(9)
        p.sync()
        before = [x.copy() for x in p.policy.behavior_critic.model.params()]
        p.reset_from(p.env.y0)
        assert p._first_tick == 9 and p.ep_start == 9 and not p._progs
        p.run(2)                                   # steps 9, 10: their updates (transitions 7, 8) are skipped
        p.sync()
        after = p.policy.behavior_critic.model.params()
        assert all(np.array_equal(x, y) for x, y in zip(before, after))
        p.run(12)
        p.sync()
        assert not all(np.array_equal(x, y) for x, y in zip(before, p.policy.behavior_critic.model.params()))
    assert torch.equal(pa.y, pb.y) and bool(torch.isfinite(pa.y).all())
    _same_networks(pa, pb)


def test_restart_in_the_middle_of_an_episode_on_the_replay_route(pkg):
    """the same restart with use_replay=True.  Transition tick-1 sits in the ring with terminal = 0 and the
    reset state becomes its next_state; reset_from() must cut it off (terminal = 1 on its rows), so that no sample bootstraps
    across the reset -- and only on ITS rows."""
    p = _make(pkg, use_replay=True, replay_steps=32)
    p.run(9); p.sync()
    tr = p.agent.trajectory
    assert tr.n_sa == tr.n_rt == 9 * p.cols
    before = tr.terminal.clone()
    assert float(before[8 * p.cols:9 * p.cols].max()) == 0.0          # step 8 is interior (E = 17): not terminal
    p.reset_from(p.env.y0)
    p.sync()
    assert float(tr.terminal[8 * p.cols:9 * p.cols].min()) == 1.0     # ... and now it is
    assert torch.equal(tr.terminal[:8 * p.cols], before[:8 * p.cols])
    p.run(12); p.sync()
    assert tr.n_sa == tr.n_rt == 21 * p.cols                          # the ring stays aligned: one (s, a) row per (r, t) row
    assert float(tr.terminal[8 * p.cols:9 * p.cols].min()) == 1.0
    assert float(tr.terminal[9 * p.cols:21 * p.cols].max()) == 0.0
    # the first row block of the new episode is the featurized reset state, i.e. the (cut-off) next_state of transition 8
    assert torch.equal(tr.state[9 * p.cols:10 * p.cols].view_as(p.state0), p.state0.to(tr.state.dtype))
    assert bool(torch.isfinite(p.y).all())
    p.close()


def test_split_update_sequence_on_one_rank_equals_the_fused_finish(pkg):
    """`bench.py --split-update`: the data-parallel launch sequence -- gradient pass -> slab reduction ->
    pdec_allreduce_grads on a ONE-rank RCCL communicator -> pdec_adam_polyak_step -- inside the two-stream pipeline leaves
    bit for bit the networks and fields of the fused single-GPU finish (an all-reduce over one rank is the identity)."""
    L = pkg._lib
    pa = _make(pkg, False)
    lib = pa.lib
    for sync in ("all", "policy"):
        red = pkg.distributed.NativeGradReducer(lib, rank=0, world_size=1, reduce_critic=(sync == "all"), force_split=True)
        assert red.active and red.verify_against_torch("cuda:0")
        setup = pkg.KSSetup.bench_C2(256)
        s_env, s_upd = torch.cuda.Stream(), torch.cuda.Stream()
        y0 = setup.generate_random_init(np.random.default_rng(0), 64) * 0.15
        env = pkg.PDEenv(setup, B=64, dtype=torch.float32, y0=y0, stream=s_env, autoreset=False)
        agent = pkg.create_agent(setup=setup, B=64, rng=np.random.default_rng(1), dtype=torch.float32, stream=s_upd, start_steps=-1,
                                 noise_seed=7, trajectory_length=1, reducer=red)
        agent.policy.act_noise = 0.3
        torch.cuda.synchronize()
        pb = pkg.TrainPipeline(env, agent, lag=2, episode_steps=17, stream_env=s_env, stream_upd=s_upd, use_graphs=False, noise_seed=99)
        assert pb.multi_rank and not pb.use_graphs
        pf = _make(pkg, False)
        pf.run(40); pb.run(40)
        pf.sync(); pb.sync()
        assert torch.equal(pf.y, pb.y) and bool(torch.isfinite(pb.y).all())
        _same_networks(pf, pb)
        pb.close(); pf.close()
        red.close()
    pa.close()


def _make_dp(pkg, red, serial=False, off_chain=None, B=64, E=17, nx=256, **agent_kw):
    """a data-parallel pipeline on ONE rank (split update sequence through `red`), with a third stream for the collective"""
    setup = pkg.KSSetup.bench_C2(nx)
    s_env = torch.cuda.Stream()
    s_upd = s_env if serial else torch.cuda.Stream()
    s_ar = torch.cuda.Stream()
    y0 = setup.generate_random_init(np.random.default_rng(0), B) * 0.15
    env = pkg.PDEenv(setup, B=B, dtype=torch.float32, y0=y0, stream=s_env, autoreset=False)
    agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, stream=s_upd, start_steps=-1,
                             noise_seed=7, trajectory_length=1, reducer=red, **agent_kw)
    agent.policy.act_noise = 0.3
    torch.cuda.synchronize()
    return pkg.TrainPipeline(env, agent, lag=2, episode_steps=E, stream_env=s_env, stream_upd=s_upd, use_graphs=False, noise_seed=99,
                             stream_ar=s_ar, ar_off_chain=off_chain)


@pytest.mark.parametrize("serial", [False, True])
def test_allreduce_off_the_update_chain_is_bit_identical(pkg, serial):
    """with the policy-gradient-only exchange and the reference's frozen target networks the
    all-reduce of the actor gradient and the ADAM launch behind it run on a third stream beside the NEXT critic half (which reads
    only the target actor, src/PDEagent.jl:385), and the next acting kernel / actor pass wait for the ADAM launch's event.  Same
    kernels, same arguments, same order of arithmetic: after 40 control steps (two episode boundaries, recorded-step replay)
    every network and the PDE state equal, bit for bit, (a) the same sequence with the collective ON the update stream and (b)
    the fused single-GPU finish -- through a 1-rank RCCL communicator and through bench.py's emulated-latency reducer (a 25 us
    spin kernel where the collective goes), on two streams and on one (config C3's issue order: critic half first)."""
    import bench
    lib = pkg._lib.load()
    for kind in ("rccl", "spin"):
        pf = _make(pkg, False)
        pf.run(40); pf.sync()
        mk = (lambda: pkg.distributed.NativeGradReducer(lib, rank=0, world_size=1, reduce_critic=False, force_split=True)) if kind == "rccl" \
            else (lambda: bench.EmulatedLatencyReducer(lib, 25.0))
        reds = [mk(), mk()]
        pc = _make_dp(pkg, reds[0], serial=serial, off_chain=False)
        ps = _make_dp(pkg, reds[1], serial=serial)                       # None = off the chain whenever legal
        assert pc.multi_rank and not pc.ar_off_chain and ps.ar_off_chain and ps.s_ar is not None and pc.s_ar is None
        pc.run(40); ps.run(40)
        pc.sync(); ps.sync()
        assert bool(ps._progs) and bool(pc._progs)                       # interior steps came from recorded call lists
        for p in (pc, ps):
            assert torch.equal(pf.y, p.y) and bool(torch.isfinite(p.y).all()), kind
            _same_networks(pf, p)
            for k in range(3):
                assert torch.equal(pf.aring[k], p.aring[k])
        # ... the order can be switched in a running pipeline (bench.py times both and keeps the faster)
        pc.set_ar_order(True); ps.set_ar_order(False)
        assert pc.ar_off_chain and not ps.ar_off_chain and not pc._progs
        pf.run(15); pc.run(15); ps.run(15)
        pf.sync(); pc.sync(); ps.sync()
        for p in (pc, ps):
            assert torch.equal(pf.y, p.y), kind
            _same_networks(pf, p)
        pc.set_ar_order(False); ps.set_ar_order(True)
        # ... and a restart in the middle of an episode (steps without an update while an apply is still in flight)
        for p in (pc, ps):
            p.reset_from(p.env.y0)
            p.run(9); p.sync()
        assert torch.equal(pc.y, ps.y)
        _same_networks(pc, ps)
        pc.close(); ps.close(); pf.close()
        for r in reds:
            r.close()


def test_allreduce_off_the_chain_is_refused_where_it_would_change_the_arithmetic(pkg):
    """the order is legal only when critic half_{k+1} cannot see ADAM(actor)_k: moving targets (the Polyak step of the target
    actor rides on that ADAM launch and the critic half reads the target actor) or an exchanged critic -> automatic selection
    keeps the collective on the chain, insisting raises"""
    lib = pkg._lib.load()
    red = pkg.distributed.NativeGradReducer(lib, rank=0, world_size=1, reduce_critic=False, force_split=True)
    p = _make_dp(pkg, red, quirk_frozen_targets=False)
    assert p.multi_rank and not p.ar_off_chain
    with pytest.raises(pkg.PdecError):
        _make_dp(pkg, red, off_chain=True, quirk_frozen_targets=False)
    red_all = pkg.distributed.NativeGradReducer(lib, rank=0, world_size=1, reduce_critic=True, force_split=True)
    assert not _make_dp(pkg, red_all).ar_off_chain
    with pytest.raises(pkg.PdecError):
        _make_dp(pkg, red_all, off_chain=True)
    # moving targets on the chain still equal the fused finish
    pm = _make_dp(pkg, red, quirk_frozen_targets=False)
    pm.run(12); pm.sync()
    assert bool(torch.isfinite(pm.y).all())
    p.close(); pm.close(); red.close(); red_all.close()


def test_checkpoint_keeps_the_device_noise_counter(pkg, tmp_path):
    """TrainPipeline's acting kernel advances a DEVICE-resident Philox counter
    (pdec_policy_act_rng_dev); save_agent stores it and load_agent restores it, so a resumed pipeline continues the
    exploration-noise stream instead of replaying it from 0.  Also: bit-generator states with arrays (MT19937) serialise."""
    pa = _make(pkg, False, B=8)
    pa.run(7); pa.sync()
    ctr = C.c_uint64()
    pkg._lib.check(pa.lib.pdec_noise_counter_get(pa.actor.handle, C.byref(ctr)))
    assert ctr.value == 7 * ((pa.cols * pa.na + 3) // 4) > 0
    pa.policy.rng = np.random.Generator(np.random.MT19937(3))
    path = str(tmp_path / "agent.npz")
    pkg.checkpoint.save_agent(path, pa.agent)
    pb = _make(pkg, False, B=8)
    pb.policy.rng = np.random.Generator(np.random.MT19937(4))
    pkg.checkpoint.load_agent(path, pb.agent)
    c2 = C.c_uint64()
    pkg._lib.check(pb.lib.pdec_noise_counter_get(pb.actor.handle, C.byref(c2)))
    assert c2.value == ctr.value
    assert pb.policy.rng.integers(0, 1 << 30) == np.random.Generator(np.random.MT19937(3)).integers(0, 1 << 30)
    _same_networks(pa, pb)


@pytest.mark.parametrize("nx", [256, 192])
def test_rk4_fd_step_hands_reward_partials_to_the_critic_pass(pkg, nx):
    """round 4: the RK4 + periodic-FD steps (one wave per trajectory at N = 256, the general form elsewhere) leave one reward
    sum per workgroup like the CNAB2 step, so the pipeline needs no pdec_reward_mean launch on the env stream for the
    reference's reward broadcast (quirk, src/PDEagent.jl:388-393): the partials add up to the rewards written, the pipeline
    uses them (rpart, stop events), and eager == graph replay bit for bit"""
    pe = _make(pkg, False, B=64, integrator="rk4_fd", nx=nx)
    pg = _make(pkg, True, B=64, integrator="rk4_fd", nx=nx)
    assert pe.rpart is not None and pe.n_rpart == 64 and pe.stop_events
    pe.run(9); pe.sync()
    k = pe.tick - 1
    tot, ref = float(pe.rpart[k % 3].double().sum()), float(pe.rring[k % 3].double().sum())
    assert abs(tot - ref) <= 1e-5 * abs(ref) and ref != 0.0
    per_traj = pe.rring[k % 3].double().sum(dim=1)
    assert torch.allclose(pe.rpart[k % 3].double(), per_traj, rtol=1e-5, atol=1e-6)
    pg.run(5); pg.capture()
    pe.run(pg.tick - pe.tick)
    pg.run(30); pe.run(30)
    pg.sync(); pe.sync()
    assert pg.n_graph_launches > 0 and torch.equal(pe.y, pg.y)
    _same_networks(pe, pg)
    assert bool(torch.isfinite(pe.y).all())
    pe.close(); pg.close()


@pytest.mark.parametrize("share", [False, True])
def test_full_size_c2_pipeline_properties(pkg, monkeypatch, share):
    """the exact instance the headline times -- config C2 at FULL size (KS N = 256, B = 512, fp32,
    3-layer nets, 32 768 update columns, two streams, LAG 2, register form (default) / SIMD-sharing form of the fused step) --
    through TrainPipeline for 64 control steps including one episode reset (E = 51): every field / network finite, the
    HIP-graph replay bit-identical to the eager issue, and the SIMD-sharing form tracking the register form (same
    arithmetic and order: <= 2e-6 per step on y, checked teacher-forced on the step kernel itself at B = 512)."""
    monkeypatch.setenv("PDEC_SHARE", "1" if share else "0")      # share: the 64-VGPR form of the step (round-2 default)
    pe = _make(pkg, False, B=512, E=51, chunks=(24, 6, 1))
    pg = _make(pkg, True, B=512, E=51, chunks=(24, 6, 1))
    assert pe.simd_sharing == share and pe.rpart is not None and pe.stop_events and pe.kick_env_after_critic
    pg.run(5); pg.capture()
    pe.run(pg.tick)
    n = 64 - pg.tick % 64 + 64
    pg.run(n); pe.run(n)
    pg.sync(); pe.sync()
    assert pg.n_graph_launches > 0 and pe.tick == pg.tick >= 64 + 51
    assert torch.equal(pe.y, pg.y) and torch.equal(pe.state, pg.state)
    for k in range(3):
        assert torch.equal(pe.aring[k], pg.aring[k]) and torch.equal(pe.rring[k], pg.rring[k])
    _same_networks(pe, pg)
    assert bool(torch.isfinite(pe.y).all()) and float(pe.y.abs().max()) < 50.0
    for n_ in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        assert all(np.isfinite(x).all() for x in getattr(pe.policy, n_).model.params())
    la, lc = pe.policy.losses()
    assert np.isfinite(la) and np.isfinite(lc)
    # SHARE vs register form of the very kernel instance, one step from the same state at B = 512
    env = pe.env
    pe.sync()
    y_in, act = pe.y.clone(), pe.aring[(pe.tick - 1) % 3].clone()
    outs = []
    for form in (True, False):
        env.set_simd_sharing(form)
        with torch.cuda.stream(pe.s_env):
            env.y.copy_(y_in)
            env(act.clone())
        pe.sync()
        outs.append(env.y.clone())
    env.set_simd_sharing(share)
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * max(1.0, float(outs[1].abs().max()))
    pg.close(); pe.close()


@pytest.mark.parametrize("B", [64, 512])
def test_round3_finish_kernel_is_bit_identical_to_the_round2_one(pkg, monkeypatch, B):
    """fused_finish_kernel was re-cut (one block per half chunk, 16-byte loads, 128 threads) for HBM
    throughput with the summation tree of every gradient element unchanged (src/custom_nna.jl:23-24, src/PDEagent.jl:415-417:
    ADAM in Float64, Polyak).  40 control steps of the training pipeline (one episode boundary; 32 and 256 gradient slabs)
    with the round-2 kernel (PDEC_FINISH_REF=1) and with the new one: all four networks, their ADAM moments, the losses and
    the fields bit-identical."""
    import ctypes as C
    monkeypatch.setenv("PDEC_FINISH_REF", "1")
    pa = _make(pkg, False, B=B, E=23)
    pa.run(40); pa.sync()
    la = pa.policy.losses()
    monkeypatch.delenv("PDEC_FINISH_REF")
    pb = _make(pkg, False, B=B, E=23)
    pb.run(40); pb.sync()
    _same_networks(pa, pb)
    assert torch.equal(pa.y, pb.y) and la == pb.policy.losses() and all(np.isfinite(la))
    for n in ("behavior_actor", "behavior_critic"):
        sa, sb = pkg.checkpoint._adam_state(getattr(pa.policy, n).model), pkg.checkpoint._adam_state(getattr(pb.policy, n).model)
        assert all(np.array_equal(x, y) for x, y in zip(sa, sb)), n


def _snapshot(p):
    red = p.pkg_distributed.GradReducer()
    out = [p.y.clone(), red._view(p.policy.behavior_critic.model).clone(), red._view(p.policy.behavior_actor.model).clone()]
    out += [x.clone() for x in p.aring] + [x.clone() for x in p.rring]
    for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        out += [torch.as_tensor(x.copy()) for x in getattr(p.policy, n).model.params()]
    return out


@pytest.mark.parametrize("n", [3, 7, 12, 19, 26, 41])
def test_results_do_not_depend_on_stream_timing(pkg, monkeypatch, n):
    """Two pipelines with the same seeds that differ only in TIMING must agree bit for bit: one issues its n control steps in
    one call (two streams, env step kicked behind the critic half, recorded-call replay), one is drained after every step, one
    reduces with the slower round-2 finish kernel.  Fields, action / reward rings, both flat gradient buffers and all four
    networks.  (The check the bf16-split passes of rounds 2 - 4 failed beside the PDE step, HISTORY.md §3.2a; they were deleted
    in round 5, the exact-f32 passes share their DMA / barrier helpers and pass it.)"""
    runs = []
    for mode in ("free", "drained", "finish_ref", "stamped"):
        if mode == "finish_ref":
            monkeypatch.setenv("PDEC_FINISH_REF", "1")
        p = _make(pkg, False, B=64, E=23)
        p.pkg_distributed = pkg.distributed
        if mode == "drained":
            for _ in range(n):
                p.run(1); torch.cuda.synchronize()
        elif mode == "stamped":
            # every critic pass in its STAMPED form (s_memtime reads and extra global stores between the phases that
            # wait on literal vmcnt counts for their LDS-DMA copies): a different instruction stream and timing, same data
            hc = p.policy.behavior_critic.model.handle
            for _ in range(n):
                pkg._lib.check(p.lib.pdec_debug_critic_stamps(hc, 1, None))
                p.run(1)
        else:
            p.run(n)
        p.sync()
        runs.append(_snapshot(p))
        p.close()
        monkeypatch.delenv("PDEC_FINISH_REF", raising=False)
    for other in runs[1:]:
        assert len(other) == len(runs[0]) and all(torch.equal(x, y) for x, y in zip(runs[0], other))


def test_a_request_for_the_deleted_bf16_split_passes_is_an_error(pkg, monkeypatch):
    """PDEC_SPLIT=a|c|1 used to select the experimental bf16-split forms of the fused passes (deleted in round 5, HISTORY.md
    §3.2a): the passes refuse the request instead of silently running exact f32"""
    setup = pkg.KSSetup.bench_C2(256)
    st = torch.cuda.Stream()
    y0 = setup.generate_random_init(np.random.default_rng(0), 8) * 0.15
    env = pkg.PDEenv(setup, B=8, dtype=torch.float32, y0=y0, stream=st, autoreset=False)
    agent = pkg.create_agent(setup=setup, B=8, rng=np.random.default_rng(1), dtype=torch.float32, stream=st, start_steps=-1,
                             noise_seed=7, trajectory_length=1)
    p = pkg.TrainPipeline(env, agent, lag=2, episode_steps=11, stream_env=st, stream_upd=st, use_graphs=False, noise_seed=99)
    monkeypatch.setenv("PDEC_SPLIT", "a")
    with pytest.raises(pkg.PdecError, match="no longer exist"):
        p.run(6)
    monkeypatch.setenv("PDEC_SPLIT", "0")
    p.close()
