"""The NN / DDPG / ADAM half pinned by what the reference's own artifacts hold about its TRAINING RUN (round 5).

scripts/KS/{KS22,KS200}/saves/agent.jld2 store, for each of the 8 parameter arrays of the behaviour actor and critic, Flux's
ADAM beta-power vector beta .^ (t + 1) as Float64 (tests/golden/ks*_agent_train.npz, extracted by make_golden.py).  Iterating
p <- p * beta in binary64 reproduces the stored bits at exactly t = 130 340: the beta2 power 0x1.d1cfb091bcab8p-189 and the beta1
power at its subnormal fixed point 5 * 2^-1074.  130 340 = 20 x 6 517: the reference's train() (scripts/KS/setup/KSSetup.jl:
304-319: 8 x run(agent, env, StopAfterEpisodeWithMinSteps(800), hook)) made 128 episodes x 51 control steps = 6 528 env steps,
and the update trigger (src/PDEagent.jl:342-361: length(traj) > update_after * n_actuators, update_freq = 1, update_loops = 20)
skipped the first 11.  A deterministic known answer for the run loop, the trigger, the replay bookkeeping and the Float64
beta-power bookkeeping of the update kernels; the KS22 buffer (never wrapped) additionally pins start_steps = 6 (zero actions),
the frame counts 52 224 / 52 232 and hook.rewards = sum over steps of mean over actuators of the stored rewards."""
import numpy as np
import pytest

from util import emulate_rlcore_wrap, load_golden, train

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _beta_powers(pkg, nna):
    import ctypes as C
    bp = (C.c_double * 2)()
    m = nna.model
    pkg._lib.check(m.lib.pdec_adam_get_state(m.handle, None, None, bp))
    return np.array([bp[0], bp[1]])


@pytest.mark.parametrize("which", ["ks22", "ks200"])
def test_training_run_reproduces_the_update_count_of_the_reference_agent(pkg, which):
    g = load_golden(f"{which}_agent_train.npz")
    setup = pkg.KSSetup.KS22() if which == "ks22" else pkg.KSSetup.KS200()
    A = int(g["n_actuators"])
    assert (setup.start_steps, setup.update_after, setup.update_freq, setup.update_loops, setup.batch_size) == (6, 10, 1, 20, 3)
    assert setup.n_actuators == A and setup.trajectory_length == int(g["capacity"])
    s_env, s_upd = torch.cuda.Stream(), torch.cuda.Stream()
    env = pkg.PDEenv(setup, B=1, dtype=torch.float64, stream=s_env)
    agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(7), noise_seed=7, stream=s_upd)
    hook = pkg.PDEhook(min_best_episode=1, use_random_init=True, init_seed=7)
    pol, tr = agent.policy, agent.trajectory
    assert pol.quirk_frozen_targets and pol.rho_effective == 1.0 and pol.p == 0.995
    targets0 = [x.copy() for n in (pol.target_actor, pol.target_critic) for x in n.params()]
    behaviour0 = [x.copy() for n in (pol.behavior_actor, pol.behavior_critic) for x in n.params()]
    train(pkg, agent, env, hook, loops=8, no_steps=800, decay=0.2)
    torch.cuda.synchronize()
    # the reference's target networks never move (tests/test_replay_golden.py::test_reference_target_networks_never_moved):
    # bit for bit the initial ones after 130 340 updates through the Polyak kernels with rho = 1
    targets1 = [x for n in (pol.target_actor, pol.target_critic) for x in n.params()]
    assert all(np.array_equal(a, b) for a, b in zip(targets0, targets1))
    assert all(np.array_equal(a, b) for a, b in zip(targets0, behaviour0))                  # force sync at creation (:76-77)
    assert all(np.abs(b).max() == 0 for b in targets1[1::2])
    assert len(hook.rewards) == 128 and np.isfinite(hook.rewards).all()
    assert tr.n_rt == 128 * 51 * A and tr.n_sa == tr.n_rt + A
    # ---- the ADAM step count, through the Float64 beta powers of the update kernels: bit for bit the reference's
    ref = g["adam_beta_pow"][0]
    for nna in (pol.behavior_actor, pol.behavior_critic):
        bp = _beta_powers(pkg, nna)
        assert np.array_equal(bp.view(np.uint64), ref.view(np.uint64)), (bp, ref)
    assert abs(pol.act_noise - 1.2 * 0.2 ** 8) < 1e-18
    if which == "ks200":
        assert len(tr) == int(g["capacity"]) == 150000 and int(g["nframes_rt"]) >= 1      # wrapped, like the reference's
        return
    # ---- KS22: the reference's buffer never wrapped; same frame counts, same structure
    assert tr.n_rt == int(g["n_rt"]) and tr.n_sa == int(g["n_sa"])
    a = tr.action[:tr.n_rt, 0].cpu().numpy().reshape(128, 51, A)
    ra = g["action"][:tr.n_rt].reshape(128, 51, A)
    zero, rzero = np.abs(a).max(axis=2) == 0, np.abs(ra).max(axis=2) == 0
    assert np.array_equal(zero, rzero) and zero[:, :6].all() and not zero[:, 6:].any()     # start_steps = 6, every episode
    t = tr.terminal[:tr.n_rt].cpu().numpy().reshape(128, 51, A)
    assert (t[:, :50] == 0).all() and (t[:, 50] == 1).all()
    # hook.rewards = sum_t mean_i r (src/PDEhook.jl:51-53) -- holds for the reference's own two files to 3.5e-8 ...
    rr = g["reward"].astype(np.float64).reshape(128, 51, A)
    assert np.abs(rr.mean(axis=2).sum(axis=1) - load_golden("ks22_hook.npz")["episode_rewards"]).max() < 1e-7
    # ... and for this run's trace and hook
    r = tr.reward[:tr.n_rt].cpu().numpy().astype(np.float64).reshape(128, 51, A)
    assert np.abs(r.mean(axis=2).sum(axis=1) - np.asarray(hook.rewards)).max() < 2e-6
    # exploration noise: 1.2 * 0.2^loop on top of the actor, clamped to +-1 (src/PDEagent.jl:201-204); the first loop of the
    # reference clips 43 % of its actions, the second 1.7 %
    clip = (np.abs(a[:, 6:]) == 1).reshape(8, -1).mean(axis=1)
    rclip = (np.abs(ra[:, 6:]) == 1).reshape(8, -1).mean(axis=1)
    assert abs(clip[0] - rclip[0]) < 0.06 and clip[1] < 0.12 and clip[2:].max() < 0.12, (clip, rclip)


@pytest.mark.parametrize("which", ["ks22", "ks22_one_stream", "ks22_early_ends", "keller_segel", "ks_three_layer", "fluid"])
def test_device_episodes_are_bit_identical_to_the_stage_loop(pkg, which):
    """VERDICT r5 item 6 (rows F1 + F2 composed; src/PDEagent.jl:175-209,237-361, src/PDEenv.jl:195-241): `run` enqueues the
    control steps of a whole episode without reading `is_terminated(env)` back after every step -- the device's halt flag
    (pdec_set_episode_halt) decides it, one read-back per episode -- and must leave EXACTLY what the stage loop leaves: the
    four networks, their ADAM moments and beta powers, the four replay traces and their counters, the Philox offsets, the
    environment's fields, hook.rewards and the best episode's rows.  KS22 (fp64 env, 2-layer nets, the register-resident
    20 x 3 update), the same on one stream, 3-layer nets (fused acting kernel), Keller-Segel (1 334-step episodes, two species,
    temporal stack), and KS22 with a blow-up threshold low enough that episodes END EARLY (max|y| > max_value,
    src/PDEenv.jl:226-240: the speculatively issued later steps must leave no trace)."""
    import ctypes as C
    run_mod = __import__("importlib").import_module(pkg.__name__ + ".run")

    def make():
        if which == "keller_segel":
            setup, loops, steps, decay = pkg.KellerSegelSetup(), 1, 2700, 0.6
        elif which == "fluid":
            setup, loops, steps, decay = pkg.FluidSetup(nx=64), 1, 700, 0.6
        elif which == "ks22_early_ends":
            setup, loops, steps, decay = pkg.KSSetup.KS22(max_value=4.0), 2, 400, 0.2
        elif which == "ks_three_layer":
            setup, loops, steps, decay = pkg.KSSetup.KS22(drop_middle_layer=False), 2, 160, 0.2
        else:
            setup, loops, steps, decay = pkg.KSSetup.KS22(), 2, 400, 0.2
        s_env = torch.cuda.Stream()
        s_upd = s_env if which == "ks22_one_stream" else torch.cuda.Stream()
        env = pkg.PDEenv(setup, B=1, dtype=torch.float64, stream=s_env)
        agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(7), noise_seed=7, stream=s_upd)
        hook = pkg.PDEhook(min_best_episode=1, use_random_init=True, init_seed=7)
        return setup, env, agent, hook, loops, steps, decay

    out = []
    for dev in (True, False):
        setup, env, agent, hook, loops, steps, decay = make()
        hook.use_random_init = True
        agent.policy.act_noise = setup.act_noise
        for _ in range(loops):
            stop = pkg.StopAfterEpisodeWithMinSteps(steps)
            assert run_mod.device_episodes_ok(agent, env, stop, hook)
            pkg.run(agent, env, stop, hook, device_episodes=dev)
            agent.policy.act_noise *= decay
        torch.cuda.synchronize()
        out.append((env, agent, hook))
    (ed, ad, hd), (es, as_, hs) = out
    pd, ps, td, ts = ad.policy, as_.policy, ad.trajectory, as_.trajectory
    assert len(hd.rewards) == len(hs.rewards) >= 3
    if which == "ks22_early_ends":
        full = len(hd.rewards) * run_mod._episode_steps(ed) * setup.state_shape[1]
        assert td.n_rt < full, "no episode ended early: the halt path was not exercised"
        assert hd.rewards_compare and len(hd.rewards_compare) < len(hd.rewards)      # some ran to te, some did not
    assert (td.n_sa, td.n_rt, pd.update_step, pd._noise_off, pd._sample_off) == (ts.n_sa, ts.n_rt, ps.update_step, ps._noise_off, ps._sample_off)
    for name in ("state", "action", "reward", "terminal"):
        assert torch.equal(getattr(td, name), getattr(ts, name)), name
    for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        md, ms = getattr(pd, n).model, getattr(ps, n).model
        for x, y in zip(md.params(), ms.params()):
            assert np.array_equal(x, y), n
        for m_ in (md, ms):
            pass
        assert np.array_equal(_beta_powers(pkg, getattr(pd, n)).view(np.uint64), _beta_powers(pkg, getattr(ps, n)).view(np.uint64)), n
    assert torch.equal(ed.y, es.y) and torch.equal(ed.state, es.state) and (ed.steps, ed.time) == (es.steps, es.time)
    assert np.array_equal(np.asarray(hd.rewards), np.asarray(hs.rewards)), np.abs(np.asarray(hd.rewards) - np.asarray(hs.rewards)).max()
    assert (hd.bestepisode, hd.bestreward, len(hd.bestDF)) == (hs.bestepisode, hs.bestreward, len(hs.bestDF))
    for rd, rs in zip(hd.bestDF, hs.bestDF):
        assert rd["timestep"] == rs["timestep"]
        for k in ("action", "p", "y", "reward"):
            assert np.array_equal(rd[k], rs[k]), k
    for x, y in zip(hd.bestNNA.model.params(), hs.bestNNA.model.params()):
        assert np.array_equal(x, y)


# ---------------------------------------------------------------------------------------------------------------------------
# Learning curves under the reference's train() hyper-parameters, against the reference's own hook.rewards.
# One saved run per experiment is ONE draw of a noisy process, so the comparison is a band over seeds (set from
# tools/train_curve_probe.py / train_seed_sweep.py runs, HISTORY.md round 5), and each band has a negative control: the other
# setting of the target-network switch must miss it.  What the bands established:
#   * KS22 / KS200 (artifacts written by Julia 1.9.1): reproduced only with FROZEN target networks -- what src/PDEagent.jl:415-417
#     does with src/custom_nna.jl:20 as committed, and what agent.jld2 holds (zero target biases).  With the Polyak loop as
#     written (rho = 0.995) no seed gets within a factor of 3 of the reference curve.
#   * Fluid_8 and the first training loop of Keller-Segel (artifacts written by Julia 1.9.4 -- another session of the authors):
#     reproduced only with MOVING targets (rho = 0.995); frozen targets leave Keller-Segel at the saturated-actor return of
#     -30 in 24 of 24 seeds.  Those runs evidently did not go through the committed custom_nna.jl.
def _curves(pkg, setup, seeds, frozen, loops, no_steps, decay, hook_kw=None, rlcore_wrap=False):
    out = []
    s_env, s_upd = torch.cuda.Stream(), torch.cuda.Stream()
    for seed in seeds:
        env = pkg.PDEenv(setup, B=1, dtype=torch.float64, stream=s_env)
        agent = pkg.create_agent(setup=setup, B=1, rng=np.random.default_rng(100 + seed), noise_seed=1000 + seed, stream=s_upd,
                                 quirk_frozen_targets=frozen)
        if rlcore_wrap:                      # RLCore's misaligned traces after wrap-around (tests/util.py), host sampling
            emulate_rlcore_wrap(pkg, agent)
        hook = pkg.PDEhook(min_best_episode=1, use_random_init=True, init_seed=2000 + seed, init_rng=np.random.default_rng(seed),
                           **(hook_kw or {}))
        train(pkg, agent, env, hook, loops=loops, no_steps=no_steps, decay=decay)
        torch.cuda.synchronize()
        out.append((np.asarray(hook.rewards), hook.bestreward))
    return out


def _ks_bands(r, best):
    first, mid, last = r[:16].mean(), r[16:32].mean(), r[-16:].mean()
    return (-18.0 <= first <= -4.0 and r[:16].min() >= -40.0, -2.8 <= mid <= -1.1, -2.8 <= last <= -0.9, -1.05 <= best <= -0.5)


def test_ks22_learning_curve_lands_in_the_band_of_the_reference_run(pkg):
    """scripts/KS/KS22/saves/hook.jld2 `rewards` (128 episodes of train(; loops = 8), KSSetup.jl:304-319): first 16 episodes
    (act_noise 1.2) -11.0 ... -3.3, mean -6.0; episodes 17-32 (noise 0.24) mean -1.67; last 16 mean -1.37; best -0.728"""
    ref = load_golden("ks22_hook.npz")["episode_rewards"]
    assert all(_ks_bands(ref, ref.max())) and len(ref) == 128           # the reference's own run sits in the band
    runs = _curves(pkg, pkg.KSSetup.KS22(), range(5), True, 8, 800, 0.2)
    ok = [_ks_bands(r, b) for r, b in runs]
    assert all(len(r) == 128 for r, _ in runs)
    assert sum(all(o) for o in ok) >= 4, (ok, [np.round(r[[0, 15, 16, 31, -1]], 2) for r, _ in runs])
    med = np.median([r[16:].mean() for r, _ in runs])
    assert abs(med - ref[16:].mean()) < 0.45, (med, ref[16:].mean())     # -1.4 in the reference
    # negative control: the Polyak loop as written (moving targets) does not reproduce this artifact
    runs = _curves(pkg, pkg.KSSetup.KS22(), range(3), False, 8, 800, 0.2)
    assert sum(all(_ks_bands(r, b)) for r, b in runs) == 0 and np.median([r[16:].mean() for r, _ in runs]) < -4.0


def test_ks200_learning_curve_before_the_buffer_wraps(pkg):
    """scripts/KS/KS200/saves/hook.jld2: same train(), 80 actuators; the 150 000-row buffer wraps in episode 37, after which
    RLCore's traces are misaligned (tests/util.py: rlcore_wrap_shift) and the reference curve relapses (-23.6 at episode 43, -26.2
    at 88); the product keeps its traces aligned, so only the episodes before the wrap are compared"""
    ref = load_golden("ks200_hook.npz")["episode_rewards"]
    band = lambda r: (-20.0 <= r[:16].mean() <= -4.0, -2.8 <= r[16:36].mean() <= -1.2)
    assert all(band(ref))
    runs = _curves(pkg, pkg.KSSetup.KS200(), range(3), True, 8, 800, 0.2)
    assert sum(all(band(r)) for r, _ in runs) >= 2, [np.round(r[:36], 1) for r, _ in runs]
    # aligned traces: no relapse after the wrap (the reference's worst episode after 36 is -26.2)
    assert np.median([r[36:].min() for r, _ in runs]) > -8.0
    # ... and with RLCore's buffer emulated -- (s, a, s') taken A - 1 rows later than (r, t) once the traces have wrapped -- this path
    # relapses after episode 37 like the reference's run (probe: 4 of 4 seeds, worst episodes -13 ... -21)
    assert ref[36:].min() < -20.0 and ref[:36].min() > -21.0
    wrapped = _curves(pkg, pkg.KSSetup.KS200(), range(3), True, 8, 800, 0.2, rlcore_wrap=True)
    assert sum(all(band(r)) for r, _ in wrapped) >= 2                      # before the wrap: the same curve
    assert np.median([r[36:].min() for r, _ in wrapped]) < -8.0, [np.round(r[36:].min(), 1) for r, _ in wrapped]


@pytest.mark.slow
@pytest.mark.parametrize("which,loops,seeds", [("fluid8", 10, 3), ("fluid16", 6, 2), ("fluid32", 5, 2)])
def test_fluid_learning_curves_need_moving_targets(pkg, which, loops, seeds):
    """scripts/Fluid/Fluid_{8,16,32}/saves/hook.jld2 `rewards`: 20 / 12 / 10 episodes of train(; loops = 10 / 6 / 5)
    (FluidSetup.jl:541-556; 128 x 128 grid; 8 x 8, 16 x 16, 32 x 32 sensors and actuators with kernel variances 0.08 / 0.04 / 0.022):
    Fluid_8 -6.26, -4.42, -2.64, -2.27, then -1.4 ... -0.4 (best -0.381); Fluid_16 -7.97 ... -1.08; Fluid_32 -7.84 ... -1.69.
    The only reference-held data that depend on the fluid path (pseudo-spectral right-hand side, RK4, sensing, actuation,
    reward): a wrong step would not learn this curve.  (The reference integrated with the adaptive do_step2 at tol = 1; this
    path with the fixed-step do_step, so the bands are wide.)  Probe, moving targets: 4 of 4 seeds in band for each of the
    three; frozen targets: 2 of 6 / 2 of 3 / 1 of 3 (the others diverge to -300 or NaN)."""
    ref = load_golden(f"{which}_hook.npz")["episode_rewards"]
    bands = {
        "fluid8": lambda r: (-14.0 <= r[0] <= -3.0, -3.2 <= r[2:6].mean() <= -1.2, -1.6 <= r[-8:].mean() <= -0.35, r.max() >= -0.95),
        "fluid16": lambda r: (-14.0 <= r[0] <= -3.0, -3.6 <= r[2:6].mean() <= -1.6, -2.0 <= r[-6:].mean() <= -0.9, r.max() >= -1.6),
        "fluid32": lambda r: (-16.0 <= r[0] <= -3.0, -6.0 <= r[2:6].mean() <= -1.8, -3.5 <= r[-4:].mean() <= -1.4, r.max() >= -2.2),
    }
    band = bands[which]
    n_ep = {"fluid8": 20, "fluid16": 12, "fluid32": 10}[which]
    assert len(ref) == n_ep and all(band(ref))
    setup = getattr(pkg.FluidSetup, {"fluid8": "Fluid_8", "fluid16": "Fluid_16", "fluid32": "Fluid_32"}[which])()
    runs = _curves(pkg, setup, range(seeds), False, loops, 580, 0.6)
    assert all(len(r) == n_ep for r, _ in runs)
    assert sum(all(band(r)) for r, _ in runs) >= seeds - 1, [np.round(r, 2) for r, _ in runs]


@pytest.mark.slow
def test_keller_segel_first_training_loop_needs_moving_targets(pkg):
    """scripts/Keller-Segel/Keller-Segel10_16/saves/hook.jld2 `rewards` (tests/golden/kseg_train.npz), first loop of train()
    (4 episodes x 1 334 steps at act_noise 1.2, KellerSegelSetup.jl:390-406): -7.09, -2.13, -2.77, -1.97.  A constant saturated
    action returns -30, zero action -10.  With moving targets this path learns a controller inside the first loop like the
    reference; with frozen targets the actor saturates (median -29).  Only the first loop is compared: over the 13 loops the
    reference run holds -1 ... -3 where 5 of 5 seeds of this path relapse at some point (HISTORY.md round 5, unexplained)."""
    ref = load_golden("kseg_train.npz")["episode_rewards"]
    assert len(ref) == 53 and -12.0 <= ref[0] <= -4.0 and ref[1:4].max() >= -2.2
    setup = pkg.KellerSegelSetup()
    moving = _curves(pkg, setup, range(5), False, 1, 5000, 0.6)
    frozen = _curves(pkg, setup, range(3), True, 1, 5000, 0.6)
    best_m = np.array([r[1:4].max() for r, _ in moving])
    best_f = np.array([r[1:4].max() for r, _ in frozen])
    assert all(len(r) >= 4 for r, _ in moving + frozen)
    assert np.median(best_m) >= -5.5 and best_m.max() >= -3.0, best_m
    assert np.median(best_f) <= -12.0, best_f
