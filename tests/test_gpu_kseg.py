"""GPU parity of the Keller-Segel path: RHS bit-level vs oracle, RK4 control step vs the
reference's golden (adaptive RK4 at tol 1e-8 -> fixed 32 sub-steps agree to <= 1e-7)."""
import numpy as np
import pytest

from util import load_golden, to_dev

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _mem(y):   # Julia [2, nx] -> memory [nx, 2]
    return np.ascontiguousarray(np.swapaxes(y, -1, -2))


def test_rhs_matches_oracle(pkg):
    from oracle import keller_segel as kg
    g = load_golden("kseg_hook.npz")
    setup, cfg = pkg.KellerSegelSetup(), kg.KSegConfig()
    B = 16
    env = pkg.PDEenv(setup, B=B, dtype=torch.float64)
    y, p = g["y_t"][:B], g["p_t1"][:B]
    out = env.rhs(to_dev(_mem(y), torch.float64), to_dev(p, torch.float64)).cpu().numpy()
    for b in range(B):
        ref = kg.f(cfg, y[b], p[b])
        assert np.abs(np.swapaxes(out[b], 0, 1) - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("prec,tol", [("f64", 1e-7), ("f32", 5e-5)])
def test_do_step_matches_reference_golden(pkg, prec, tol):
    g = load_golden("kseg_hook.npz")
    dt = torch.float64 if prec == "f64" else torch.float32
    B = len(g["idx"])
    env = pkg.PDEenv(pkg.KellerSegelSetup(), B=B, dtype=dt)
    out, flags = env.do_step(to_dev(_mem(g["y_t"]), dt), to_dev(g["p_t1"], dt))
    err = np.abs(np.swapaxes(out.cpu().numpy().astype(np.float64), 1, 2) - g["y_t1"]).max()
    assert err <= tol, err
    assert int(flags.sum()) == 0


def test_env_step_fused(pkg):
    from oracle import keller_segel as kg
    g = load_golden("kseg_hook.npz")
    setup, cfg = pkg.KellerSegelSetup(), kg.KSegConfig()
    dt = torch.float64
    B = 40
    env = pkg.PDEenv(setup, B=B, dtype=dt)
    env.y.copy_(to_dev(_mem(g["y_t"][:B]), dt))
    prev_state = np.stack([kg.featurize(cfg, g["y_t"][b], None) for b in range(B)])   # [B, ns, A]
    env.state.copy_(to_dev(np.swapaxes(prev_state, 1, 2), dt))
    env.action.copy_(to_dev(g["action_t"][:B], dt).reshape(env._ashape))
    env(to_dev(g["action_t1"][:B], dt).reshape(env._ashape))
    assert np.abs(env.p.cpu().numpy() - g["p_t1"][:B]).max() <= 1e-12
    assert np.abs(np.swapaxes(env.y.cpu().numpy(), 1, 2) - g["y_t1"][:B]).max() <= 1e-7
    assert np.abs(env.reward.cpu().numpy() - g["reward_t1"][:B]).max() <= 1e-9
    for b in (0, 7, 39):
        ynew = np.swapaxes(env.y[b].cpu().numpy(), 0, 1)
        st = kg.featurize(cfg, ynew, prev_state[b])
        assert np.abs(env.state[b].cpu().numpy().T - st).max() <= 1e-12


def test_reset_form_featurize(pkg):
    from oracle import keller_segel as kg
    setup, cfg = pkg.KellerSegelSetup(), kg.KSegConfig()
    env = pkg.PDEenv(setup, B=2, dtype=torch.float64)
    y0 = setup.y0_standard()
    st = kg.featurize(cfg, y0, None)
    assert np.abs(env.state[1].cpu().numpy().T - st).max() <= 1e-13


def test_builtin_midpoint_integrator(pkg):
    """PDEenv's own integrator when no do_step closure is given (src/PDEenv.jl:208-214): explicit midpoint rule with
    `oversampling` sub-steps on the Keller-Segel right-hand side"""
    from oracle import keller_segel as kg
    g = load_golden("kseg_hook.npz")
    K = 8
    setup, cfg = pkg.KellerSegelSetup(integrator="midpoint", substeps=K), kg.KSegConfig()
    B = 6
    env = pkg.PDEenv(setup, B=B, dtype=torch.float64)
    y, p = g["y_t"][:B], g["p_t1"][:B]
    out, _ = env.do_step(to_dev(_mem(y), torch.float64), to_dev(p, torch.float64))
    for b in range(B):
        ref = kg.do_step_midpoint(cfg, y[b], p[b], K)
        assert np.abs(np.swapaxes(out[b].cpu().numpy(), 0, 1) - ref).max() <= 1e-12
        assert np.abs(ref - g["y_t1"][b]).max() <= 1e-3        # second order: near (5e-4), not at, the adaptive solution
