"""Keller-Segel (1-D, 2 species, zero-flux edges) environment pieces, fp64 NumPy restatement.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED by tests/golden/kseg_hook.npz
(the reference integrates with OrdinaryDiffEq's adaptive RK4 at tol 1e-8, which cannot be
reproduced step-for-step; fixed classical RK4 with 32 sub-steps agrees to ~1e-8)."""
import numpy as np

from .julia_compat import circshift


class KSegConfig:
    """scripts/Keller-Segel/setup/KellerSegelSetup.jl:26-84 + Keller-Segel10_16.jl:8-14."""

    def __init__(self, nx=100, Lx=10.0, sensor_positions=None, actuators_to_sensors=None,
                 dt=0.006, te=8.0, agent_power=10.0, window_size=3, temporal_steps=2,
                 action_punish=0.0, delta_action_punish=0.0, half_window=2, max_value=20.0,
                 substeps=32):
        self.nx, self.Lx = int(nx), float(Lx)
        self.dx = self.Lx / self.nx
        self.sensor_positions = (np.arange(3, nx + 1, 5) if sensor_positions is None
                                 else np.asarray(sensor_positions, dtype=np.int64))
        self.actuators_to_sensors = (np.arange(3, 19) if actuators_to_sensors is None
                                     else np.asarray(actuators_to_sensors, dtype=np.int64))
        self.actuator_positions = self.sensor_positions[self.actuators_to_sensors - 1]
        self.dt, self.te, self.agent_power = dt, te, agent_power
        self.window_size, self.temporal_steps = window_size, temporal_steps
        self.action_punish, self.delta_action_punish = action_punish, delta_action_punish
        self.max_value = max_value                      # PDEenv default, src/PDEenv.jl:80
        self.substeps = substeps
        self.gaussians = prepare_rectangles(self, half_window)                 # :128
        self.gaussians_actuators = self.gaussians[self.actuators_to_sensors - 1]  # :129


def prepare_rectangles(cfg, half_window_size=2):
    """KellerSegelSetup.jl:112-126: box of ones on cells position-hw .. position+hw (1-based)."""
    out = []
    for position in cfg.sensor_positions:
        p = np.zeros(cfg.nx)
        p[position - half_window_size - 1: position + half_window_size] = 1.0   # :120
        out.append(p)
    return np.array(out)


def f(cfg, y, p):
    """RHS, KellerSegelSetup.jl:213-232 (stencil :63-64).  y is [2, nx] (row 0 = u, row 1 = v)."""
    dx = cfg.dx
    u, v = y[0].copy(), y[1].copy()
    um, up = circshift(u, 1), circshift(u, -1)                                # :217
    vm, vp = circshift(v, 1), circshift(v, -1)                                # :218
    um[0], up[-1] = u[0], u[-1]                                               # :220-221 zero-flux
    vm[0], vp[-1] = v[0], v[-1]                                               # :222-223
    du1 = -0.5 / dx * um + 0.5 / dx * up                                      # :225 row 1
    du2 = um / dx ** 2 - 2.0 / dx ** 2 * u + up / dx ** 2                     # row 2
    dv1 = -0.5 / dx * vm + 0.5 / dx * vp                                      # :226
    dv2 = vm / dx ** 2 - 2.0 / dx ** 2 * v + vp / dx ** 2
    vdot = dv2 - v + u + p                                                    # :228
    udot = du2 + u - 5.6 * du1 * dv1 - 5.6 * u * dv2 - u ** 2                 # :229
    return np.stack([udot, vdot])                                             # :231


def rk4_step(cfg, y, p, h):
    """Classical RK4 with the forcing frozen over the step (same tableau as
    src/fluid_rk4.jl:122-132 and OrdinaryDiffEq RK4())."""
    k1 = f(cfg, y, p)
    k2 = f(cfg, y + 0.5 * h * k1, p)
    k3 = f(cfg, y + 0.5 * h * k2, p)
    k4 = f(cfg, y + h * k3, p)
    return y + h / 6 * (k1 + 2 * (k2 + k3) + k4)


def do_step(cfg, y, p, substeps=None):
    """KellerSegelSetup.jl:234-239 answered with `substeps` fixed RK4 steps of dt/substeps."""
    n = cfg.substeps if substeps is None else substeps
    h = cfg.dt / n
    y = np.asarray(y, dtype=np.float64)
    for _ in range(n):
        y = rk4_step(cfg, y, p, h)
    return y


def do_step_midpoint(cfg, y, p, oversampling):
    """PDEenv's built-in integrator when no do_step is supplied (src/PDEenv.jl:208-214): explicit midpoint rule,
    `oversampling` sub-steps of dt/oversampling."""
    h = cfg.dt / oversampling
    y = np.asarray(y, dtype=np.float64)
    for _ in range(oversampling):
        y_old = y
        y = y + 0.5 * h * f(cfg, y, p)                                        # :211
        y = y_old + h * f(cfg, y, p)                                          # :212
    return y


def reward_function(cfg, y, action, delta_action):
    """KellerSegelSetup.jl:241-257."""
    a2s = cfg.actuators_to_sensors - 1
    sensors = (cfg.gaussians[a2s] @ (y[0] - 1.0)) ** 2 / 800                  # :248
    r = -np.abs(sensors)                                                      # :250
    a, da = np.asarray(action)[0, :], np.asarray(delta_action)[0, :]
    return r - cfg.action_punish * a ** 2 - cfg.delta_action_punish * da ** 2  # :257


def featurize(cfg, y, prev_state=None, action=None):
    """KellerSegelSetup.jl:265-316 (sees_action=false).  prev_state=None is the
    constructor/reset call (`isnothing(env)` branch: the fresh rows are repeated).  memory_size > 0 (cfg.memory_size, :303,
    :307-313; 0 in the shipped script): the last rows are rows 2.. of env.action (`action`), zeros in the reset form."""
    s1 = cfg.gaussians @ y[0] / 4                                             # :276
    s2 = cfg.gaussians @ y[1] / 4                                             # :277
    w = int(np.floor(cfg.window_size / 2))
    a2s = cfg.actuators_to_sensors - 1
    r1 = np.stack([circshift(s1, i) for i in range(-w, w + 1)])[:, a2s]       # :281-282
    r2 = np.stack([circshift(s2, i) for i in range(-w, w + 1)])[:, a2s]       # :283-284
    result = np.concatenate([r1, r2])                                         # :286
    if cfg.temporal_steps > 1:
        if prev_state is None:
            result = np.concatenate([result] * cfg.temporal_steps)           # :297-301
        else:
            keep = prev_state.shape[0] - result.shape[0] - int(getattr(cfg, "memory_size", 0))
            result = np.concatenate([result, prev_state[:keep]])              # :303
    m = int(getattr(cfg, "memory_size", 0))
    if m > 0:                                                                 # :307
        if action is None:
            result = np.concatenate([result, np.zeros((m, result.shape[1]))])  # :309
        else:
            result = np.concatenate([result, np.asarray(action, dtype=np.float64)[-m:, :]])   # :311
    return result


def prepare_action(cfg, action):
    """KellerSegelSetup.jl:318-332."""
    a = np.asarray(action, dtype=np.float64)[0, :]
    p = np.zeros(cfg.nx)
    for i in range(len(cfg.actuator_positions)):
        p = p + cfg.agent_power * a[i] * cfg.gaussians_actuators[i]
    return p
