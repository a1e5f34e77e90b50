"""KS experiment configuration: the globals of scripts/KS/setup/KSSetup.jl:20-77 (and the
mono twin KSglobalSetup.jl) plus the per-experiment script values (scripts/KS/KS22/KS22.jl:2-21),
turned into the tables libpdeconv needs.  Setup-time host code; the per-step closures
(do_step / featurize / prepare_action / reward_function) run on the GPU."""
import numpy as np

from .. import _lib
from ..julia_compat import float_range


def prepare_gaussians(nx, Lx, positions, sigma, norm_mode):
    """scripts/KS/setup/KSSetup.jl:82-109 (note the reference's precedence: the exponent is
    -(t^2/2*sigma^2), the prefactor 1/sqrt(2*pi*sigma)); tails wrapped periodically."""
    dx = Lx / nx
    extra = 50
    t = float_range(dx - extra * dx, dx, Lx + extra * dx)
    out = np.empty((len(positions), nx))
    for i, position in enumerate(positions):
        p = (1.0 / np.sqrt(2 * np.pi * sigma)) * np.exp(-(((t - position * dx) * 1) ** 2 / 2 * sigma ** 2))
        p = p / p.sum() if norm_mode == 1 else p / p.max()
        pleft, pright = p[:extra], p[extra + nx:]
        q = p[extra:extra + nx].copy()
        q[nx - len(pleft):] += pleft
        q[:len(pright)] += pright
        out[i] = q
    return out


def _refuse_unbuilt_branches(setup, where, memory_built=False, reward_check_with_memory=False):
    """memory_size > 0 (action memory: src/PDEagent.jl:201 + the featurize branch cited in `where`) is built for the 1-D
    per-actuator environments (KS, Keller-Segel: the composed env step, csrc/env.hip env_step_composed) and the fluid (its
    sensing kernels); elsewhere -- the global agent, the 2-D Keller-Segel grid -- and a temporal stack for the mono / global agent are optional branches no shipped
    script sets; a caller who sets them gets an error, not silence"""
    if setup.memory_size < 0 or (setup.memory_size != 0 and not memory_built):
        raise _lib.PdecError(f"memory_size = {setup.memory_size}: the action-memory branch of featurize / the acting path "
                             f"({where}, src/PDEagent.jl:201) is not built for this setup; only memory_size = 0 is supported")
    if setup.memory_size != 0 and getattr(setup, "check_max_value", "y") == "reward" and not reward_check_with_memory:
        raise _lib.PdecError("memory_size > 0 with check_max_value = 'reward' is not built (the composed env step tests 'y' only)")
    if setup.temporal_steps < 1 or (getattr(setup, "mono", False) and setup.temporal_steps != 1):
        raise _lib.PdecError(f"temporal_steps = {setup.temporal_steps} is not supported for this setup ({where})")


class KSSetup:
    # Which target-network regime reproduces the reference's saved runs of this experiment family (create_agent's default for
    # `quirk_frozen_targets`): KS22 / KS200 were produced with src/custom_nna.jl:20 as committed -- the Polyak loop of
    # src/PDEagent.jl:415-417 iterates over an empty parameter list, the targets in agent.jld2 are the initial ones
    # (tests/test_replay_golden.py) -- and their learning curves are reproduced only with frozen targets (HISTORY.md 5.1).
    reproduces_reference_with = "frozen"

    def __init__(self, nx, Lx, sensor_positions, actuator_positions=None, actuators_to_sensors=None,
                 sigma_sensors=1.0, sigma_actuators=1.0, mu=0.0, te=5.0, t0=0.0, dt=0.1, oversampling=30,
                 max_value=30.0, check_max_value="y", agent_power=7.5, action_punish=0.002,
                 delta_action_punish=0.002, window_size=1, mono=False,
                 nna_scale=0.6, nna_scale_critic=7.0, drop_middle_layer=True,
                 gamma=0.99, rho=0.995, batch_size=3, start_steps=6, update_after=10, update_freq=1,
                 update_loops=20, learning_rate=0.0005, learning_rate_critic=0.001, act_limit=1.0,
                 act_noise=1.2, trajectory_length=150_000, integrator="cnab2", temporal_steps=1, memory_size=0):
        self.nx, self.Lx = int(nx), float(Lx)
        self.dx = self.Lx / self.nx
        # "cnab2": the reference's spectral CNAB2 step (KSSetup.jl:130-160); "rk4_fd": RK4 + periodic 5-point FD variant
        self.integrator = integrator
        self.sensor_positions = np.asarray(sensor_positions, dtype=np.int64)
        self.actuator_positions = (self.sensor_positions if actuator_positions is None
                                   else np.asarray(actuator_positions, dtype=np.int64))
        n_act = len(self.actuator_positions)
        self.actuators_to_sensors = (np.arange(1, n_act + 1) if actuators_to_sensors is None
                                     else np.asarray(actuators_to_sensors, dtype=np.int64))  # 1-based
        self.sigma_sensors, self.sigma_actuators, self.mu = sigma_sensors, sigma_actuators, mu
        self.te, self.t0, self.dt, self.oversampling = te, t0, dt, int(oversampling)
        self.max_value, self.check_max_value = max_value, check_max_value
        self.agent_power = agent_power
        self.action_punish, self.delta_action_punish = action_punish, delta_action_punish
        self.window_size, self.mono = window_size, mono
        self.nna_scale, self.nna_scale_critic, self.drop_middle_layer = nna_scale, nna_scale_critic, drop_middle_layer
        self.gamma, self.rho, self.batch_size = gamma, rho, batch_size
        self.start_steps, self.update_after, self.update_freq, self.update_loops = \
            start_steps, update_after, update_freq, update_loops
        self.learning_rate, self.learning_rate_critic = learning_rate, learning_rate_critic
        self.act_limit, self.act_noise, self.trajectory_length = act_limit, act_noise, trajectory_length
        # featurize's optional branches (KSSetup.jl:209-226): temporal stacking runs in the step kernels' general
        # featurize path; the action-memory rows (memory_size > 0: extra actor outputs fed back as state rows,
        # src/PDEagent.jl:201) go through the composed env step (four launches; per-actuator agents only)
        self.temporal_steps, self.memory_size, self.n_species = int(temporal_steps), int(memory_size), 1
        _refuse_unbuilt_branches(self, "scripts/KS/setup/KSSetup.jl:209-226", memory_built=not mono)
        self.gaussians = prepare_gaussians(self.nx, self.Lx, self.sensor_positions, sigma_sensors, 1)
        if mono:   # KSglobalSetup.jl:99-102,125
            self.gaussians_actuators = prepare_gaussians(self.nx, self.Lx, self.actuator_positions, sigma_actuators, 2)
        else:      # KSSetup.jl:112-113
            self.gaussians_actuators = prepare_gaussians(self.nx, self.Lx, self.sensor_positions,
                                                         sigma_actuators, 2)[self.actuators_to_sensors - 1]

    # shapes of the RL.jl-facing arrays (per trajectory, Julia shapes)
    @property
    def n_actuators(self):
        return len(self.actuator_positions)

    @property
    def n_sensors(self):
        return len(self.sensor_positions)

    @property
    def state_shape(self):          # size(state_space)
        return (self.n_sensors, 1) if self.mono else (self.window_size * self.temporal_steps + self.memory_size, self.n_actuators)

    @property
    def action_shape(self):         # size(action_space)
        return (self.n_actuators,) if self.mono else (1 + self.memory_size, self.n_actuators)

    @property
    def reward_len(self):
        return 1 if self.mono else self.n_actuators

    @property
    def y_shape(self):
        return (self.nx,)

    def y0_standard(self):
        """y0_1D_standard, KSSetup.jl:53"""
        return np.array([0.5 if 4 <= i <= 44 else 0.0 for i in range(1, self.nx + 1)])

    def generate_random_init(self, rng, B=1):
        """KSSetup.jl:288-298, batched; rng = numpy Generator"""
        number_sin = 8
        xx = self.dx * np.arange(1, self.nx + 1)
        a = rng.uniform(-1, 1, (B, number_sin))
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        y0 = np.zeros((B, self.nx))
        for i in range(1, number_sin + 1):
            y0 += a[:, i - 1:i] * np.sin(i * xx / (2 * np.pi))[None, :]
        return y0 * 30 / np.linalg.norm(y0, axis=1, keepdims=True)

    def env_cfg(self, B, dtype_code):
        c = _lib.EnvCfg()
        # "cnab2" (the scripts' do_step), "rk4_fd" (north-star variant on the KSSetup.jl:55-59 stencil table) or
        # "midpoint_fd" (the same right-hand side under PDEenv's built-in explicit-midpoint integrator, src/PDEenv.jl:208-214)
        kind = _lib.PDE_KS_RK4_FD if self.integrator in ("rk4_fd", "midpoint_fd") else _lib.PDE_KS_CNAB2
        c.integrator = 1 if self.integrator == "midpoint_fd" else 0
        c.memory_size = self.memory_size
        c.pde_kind, c.dtype, c.B, c.N, c.n_species = kind, dtype_code, B, self.nx, 1
        c.S, c.A, c.window, c.temporal_steps, c.mono = self.n_sensors, self.n_actuators, self.window_size, self.temporal_steps, int(self.mono)
        c.K = self.oversampling
        c.check_max_value = {"y": 1, "reward": 2}.get(self.check_max_value, 0)
        c.Lx, c.dt, c.mu, c.max_value = self.Lx, self.dt, (0.0 if self.mono else self.mu), self.max_value
        c.sensor_scale = 1.0 / self.max_value                      # KSSetup.jl:201
        c.agent_power = self.agent_power                           # :241
        c.reward_in_scale, c.reward_offset = 6.0, 0.0              # :163
        c.reward_power, c.reward_denom = 1.3, self.max_value * 3   # :169
        c.action_punish, c.delta_action_punish = self.action_punish, self.delta_action_punish
        return c

    def tables(self):
        return (np.ascontiguousarray(self.gaussians, dtype=np.float64),
                np.ascontiguousarray(self.gaussians_actuators, dtype=np.float64),
                np.ascontiguousarray(self.actuators_to_sensors - 1, dtype=np.int32))

    # ---- shipped experiments (scripts/KS/*/*.jl)
    @classmethod
    def KS22(cls, **kw):
        return cls(192, 22.0, np.arange(1, 193, 24), sigma_sensors=0.7, sigma_actuators=0.7, **kw)

    @classmethod
    def KS200(cls, **kw):
        return cls(240, 200.0, np.arange(1, 241, 3), sigma_sensors=1.0, sigma_actuators=1.0, **kw)

    @classmethod
    def KS500(cls, **kw):
        return cls(600, 500.0, np.arange(1, 601, 3), sigma_sensors=1.0, sigma_actuators=1.0, **kw)

    @classmethod
    def KS22_global(cls, **kw):
        kw.setdefault("nna_scale", 4.8)
        kw.setdefault("nna_scale_critic", 56.0)
        kw.setdefault("trajectory_length", 700_000)
        return cls(192, 22.0, np.arange(1, 193, 24), sigma_sensors=0.7, sigma_actuators=0.7, mono=True, **kw)

    @classmethod
    def bench_C2(cls, nx=256, **kw):
        """KS200 geometry (dx = 200/240) at nx cells, sensors =
        actuators every 4 cells, window 3, 3-layer nets (h=16, H=140)."""
        kw.setdefault("window_size", 3)
        kw.setdefault("nna_scale", 1.6)
        kw.setdefault("nna_scale_critic", 7.0)
        kw.setdefault("drop_middle_layer", False)
        return cls(nx, nx * (200.0 / 240.0), np.arange(1, nx + 1, 4), sigma_sensors=1.0, sigma_actuators=1.0, **kw)
