"""Keller-Segel on a 2-D grid -- BASELINE.json configs[3] ("scripts/Keller-Segel: 2D 256x256 grid, batch=128,
2D conv actor").  The reference's Keller-Segel is 1-D (scripts/Keller-Segel/setup/KellerSegelSetup.jl); this
setup applies its rules along both axes (csrc/kseg2d.hip): same constants (:26-84), 5x5 boxes of ones as sensor /
actuator kernels (:112-129), 3x3 circular window over the sensor grid x 2 species x temporal_steps as the im2col
state (`Dense` over its columns = 2-D convolution with circular padding and stride = sensor spacing), actuators
on the sensors that are at least `border` sensors away from the edge (the 1-D script uses sensors 3..18 of 20)."""
import numpy as np

from .. import _lib


class KellerSegel2DSetup:
    # create_agent's default for `quirk_frozen_targets`: the Keller-Segel family learns only with moving targets (KellerSegelSetup)
    reproduces_reference_with = "moving"

    is_kseg2d = True

    def __init__(self, nx=256, ny=256, Lx=None, sensor_step=8, sensor_x=None, sensor_y=None, border=2, border_y=None,
                 half_window=2, te=8.0, t0=0.0, dt=0.006, substeps=32, max_value=20.0, check_max_value="y",
                 agent_power=10.0, action_punish=0.0, delta_action_punish=0.0, window_size=3, temporal_steps=2,
                 nna_scale=2.0, nna_scale_critic=17.0, drop_middle_layer=True, gamma=0.99, rho=0.995,
                 batch_size=3, start_steps=-1, update_after=1, update_freq=1, update_loops=20,
                 learning_rate=0.0005, learning_rate_critic=0.001, act_limit=1.0, act_noise=1.2,
                 trajectory_length=100_000, memory_size=0):
        self.nx, self.ny = int(nx), int(ny)
        self.Lx = 0.1 * self.nx if Lx is None else float(Lx)          # dx = 0.1 as in the 1-D script (:39)
        self.dx = self.Lx / self.nx
        self.half_window = int(half_window)
        # 1-based positions like the reference (3:5:nx); the default 2-D grid uses a spacing that divides 256
        self.sensor_x = (np.arange(3, nx + 1, sensor_step) if sensor_x is None else np.asarray(sensor_x)).astype(np.int64)
        self.sensor_y = (np.arange(3, ny + 1, sensor_step) if sensor_y is None else np.asarray(sensor_y)).astype(np.int64)
        self.Sx, self.Sy = len(self.sensor_x), len(self.sensor_y)
        bx = int(border)
        by = min(bx, (self.Sy - 1) // 2) if border_y is None else int(border_y)
        ix, iy = np.arange(bx, self.Sx - bx), np.arange(by, self.Sy - by)
        self.actuators_to_sensors = (iy[:, None] * self.Sx + ix[None, :]).reshape(-1) + 1     # 1-based, like the reference
        self.te, self.t0, self.dt = te, t0, dt
        self.oversampling = int(substeps)
        self.max_value, self.check_max_value, self.agent_power = max_value, check_max_value, agent_power
        self.action_punish, self.delta_action_punish = action_punish, delta_action_punish
        self.window_size, self.temporal_steps, self.memory_size, self.mono, self.n_species = \
            window_size, int(temporal_steps), int(memory_size), False, 2
        from .ks import _refuse_unbuilt_branches
        _refuse_unbuilt_branches(self, "scripts/Keller-Segel/setup/KellerSegelSetup.jl:295-314")
        self.nna_scale, self.nna_scale_critic, self.drop_middle_layer = nna_scale, nna_scale_critic, drop_middle_layer
        self.gamma, self.rho, self.batch_size = gamma, rho, batch_size
        self.start_steps, self.update_after, self.update_freq, self.update_loops = \
            start_steps, update_after, update_freq, update_loops
        self.learning_rate, self.learning_rate_critic = learning_rate, learning_rate_critic
        self.act_limit, self.act_noise, self.trajectory_length = act_limit, act_noise, trajectory_length

    @property
    def n_actuators(self):
        return len(self.actuators_to_sensors)

    @property
    def n_sensors(self):
        return self.Sx * self.Sy

    @property
    def state_shape(self):
        return (2 * self.window_size ** 2 * self.temporal_steps, self.n_actuators)

    @property
    def action_shape(self):
        return (1, self.n_actuators)

    @property
    def reward_len(self):
        return self.n_actuators

    @property
    def y_shape(self):
        return (2, self.nx, self.ny)      # Julia y[2, nx, ny]  ->  memory [ny][nx][2]

    @property
    def p_shape(self):
        return (self.nx, self.ny)

    def y0_standard(self):
        """y0_2D_standard (KellerSegelSetup.jl:60-61) on the grid: u = 1, v = 1.01; host image [2, ny, nx]"""
        return np.stack([np.ones((self.ny, self.nx)), 1.01 * np.ones((self.ny, self.nx))])

    def generate_random_init(self, rng, B=1):
        """KellerSegelSetup.jl:373-384 with the sine sums taken along x and along y -> [B, 2, ny, nx]"""
        nsx, nsy = int(np.ceil(self.Lx / 3)), int(np.ceil(self.ny * self.dx / 3))
        xx, yy = self.dx * np.arange(1, self.nx + 1), self.dx * np.arange(1, self.ny + 1)
        a = rng.uniform(-1, 1, (B, 2, nsx + nsy))
        a /= np.linalg.norm(a.reshape(B, -1), axis=1)[:, None, None]
        y0 = np.ones((B, 2, self.ny, self.nx))
        for i in range(1, nsx + 1):
            y0 += a[:, :, i - 1, None, None] * np.sin(i * xx / (2 * np.pi * (self.Lx / 22)))[None, None, None, :]
        for i in range(1, nsy + 1):
            y0 += a[:, :, nsx + i - 1, None, None] * np.sin(i * yy / (2 * np.pi * (self.ny * self.dx / 22)))[None, None, :, None]
        return y0

    def env_cfg(self, B, dtype_code):
        c = _lib.EnvCfg()
        c.pde_kind, c.dtype, c.B, c.N, c.Ny, c.n_species = _lib.PDE_KSEG2D_RK4, dtype_code, B, self.nx, self.ny, 2
        c.S, c.A, c.window, c.temporal_steps, c.mono = self.n_sensors, self.n_actuators, self.window_size, self.temporal_steps, 0
        c.K = self.oversampling
        c.check_max_value = {"y": 1, "reward": 2}.get(self.check_max_value, 0)
        c.Lx, c.dt, c.mu, c.max_value = self.Lx, self.dt, 0.0, self.max_value
        side = 2 * self.half_window + 1
        c.sensor_scale = 0.25 / side                                # :276 (1/4) x box height: y-invariant data give the 1-D values
        c.agent_power = self.agent_power
        c.reward_in_scale, c.reward_offset = 1.0 / side, -1.0      # :248  <u - 1, box> / box height
        c.reward_power, c.reward_denom = 2.0, 800.0
        c.action_punish, c.delta_action_punish = self.action_punish, self.delta_action_punish
        return c

    def tables(self):
        return (np.ascontiguousarray(self.sensor_x - 1, dtype=np.int32),
                np.ascontiguousarray(self.sensor_y - 1, dtype=np.int32),
                np.ascontiguousarray(self.actuators_to_sensors - 1, dtype=np.int32))
