"""2-D fluid (vorticity) experiment configuration: the globals of
scripts/Fluid/setup/FluidSetup.jl:28-161 plus the per-experiment values
(scripts/Fluid/Fluid_8/Fluid_8.jl:2-16), turned into the tables libpdeconv needs.
Setup-time host code (initial conditions, sensor bumps); the per-step closures
(do_step / rk4 / rhs / featurize / prepare_action / reward_function) run on the GPU.

Arrays follow Julia: fields are [ny, nx] = [row y, column x]; numpy.fft.fft2 / ifft2 have
FFTW's conventions.  Device memory is the column-major image: [nx][ny]."""
import numpy as np

from .. import _lib


def taylorvtx(xx, yy, Lx, Ly, x0, y0, a0, U_max):
    """src/fluid_rk4.jl:54-69: the 9 periodic images of a Taylor vortex, returned as spectrum."""
    omg = np.zeros_like(xx)
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            r2 = (xx - x0 - i * Lx) ** 2 + (yy - y0 - j * Ly) ** 2
            omg = omg + U_max / a0 * (2 - r2 / a0 ** 2) * np.exp(0.5 * (1 - r2 / a0 ** 2))
    return np.fft.fft2(omg)


def _ring_window(mask):
    """first index and length of the shortest circular window covering every True of `mask`"""
    n = len(mask)
    if not mask.any():
        return 0, 1
    if mask.all():
        return 0, n
    idx = np.flatnonzero(mask)
    gaps = np.diff(np.concatenate([idx, [idx[0] + n]]))      # zero runs between consecutive non-zeros (+1)
    g = int(np.argmax(gaps))
    start = int(idx[(g + 1) % len(idx)])
    return start, int(n - (gaps[g] - 1))


def boxes_of(kernels):
    """dense [S][ny, nx] kernels -> (boxes [S][BW][BH], origin [S][2] = (j0, i0), BH, BW): the periodic
    bounding box of each kernel's support (what the reference keeps as `sparse(p)`, FluidSetup.jl:151)."""
    S, ny, nx = kernels.shape
    win = []
    for g in kernels:
        nz = g != 0.0
        i0, h = _ring_window(nz.any(axis=1))      # rows (y)
        j0, w = _ring_window(nz.any(axis=0))      # columns (x)
        win.append((i0, h, j0, w))
    BH = max(w[1] for w in win)
    BW = max(w[3] for w in win)
    boxes = np.zeros((S, BW, BH))
    origin = np.zeros((S, 2), dtype=np.int32)
    for s, (i0, h, j0, w) in enumerate(win):
        ii = (i0 + np.arange(BH)) % ny
        jj = (j0 + np.arange(BW)) % nx
        sub = kernels[s][np.ix_(ii, jj)].T.copy()           # [BW][BH]
        # a box wider than the support wraps onto real entries only if BH/BW exceed the ring; mask repeats
        if BH > h:
            sub[:, h:] = 0.0
        if BW > w:
            sub[w:, :] = 0.0
        boxes[s] = sub
        origin[s] = (j0, i0)
    return boxes, origin, BH, BW


class FluidSetup:
    # create_agent's default for `quirk_frozen_targets` (see KSSetup): the Fluid_8 / 16 / 32 learning curves are reproduced with
    # MOVING targets; with frozen ones 4 of 6 seeds diverge (HISTORY.md 5.1, tests/test_gpu_training.py).
    reproduces_reference_with = "moving"

    is_fluid = True

    def __init__(self, nx=128, Lx=1.0, nu=5e-5, te=6.0, t0=0.0, dt=0.02, oversampling=None, ifpad=1,
                 sensors_per_axis=8, variance=0.08, max_value=3.0, check_max_value="reward", agent_power=70.0,
                 action_punish=0.002, delta_action_punish=0.002, window_size=3, temporal_steps=1,
                 nna_scale=1.8, nna_scale_critic=17.0, drop_middle_layer=True, gamma=0.99, rho=0.995,
                 batch_size=3, start_steps=10, update_after=10, update_freq=1, update_loops=20,
                 learning_rate=0.0005, learning_rate_critic=0.001, act_limit=1.0, act_noise=1.2,
                 trajectory_length=1_800_000, evaluation=False, memory_size=0):
        self.nx = self.ny = int(nx)
        self.Lx = self.Ly = float(Lx)
        self.dx = self.dy = self.Lx / self.nx
        self.nu, self.te, self.t0, self.dt, self.ifpad = nu, te, t0, dt, int(ifpad)
        self.oversampling = int(np.floor(16 * nx * dt)) if oversampling is None else int(oversampling)   # :47
        self.sensors_per_axis, self.variance = int(sensors_per_axis), variance
        self.max_value, self.check_max_value, self.agent_power = max_value, check_max_value, agent_power
        self.action_punish, self.delta_action_punish = action_punish, delta_action_punish
        self.window_size, self.temporal_steps, self.memory_size, self.mono, self.n_species = \
            window_size, int(temporal_steps), int(memory_size), False, 1
        from .ks import _refuse_unbuilt_branches
        _refuse_unbuilt_branches(self, "scripts/Fluid/setup/FluidSetup.jl:229-241", memory_built=True, reward_check_with_memory=True)
        self.nna_scale, self.nna_scale_critic, self.drop_middle_layer = nna_scale, nna_scale_critic, drop_middle_layer
        self.gamma, self.rho, self.batch_size = gamma, rho, batch_size
        self.start_steps, self.update_after, self.update_freq, self.update_loops = \
            start_steps, update_after, update_freq, update_loops
        self.learning_rate, self.learning_rate_critic = learning_rate, learning_rate_critic
        self.act_limit, self.act_noise, self.trajectory_length = act_limit, act_noise, trajectory_length
        self.evaluation = evaluation
        n = self.nx
        x1 = np.linspace(0, self.Lx, n + 1)[:n]                                   # :127-131
        self.xx = np.ones((n, n)) * x1[None, :]                                   # meshgrid, fluid_rk4.jl:10-15
        self.yy = np.ones((n, n)) * x1[:, None]
        st = n // self.sensors_per_axis
        self.sensor_positions = [(i, j) for i in range(1, n + 1, st) for j in range(1, n + 1, st)]   # :61
        self.actuator_positions = list(self.sensor_positions)                     # :62
        self.actuators_to_sensors = np.arange(1, len(self.sensor_positions) + 1)  # :63
        self.gaussians = self.prepare_gaussians(1)                                # :159
        self.gaussians_actuators = self.prepare_gaussians(2)[self.actuators_to_sensors - 1]   # :160-161

    def taylorvtx(self, x0, y0, a0, U_max):
        return taylorvtx(self.xx, self.yy, self.Lx, self.Ly, x0, y0, a0, U_max)

    def prepare_gaussians(self, norm_mode):
        """FluidSetup.jl:139-157: thresholded Taylor-vortex bumps, sum- or max-normalised"""
        out = np.empty((len(self.sensor_positions), self.ny, self.nx))
        for s, (i, j) in enumerate(self.sensor_positions):
            p = np.real(np.fft.ifft2(self.taylorvtx(i * self.dx - self.dx, j * self.dy - self.dy, self.variance, 1.0)))
            p[p < 0.1] = 0.0
            out[s] = p / p.sum() if norm_mode == 1 else p / p.max()
        return out

    def ic(self, caseno, rng):
        """src/fluid_rk4.jl:72-120 (rng: numpy Generator; same distributions, not Julia's stream)"""
        Lx, Ly = self.Lx, self.Ly
        if caseno == 1:
            return self.taylorvtx(Lx / 2, Ly / 2, Lx / 8, 1.0)
        if caseno == 2:
            return self.taylorvtx(Lx / 2, 0.4 * Ly, Lx / 10, 1.0) + self.taylorvtx(Lx / 2, 0.6 * Ly, Lx / 10, 1.0)
        out = 0
        for _ in range(30 if caseno == 3 else 50):
            x0, y0 = rng.random() * Lx, rng.random() * Ly
            a0 = Lx / 20 if caseno == 3 else Lx / 20 * (0.5 + rng.random())
            out = out + self.taylorvtx(x0, y0, a0, rng.random() * 2 - 1.0)
        return out

    def ic_vortices(self, caseno, rng, B=1):
        """the vortex table [B, nv, 4] = (x0, y0, a0, U_max) of ic(caseno) with the same draws, in the same order,
        as `ic` above (src/fluid_rk4.jl:72-120); input of the device initialiser pdec_fluid_ic"""
        Lx, Ly = self.Lx, self.Ly
        out = []
        for _ in range(B):
            if caseno == 1:
                v = [(Lx / 2, Ly / 2, Lx / 8, 1.0)]
            elif caseno == 2:
                v = [(Lx / 2, 0.4 * Ly, Lx / 10, 1.0), (Lx / 2, 0.6 * Ly, Lx / 10, 1.0)]
            else:
                v = []
                for _ in range(30 if caseno == 3 else 50):
                    x0, y0 = rng.random() * Lx, rng.random() * Ly
                    a0 = Lx / 20 if caseno == 3 else Lx / 20 * (0.5 + rng.random())
                    v.append((x0, y0, a0, rng.random() * 2 - 1.0))
            out.append(v)
        return np.ascontiguousarray(out, dtype=np.float64)

    def random_init_device(self, env, rng):
        """generate_random_init (FluidSetup.jl:386-394) evaluated on the GPU: returns a device tensor shaped like env.y"""
        import ctypes as C
        import torch
        v = self.ic_vortices(4 if self.evaluation else 3, rng, env.B)
        out = torch.empty_like(env.y)
        _lib.check(env.lib.pdec_fluid_ic(env.handle, v.ctypes.data_as(C.POINTER(C.c_double)), v.shape[1], _lib.ptr(out)))
        return out

    # shapes of the RL.jl-facing arrays (per trajectory, Julia shapes)
    @property
    def n_actuators(self):
        return len(self.actuator_positions)

    @property
    def n_sensors(self):
        return len(self.sensor_positions)

    @property
    def state_shape(self):
        return (self.window_size ** 2 * self.temporal_steps + self.memory_size, self.n_actuators)

    @property
    def action_shape(self):
        return (1 + self.memory_size, self.n_actuators)

    @property
    def reward_len(self):
        return self.n_actuators

    @property
    def y_shape(self):
        return (self.ny, self.nx)

    def y0_standard(self, rng=None):
        """y0_2D_standard = ic(4, rng), FluidSetup.jl:134"""
        return self.ic(4, rng or np.random.default_rng(0))

    def generate_random_init(self, rng, B=1):
        """FluidSetup.jl:386-394: ic(4) in evaluation, ic(3) in training; batched -> complex [B, ny, nx]"""
        return np.stack([self.ic(4 if self.evaluation else 3, rng) for _ in range(B)])

    def error_detection(self, y):
        """scripts/Fluid/setup/FluidSetup.jl:263-273, handed to PDEhook (FluidSetup.jl:373-377; src/PDEhook.jl:78-82): an episode
        that ended early counts as errored when neighbouring cells of real(ifft(y)) differ by more than 10 along either axis.
        `y`: the spectral field of the env, [ny, nx] or a batch [B, ny, nx] (device tensor or array); batched: any trajectory."""
        import torch
        yt = torch.as_tensor(y)
        w = torch.fft.ifft2(yt, dim=(-2, -1)).real
        y_x = (torch.roll(w, 1, dims=-2) - w).abs()
        y_y = (torch.roll(w, 1, dims=-1) - w).abs()
        return bool(max(float(y_x.max()), float(y_y.max())) > 10.0)

    def make_hook(self, **kw):
        """the training hook of the fluid script (FluidSetup.jl:373-377): PDEhook with this setup's error_detection"""
        from ..hook import PDEhook
        kw.setdefault("error_detection", self.error_detection)
        return PDEhook(**kw)

    def env_cfg(self, B, dtype_code):
        c = _lib.EnvCfg()
        c.pde_kind, c.dtype, c.B, c.N, c.n_species = _lib.PDE_FLUID_RK4, dtype_code, B, self.nx, 1
        c.S, c.A, c.window, c.temporal_steps, c.mono = self.n_sensors, self.n_actuators, self.window_size, self.temporal_steps, 0
        c.K = self.oversampling
        c.check_max_value = {"y": 1, "reward": 2}.get(self.check_max_value, 0)
        c.Lx, c.dt, c.mu, c.max_value = self.Lx, self.dt, 0.0, self.max_value
        c.sensor_scale = 1.0 / 70.0                                 # FluidSetup.jl:216
        c.agent_power = self.agent_power                            # :257
        c.reward_in_scale, c.reward_offset = 1.0, 0.0
        c.reward_power, c.reward_denom = 1.1, 320.0                 # :197
        c.action_punish, c.delta_action_punish = self.action_punish, self.delta_action_punish
        c.ifpad, c.sensors_per_axis, c.nu = self.ifpad, self.sensors_per_axis, self.nu
        c.memory_size = self.memory_size
        return c

    def box_tables(self):
        sb, so, bh1, bw1 = boxes_of(self.gaussians)
        ab, ao, bh2, bw2 = boxes_of(self.gaussians_actuators)
        BH, BW = max(bh1, bh2), max(bw1, bw2)

        def grow(b, bh, bw):
            if (bh, bw) == (BH, BW):
                return b
            out = np.zeros((b.shape[0], BW, BH))
            out[:, :bw, :bh] = b
            return out
        return (np.ascontiguousarray(grow(sb, bh1, bw1)), so, np.ascontiguousarray(grow(ab, bh2, bw2)), ao, BH, BW,
                np.ascontiguousarray(self.actuators_to_sensors - 1, dtype=np.int32))

    @classmethod
    def Fluid_8(cls, **kw):
        return cls(sensors_per_axis=8, variance=0.08, **kw)

    @classmethod
    def Fluid_16(cls, **kw):
        return cls(sensors_per_axis=16, variance=0.04, **kw)

    @classmethod
    def Fluid_32(cls, **kw):
        return cls(sensors_per_axis=32, variance=0.022, **kw)
