"""2-D pseudo-spectral vorticity solver pieces, fp64 NumPy restatement.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the reference saved no
fluid trajectory (collect_bestDF=false, scripts/Fluid/setup/FluidSetup.jl:373-377); pinned
only by analytic known-answer tests in tests/test_oracle_fluid.py.
Arrays are [ny, nx] = Julia [row=y, col=x]; numpy.fft.fft2/ifft2 match FFTW's conventions
(unnormalised forward, 1/(nx*ny) inverse)."""
import numpy as np


class FluidConfig:
    """scripts/Fluid/setup/FluidSetup.jl:28-47,98-118."""

    def __init__(self, nx=128, Lx=1.0, Ly=1.0, nu=5e-5, dt=0.02, ifpad=1, sensors_per_axis=8,
                 variance=0.08, agent_power=70.0, action_punish=0.002, delta_action_punish=0.002,
                 window_size=3, te=6.0, max_value=3.0, oversampling=None):
        self.nx = self.ny = int(nx)
        self.Lx, self.Ly, self.nu, self.dt, self.ifpad = Lx, Ly, nu, dt, ifpad
        self.dx, self.dy = Lx / nx, Ly / nx
        self.oversampling = int(np.floor(16 * nx * dt)) if oversampling is None else oversampling  # :47
        self.nxp = self.nyp = nx * 3 // 2                                     # :103
        n = self.nx
        self.kx = np.concatenate([np.arange(0, n // 2 + 1), np.arange(-n // 2 + 1, 0)]) / Lx * 2 * np.pi  # :106
        self.ky = np.concatenate([np.arange(0, n // 2 + 1), np.arange(-n // 2 + 1, 0)]) / Ly * 2 * np.pi  # :107
        self.kx2ky2 = self.ky[:, None] ** 2 + self.kx[None, :] ** 2           # :116 [i,j]=ky2[i]+kx2[j]
        self.kx_repeat = np.tile(self.kx[None, :], (n, 1))                    # :117
        self.ky_repeat = np.tile(self.ky[:, None], (1, n))                    # :118
        x1 = np.linspace(0, Lx, n + 1)[:n]                                    # :127-131
        y1 = np.linspace(0, Ly, n + 1)[:n]
        self.xx = np.ones((n, n)) * x1[None, :]                               # meshgrid, fluid_rk4.jl:10-15
        self.yy = np.ones((n, n)) * y1[:, None]
        self.sensors_per_axis, self.variance = sensors_per_axis, variance
        self.agent_power, self.window_size = agent_power, window_size
        self.action_punish, self.delta_action_punish = action_punish, delta_action_punish
        self.te, self.max_value = te, max_value
        st = n // sensors_per_axis
        self.sensor_positions = [(i, j) for i in range(1, n + 1, st) for j in range(1, n + 1, st)]  # :61
        self._g = None

    @property
    def gaussians(self):
        if self._g is None:
            self._g = (prepare_gaussians(self, 1), prepare_gaussians(self, 2))
        return self._g[0]

    @property
    def gaussians_actuators(self):
        self.gaussians
        return self._g[1]


def taylorvtx(cfg, x0, y0, a0, U_max):
    """src/fluid_rk4.jl:54-69: sum of the 9 periodic images of a Taylor vortex, then fft."""
    omg = np.zeros_like(cfg.xx)
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            r2 = (cfg.xx - x0 - i * cfg.Lx) ** 2 + (cfg.yy - y0 - j * cfg.Ly) ** 2
            omg = omg + U_max / a0 * (2 - r2 / a0 ** 2) * np.exp(0.5 * (1 - r2 / a0 ** 2))
    return np.fft.fft2(omg)


def ic(cfg, caseno, rng):
    """src/fluid_rk4.jl:72-120 (cases 1-4; rng = numpy Generator -> distribution only)."""
    Lx, Ly = cfg.Lx, cfg.Ly
    if caseno == 1:
        return taylorvtx(cfg, Lx / 2, Ly / 2, Lx / 8, 1.0)
    if caseno == 2:
        return taylorvtx(cfg, Lx / 2, 0.4 * Ly, Lx / 10, 1.0) + taylorvtx(cfg, Lx / 2, 0.6 * Ly, Lx / 10, 1.0)
    nv = 30 if caseno == 3 else 50
    out = 0
    for _ in range(nv):
        x0, y0 = rng.random() * Lx, rng.random() * Ly
        a0 = Lx / 20 if caseno == 3 else Lx / 20 * (0.5 + rng.random())
        out = out + taylorvtx(cfg, x0, y0, a0, rng.random() * 2 - 1.0)
    return out


def pad(cfg, f):
    """src/fluid_rk4.jl:192-210."""
    ny, nx, nyp, nxp = cfg.ny, cfg.nx, cfg.nyp, cfg.nxp
    fp = np.zeros((nyp, nxp), dtype=complex)
    yh, xh = ny // 2, nx // 2
    fp[:yh + 1, :xh + 1] = f[:yh + 1, :xh + 1]
    fp[:yh + 1, nxp - xh + 1:] = f[:yh + 1, xh + 1:]
    fp[nyp - yh + 1:, :xh + 1] = f[yh + 1:, :xh + 1]
    fp[nyp - yh + 1:, nxp - xh + 1:] = f[yh + 1:, xh + 1:]
    return fp


def chop(cfg, fp):
    """src/fluid_rk4.jl:212-229."""
    ny, nx, nyp, nxp = cfg.ny, cfg.nx, cfg.nyp, cfg.nxp
    f = np.zeros((ny, nx), dtype=complex)
    yh, xh = ny // 2, nx // 2
    f[:yh + 1, :xh + 1] = fp[:yh + 1, :xh + 1]
    f[:yh + 1, xh + 1:] = fp[:yh + 1, nxp - xh + 1:]
    f[yh + 1:, :xh + 1] = fp[nyp - yh + 1:, :xh + 1]
    f[yh + 1:, xh + 1:] = fp[nyp - yh + 1:, nxp - xh + 1:]
    return f


def advection(cfg, omghat):
    """src/fluid_rk4.jl:145-190."""
    with np.errstate(divide="ignore", invalid="ignore"):
        psihat = omghat / cfg.kx2ky2                                          # :152
    psihat[0, 0] = 0.0                                                        # :153
    domgdx = 1j * omghat * cfg.kx_repeat                                      # :156
    domgdy = 1j * omghat * cfg.ky_repeat                                      # :157
    vhat = -1j * psihat * cfg.kx_repeat                                       # :160
    uhat = 1j * psihat * cfg.ky_repeat                                        # :161
    if cfg.ifpad == 1:
        up = np.real(np.fft.ifft2(pad(cfg, uhat)))                            # :169
        vp = np.real(np.fft.ifft2(pad(cfg, vhat)))                            # :170
        dxp = np.real(np.fft.ifft2(pad(cfg, domgdx)))                         # :171
        dyp = np.real(np.fft.ifft2(pad(cfg, domgdy)))                         # :172
        return chop(cfg, np.fft.fft2(-up * dxp - vp * dyp)) * 1.5 * 1.5        # :175-176
    u = np.real(np.fft.ifft2(uhat))                                           # :183
    v = np.real(np.fft.ifft2(vhat))                                           # :184
    return np.fft.fft2(-u * np.real(np.fft.ifft2(domgdx)) - v * np.real(np.fft.ifft2(domgdy)))  # :187


def rhs(cfg, omghat, p):
    """src/fluid_rk4.jl:134-143."""
    return -cfg.nu * (cfg.kx2ky2 * omghat) + advection(cfg, omghat) + p


def rk4(cfg, f, p, dt):
    """src/fluid_rk4.jl:122-132."""
    k1 = rhs(cfg, f, p)
    k2 = rhs(cfg, f + 0.5 * dt * k1, p)
    k3 = rhs(cfg, f + 0.5 * dt * k2, p)
    k4 = rhs(cfg, f + dt * k3, p)
    return f + dt / 6 * (k1 + 2 * (k2 + k3) + k4)


def do_step(cfg, y, p, oversampling=None):
    """scripts/Fluid/setup/FluidSetup.jl:163-172 (fixed-step variant; the adaptive do_step2 at
    :181-186 with tol=1e0 is what initialize_setup wires in and is not reproducible)."""
    K = cfg.oversampling if oversampling is None else oversampling
    h = cfg.dt / K
    for _ in range(K):
        y = rk4(cfg, y, p, h)
    return y


def prepare_gaussians(cfg, norm_mode):
    """scripts/Fluid/setup/FluidSetup.jl:139-157: thresholded Taylor-vortex bumps."""
    out = []
    for (i, j) in cfg.sensor_positions:
        p = np.real(np.fft.ifft2(taylorvtx(cfg, i * cfg.dx - cfg.dx, j * cfg.dy - cfg.dy, cfg.variance, 1.0)))
        p[p < 0.1] = 0.0                                                      # :145
        p = p / p.sum() if norm_mode == 1 else p / p.max()                    # :146-150
        out.append(p)
    return np.array(out)


def reward_function(cfg, yhat, action, delta_action):
    """scripts/Fluid/setup/FluidSetup.jl:188-202."""
    y = np.real(np.fft.ifft2(yhat))
    dots = np.tensordot(cfg.gaussians, y, axes=([1, 2], [0, 1]))
    sensors = np.abs(dots) ** 1.1 / 320                                       # :197
    a, da = np.asarray(action)[0, :], np.asarray(delta_action)[0, :]
    return -np.abs(sensors) - cfg.action_punish * a ** 2 - cfg.delta_action_punish * da ** 2


def error_detection(yhat):
    """scripts/Fluid/setup/FluidSetup.jl:263-273: the episode counts as errored when neighbouring cells of the vorticity field
    (real(ifft(y))) differ by more than 10 along either axis (periodic neighbours, circshift)"""
    y = np.real(np.fft.ifft2(np.asarray(yhat)))
    y_x = np.abs(np.roll(y, 1, axis=0) - y)
    y_y = np.abs(np.roll(y, 1, axis=1) - y)
    return bool(y_x.max() > 10.0 or y_y.max() > 10.0)


def featurize(cfg, yhat, prev_state=None, action=None):
    """scripts/Fluid/setup/FluidSetup.jl:204-245.  temporal_steps > 1 (:229-237; 1 in the shipped scripts):
    prev_state=None is the `isnothing(env)` branch (fresh rows repeated), otherwise the fresh rows are stacked on the newest
    rows of the previous state (without its memory rows).  memory_size > 0 (cfg.memory_size, :238-244; 0 in the shipped script):
    the last rows are rows 2.. of env.action (`action`), zeros in the `isnothing(env)` form."""
    y = np.real(np.fft.ifft2(yhat))
    spa = cfg.sensors_per_axis
    dots = np.tensordot(cfg.gaussians, y, axes=([1, 2], [0, 1])) / 70         # :216
    sensors = dots.reshape(spa, spa)           # sensors[floor((i-1)/spa), (i-1)%spa]
    w = int(np.floor(cfg.window_size / 2))
    rows = []
    for i in range(-w, w + 1):
        for j in range(-w, w + 1):
            sh = np.roll(np.roll(sensors, i, axis=0), j, axis=1)              # circshift(sensors,[i,j])
            # reshape(sh', (1, S)) in column-major Julia == row-major flatten of sh
            rows.append(sh.reshape(-1))                                       # :220-222
    result = np.stack(rows)
    T = getattr(cfg, "temporal_steps", 1)
    if T > 1:                                                                 # :229
        if prev_state is None:
            result = np.concatenate([result] * T)                             # :231-234
        else:
            prev = np.asarray(prev_state, dtype=np.float64)
            result = np.concatenate([result, prev[:prev.shape[0] - result.shape[0] - int(getattr(cfg, "memory_size", 0)), :]])   # :236
    m = int(getattr(cfg, "memory_size", 0))
    if m > 0:                                                                 # :240
        if action is None:
            result = np.concatenate([result, np.zeros((m, result.shape[1]))])  # :242
        else:
            result = np.concatenate([result, np.asarray(action, dtype=np.float64)[-m:, :]])   # :244
    return result


def prepare_action(cfg, action):
    """scripts/Fluid/setup/FluidSetup.jl:247-261."""
    a = np.asarray(action, dtype=np.float64)[0, :]
    p = np.tensordot(cfg.agent_power * a, cfg.gaussians_actuators, axes=(0, 0))
    return np.fft.fft2(p)
