"""Kuramoto-Sivashinsky environment pieces, fp64 NumPy restatement.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED by tests/golden/ks*_hook.npz."""
import numpy as np

from .julia_compat import julia_float_range, circshift


class KSConfig:
    """Globals of scripts/KS/setup/KSSetup.jl:20-77 + the experiment script
    (e.g. scripts/KS/KS22/KS22.jl:2-21)."""

    def __init__(self, nx, Lx, sensor_positions, actuator_positions=None, actuators_to_sensors=None,
                 sigma_sensors=1.0, sigma_actuators=1.0, mu=0.0, dt=0.1, oversampling=30,
                 max_value=30.0, agent_power=7.5, action_punish=0.002, delta_action_punish=0.002,
                 window_size=1, te=5.0, mono=False, disturbance_in_step=True, temporal_steps=1):
        self.nx, self.Lx = int(nx), float(Lx)
        self.dx = self.Lx / self.nx                                   # KSSetup.jl:34
        self.sensor_positions = np.asarray(sensor_positions, dtype=np.int64)      # 1-based cells
        self.actuator_positions = (self.sensor_positions if actuator_positions is None
                                   else np.asarray(actuator_positions, dtype=np.int64))
        n_act = len(self.actuator_positions)
        self.actuators_to_sensors = (np.arange(1, n_act + 1) if actuators_to_sensors is None
                                     else np.asarray(actuators_to_sensors, dtype=np.int64))  # 1-based
        self.sigma_sensors, self.sigma_actuators = sigma_sensors, sigma_actuators
        self.mu, self.dt, self.oversampling = mu, dt, int(oversampling)
        self.max_value, self.agent_power = max_value, agent_power
        self.action_punish, self.delta_action_punish = action_punish, delta_action_punish
        self.window_size, self.te, self.mono = window_size, te, mono
        self.temporal_steps = int(temporal_steps)                     # KSSetup.jl:30 (1 in every shipped script)
        self.disturbance_in_step = disturbance_in_step                # absent in KSglobalSetup.jl:167
        self.xx = self.dx * np.arange(1, self.nx + 1)                 # collect(dx:dx:Lx), KSSetup.jl:36
        self.gaussians = prepare_gaussians(self, sigma_sensors, 1, self.sensor_positions)
        if mono:   # KSglobalSetup.jl:99-102,125: actuator kernels at actuator_positions, no re-indexing
            self.gaussians_actuators = prepare_gaussians(self, sigma_actuators, 2, self.actuator_positions)
        else:      # KSSetup.jl:112-113
            ga = prepare_gaussians(self, sigma_actuators, 2, self.sensor_positions)
            self.gaussians_actuators = ga[self.actuators_to_sensors - 1]


def prepare_gaussians(cfg, sigma, norm_mode, positions):
    """scripts/KS/setup/KSSetup.jl:82-109.  Note the reference's operator precedence:
    exp(-((t-pos*dx)^2 / 2 * sigma^2)) (MULTIPLIED by sigma^2) and 1/sqrt(2*pi*sigma)."""
    dx, nx, Lx = cfg.dx, cfg.nx, cfg.Lx
    extra = 50
    t = julia_float_range(dx - extra * dx, dx, Lx + extra * dx)              # :87
    out = []
    for position in positions:
        p = (1.0 / np.sqrt(2 * np.pi * sigma)) * np.exp(-(((t - position * dx) * 1) ** 2 / 2 * sigma ** 2))  # :90
        if norm_mode == 1:
            p = p / p.sum()                                                   # :93
        else:
            p = p / p.max()                                                   # :95
        pleft = p[:extra]                                                     # :98
        pright = p[extra + nx:]                                               # :99
        q = p[extra:extra + nx].copy()                                        # :100
        q[nx - len(pleft):] += pleft                                          # :101
        q[:len(pright)] += pright                                             # :102
        out.append(q)
    return np.array(out)


def ks_operators(cfg, K=None):
    """scripts/KS/setup/KSSetup.jl:115-123,131-135."""
    nx, Lx = cfg.nx, cfg.Lx
    K = cfg.oversampling if K is None else K
    kx = np.concatenate([np.arange(0, nx // 2), [0], np.arange(-nx // 2 + 1, 0)]).astype(np.float64)  # :115
    alpha = 2 * np.pi * kx / Lx                                               # :116
    D = 1j * alpha                                                            # :117
    L = alpha ** 2 - alpha ** 4                                               # :118
    G = -0.5 * D                                                              # :119
    h = cfg.dt / K                                                            # :131
    dt2, dt32 = h / 2, 3 * h / 2                                              # :132-133
    A_inv = (np.ones(nx) - dt2 * L) ** (-1)                                   # :134
    B = np.ones(nx) + dt2 * L                                                 # :135
    return dict(kx=kx, alpha=alpha, L=L, G=G, h=h, dt2=dt2, dt32=dt32, A_inv=A_inv, B=B)


def do_step(cfg, y, p, ctype=np.complex128):
    """KS CNAB2 control step, scripts/KS/setup/KSSetup.jl:130-160 (twin
    KSglobalSetup.jl:142-172 without the disturbance term).  `ctype=np.complex64` runs the
    same arithmetic in single precision (used to set the fp32 tolerance)."""
    op = ks_operators(cfg)
    rtype = np.float32 if ctype == np.complex64 else np.float64
    G = op["G"].astype(ctype)
    A_inv, B = op["A_inv"].astype(rtype), op["B"].astype(rtype)
    dt2, dt32, h = rtype(op["dt2"]), rtype(op["dt32"]), rtype(op["h"])
    fft = lambda a: np.fft.fft(a).astype(ctype)
    ifft = lambda a: np.fft.ifft(a).astype(ctype)
    u = np.asarray(y).astype(ctype)                                           # :137-138
    Nn = G * fft(u ** 2)                                                      # :140
    Nn1 = Nn.copy()                                                           # :141
    u = fft(u)                                                                # :142
    P = fft(np.asarray(p).astype(ctype))
    if cfg.disturbance_in_step:
        Dist = h * fft((cfg.mu * np.cos((2 + np.pi + cfg.xx / (cfg.Lx / 2)))).astype(ctype))
    else:
        Dist = 0
    for _ in range(cfg.oversampling):                                         # :144
        Nn1 = Nn                                                              # :145
        w = ifft(u)                                                           # :146-148
        Nn = G * fft(w * w)                                                   # :149-152
        u = A_inv * (B * u + dt32 * Nn - dt2 * Nn1 + h * P) + Dist            # :155
    return np.real(ifft(u)).astype(rtype)                                     # :158-159


def rhs_fd(cfg, u, p):
    """The north-star's RK4 + periodic finite-difference KS variant, u_t = -u u_x - u_xx - u_xxxx + p + disturbance, on
    the 5-point stencil rows the reference defines but never uses (scripts/KS/setup/KSSetup.jl:55-59).  NOT the
    reference's integrator (that is the CNAB2 spectral do_step above): own known-answer tests only."""
    u = np.asarray(u, dtype=np.float64)
    dx = cfg.dx
    m2, m1, p1, p2 = np.roll(u, 2), np.roll(u, 1), np.roll(u, -1), np.roll(u, -2)
    ux = (p1 - m1) * (0.5 / dx)
    uxx = (m1 - 2 * u + p1) * (1.0 / (dx * dx))
    uxxxx = (m2 - 4 * m1 + 6 * u - 4 * p1 + p2) * (1.0 / (dx * dx)) ** 2
    dist = cfg.mu * np.cos(2 + np.pi + cfg.xx / (cfg.Lx / 2)) if cfg.disturbance_in_step else 0.0
    return -u * ux - uxx - uxxxx + np.asarray(p, dtype=np.float64) + dist


def do_step_rk4_fd(cfg, y, p, K=None):
    """K classical RK4 sub-steps (src/fluid_rk4.jl:122-132 form) of rhs_fd over one control interval dt"""
    K = cfg.oversampling if K is None else K
    h = cfg.dt / K
    u = np.asarray(y, dtype=np.float64)
    for _ in range(K):
        k1 = rhs_fd(cfg, u, p)
        k2 = rhs_fd(cfg, u + 0.5 * h * k1, p)
        k3 = rhs_fd(cfg, u + 0.5 * h * k2, p)
        k4 = rhs_fd(cfg, u + h * k3, p)
        u = u + h / 6 * (k1 + 2 * (k2 + k3) + k4)
    return u


def do_step_midpoint_fd(cfg, y, p, K=None):
    """PDEenv's built-in integrator (src/PDEenv.jl:208-214: explicit midpoint, K = oversampling sub-steps) on rhs_fd"""
    K = cfg.oversampling if K is None else K
    h = cfg.dt / K
    u = np.asarray(y, dtype=np.float64)
    for _ in range(K):
        u_old = u
        u = u + 0.5 * h * rhs_fd(cfg, u, p)
        u = u_old + h * rhs_fd(cfg, u, p)
    return u


def sensor_dots(cfg, y):
    return cfg.gaussians @ np.asarray(y, dtype=np.float64)


def reward_function(cfg, y, action, delta_action):
    """scripts/KS/setup/KSSetup.jl:162-178; mono variant KSglobalSetup.jl:175-199."""
    y6 = np.asarray(y, dtype=np.float64) * 6                                  # :163
    a2s = cfg.actuators_to_sensors - 1
    sensors = np.abs(cfg.gaussians[a2s] @ y6) ** 1.3 / (cfg.max_value * 3)    # :169
    sensor_rewards = -np.abs(sensors)                                         # :171
    a = np.asarray(action, dtype=np.float64).reshape(-1)[:len(a2s)] if cfg.mono else np.asarray(action)[0, :]
    da = np.asarray(delta_action, dtype=np.float64).reshape(-1)[:len(a2s)] if cfg.mono else np.asarray(delta_action)[0, :]
    r = sensor_rewards - cfg.action_punish * a ** 2 - cfg.delta_action_punish * da ** 2   # :178
    if cfg.mono:
        return np.array([r.mean()])                                           # KSglobalSetup.jl:199
    return r


def featurize(cfg, y, prev_state=None, action=None):
    """scripts/KS/setup/KSSetup.jl:190-229; mono variant KSglobalSetup.jl:211-249 returns the
    [S,1] column.  temporal_steps > 1 (:209-218; 1 in every shipped KS script): prev_state=None is the reference's
    `isnothing(env)` branch (the fresh rows repeated), otherwise the fresh rows stacked on the newest rows of the
    previous state (`env.state[1:end-size(result)[1]-memory_size, :]`).  memory_size > 0 (cfg.memory_size; 0 in every shipped
    script; :220-226): the last memory_size rows are rows 2.. of env.action [1 + memory_size, A] -- `action` here -- or zeros
    in the `isnothing(env)` form (action=None)."""
    sensors = sensor_dots(cfg, y) / cfg.max_value                             # :201
    if cfg.mono:
        return sensors.reshape(-1, 1)                                         # KSglobalSetup.jl:225-227
    w = int(np.floor(cfg.window_size / 2))                                    # :204
    rows = [circshift(sensors, i) for i in range(-w, w + 1)]                  # :205
    result = np.stack(rows)
    result = result[:, cfg.actuators_to_sensors - 1]                          # :207
    T = getattr(cfg, "temporal_steps", 1)
    m = int(getattr(cfg, "memory_size", 0))
    if T > 1:                                                                 # :209
        if prev_state is None:
            result = np.concatenate([result] * T)                             # :211-214
        else:
            prev = np.asarray(prev_state, dtype=np.float64)
            result = np.concatenate([result, prev[:prev.shape[0] - result.shape[0] - m, :]])   # :216
    if m > 0:                                                                 # :220
        if action is None:
            result = np.concatenate([result, np.zeros((m, result.shape[1]))])  # :222
        else:
            result = np.concatenate([result, np.asarray(action, dtype=np.float64)[-m:, :]])   # :224
    return result


def prepare_action(cfg, action):
    """scripts/KS/setup/KSSetup.jl:231-245."""
    a = np.asarray(action, dtype=np.float64)
    a = a.reshape(-1) if cfg.mono else a[0, :]
    p = np.zeros(cfg.nx)
    for i in range(len(cfg.actuator_positions)):
        p = p + cfg.agent_power * a[i] * cfg.gaussians_actuators[i]           # :241
    return p


def generate_random_init(cfg, rng):
    """scripts/KS/setup/KSSetup.jl:288-298 (rng: numpy Generator; the reference draws from
    Julia's global RNG, so only the distribution is reproduced)."""
    number_sin = 8
    a_i = rng.uniform(-1, 1, number_sin)
    a_i = a_i / np.linalg.norm(a_i)
    y0 = np.zeros(cfg.nx)
    for i in range(1, number_sin + 1):
        y0 += a_i[i - 1] * np.sin(i * cfg.xx / (2 * np.pi))
    return y0 * 30 / np.linalg.norm(y0)


def env_step(cfg, y, action_prev, action, time, prev_state=None):
    """(env::PDEenv)(action), src/PDEenv.jl:195-241, with the KS closures (prev_state: env.state before the step, read
    by featurize when temporal_steps > 1)."""
    action = np.asarray(action, dtype=np.float64)
    delta_action = action - np.asarray(action_prev, dtype=np.float64)         # :196
    p = prepare_action(cfg, action)                                           # :199
    y_new = do_step(cfg, y, p)                                                # :217
    reward = reward_function(cfg, y_new, action, delta_action)                # :220
    state = featurize(cfg, y_new, prev_state, action if getattr(cfg, "memory_size", 0) else None)   # :222
    time = time + cfg.dt                                                      # :225
    done = bool(time >= cfg.te or np.max(np.abs(y_new)) > cfg.max_value)      # :227
    return dict(y=y_new, p=p, reward=reward, state=state, done=done, time=time,
                delta_action=delta_action)
