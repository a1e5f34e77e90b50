"""Keller-Segel on a 2-D ny x nx grid (BASELINE.json configs[3]), fp64 NumPy.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's Keller-Segel is 1-D (scripts/Keller-Segel/setup/KellerSegelSetup.jl); the 2-D case of
BASELINE.json has NO reference counterpart.  This file states the extension the build uses and
keeps every 1-D rule of the reference along each axis:
  * RHS (KellerSegelSetup.jl:213-232): d/dx -> central gradient, d2/dx2 -> 5-point Laplacian
    (the :63-64 weights applied along x and along y), zero-flux edges by the same ghost = edge-cell fix-up
    (:220-223) on all four sides;  v' = Lap v - v + u + p,
    u' = Lap u + u - 5.6 grad u . grad v - 5.6 u Lap v - u^2;
  * sensors = boxes of ones with half-window 2 (:112-126) on a tensor grid of positions, divided by
    4 * (2*hw+1) (the 1-D scale 4 of :276 times the box height), so that a field that does not depend on y
    gives exactly the 1-D sensor values; 3x3 circular window over the sensor grid (the 2-D shift order of
    scripts/Fluid/setup/FluidSetup.jl:219-223), species u rows then v rows (:281-286), temporal stacking (:297-303);
  * reward (:241-257) with the box mean over y in place of the 1-D box sum; prepare_action (:318-332).
PARITY: pinned through the y-invariant reduction to the 1-D path, which tests/golden/kseg_hook.npz pins to the
reference (tests/test_oracle.py); everything that is genuinely 2-D is "parity unpinned" and covered by KATs."""
import numpy as np


class KSeg2DConfig:
    def __init__(self, nx=100, ny=5, Lx=10.0, sensor_x=None, sensor_y=None, border_x=2, border_y=None,
                 half_window=2, dt=0.006, te=8.0, agent_power=10.0, window_size=3, temporal_steps=2,
                 action_punish=0.0, delta_action_punish=0.0, max_value=20.0, substeps=32):
        self.nx, self.ny, self.Lx = int(nx), int(ny), float(Lx)
        self.dx = self.Lx / self.nx                       # square cells: dy = dx
        self.hw = int(half_window)
        step = 2 * self.hw + 1
        self.sensor_x = np.arange(3, nx + 1, step) if sensor_x is None else np.asarray(sensor_x, dtype=np.int64)
        self.sensor_y = np.arange(3, ny + 1, step) if sensor_y is None else np.asarray(sensor_y, dtype=np.int64)
        self.Sx, self.Sy = len(self.sensor_x), len(self.sensor_y)
        self.border_x = int(border_x)
        self.border_y = min(self.border_x, (self.Sy - 1) // 2) if border_y is None else int(border_y)
        ix = np.arange(self.border_x, self.Sx - self.border_x)
        iy = np.arange(self.border_y, self.Sy - self.border_y)
        # sensor s = iy * Sx + ix (row-major over the sensor grid); 0-based actuator -> sensor map
        self.a2s = (iy[:, None] * self.Sx + ix[None, :]).reshape(-1)
        self.dt, self.te, self.agent_power = dt, te, agent_power
        self.window_size, self.temporal_steps = window_size, temporal_steps
        self.action_punish, self.delta_action_punish = action_punish, delta_action_punish
        self.max_value, self.substeps = max_value, substeps

    @property
    def S(self):
        return self.Sx * self.Sy

    @property
    def A(self):
        return len(self.a2s)

    def box(self, s):
        """(row slice, column slice) of sensor s: cells position-hw .. position+hw, 1-based positions"""
        iy, ix = divmod(int(s), self.Sx)
        cy, cx = int(self.sensor_y[iy]), int(self.sensor_x[ix])
        return (slice(max(cy - self.hw - 1, 0), min(cy + self.hw, self.ny)),
                slice(max(cx - self.hw - 1, 0), min(cx + self.hw, self.nx)))


def _nb(a):
    """west, east, south, north neighbours with ghost = edge cell (zero flux)"""
    w = np.concatenate([a[:, :1], a[:, :-1]], axis=1)
    e = np.concatenate([a[:, 1:], a[:, -1:]], axis=1)
    s = np.concatenate([a[:1, :], a[:-1, :]], axis=0)
    n = np.concatenate([a[1:, :], a[-1:, :]], axis=0)
    return w, e, s, n


def f(cfg, y, p):
    """y [2, ny, nx] (0 = u, 1 = v), p [ny, nx]"""
    dx = cfg.dx
    u, v = y[0], y[1]
    uw, ue, us, un = _nb(u)
    vw, ve, vs, vn = _nb(v)
    ux = -0.5 / dx * uw + 0.5 / dx * ue
    uy = -0.5 / dx * us + 0.5 / dx * un
    vx = -0.5 / dx * vw + 0.5 / dx * ve
    vy = -0.5 / dx * vs + 0.5 / dx * vn
    lu = (uw / dx ** 2 - 2.0 / dx ** 2 * u + ue / dx ** 2) + (us / dx ** 2 - 2.0 / dx ** 2 * u + un / dx ** 2)
    lv = (vw / dx ** 2 - 2.0 / dx ** 2 * v + ve / dx ** 2) + (vs / dx ** 2 - 2.0 / dx ** 2 * v + vn / dx ** 2)
    vdot = lv - v + u + p
    udot = lu + u - 5.6 * ux * vx - 5.6 * uy * vy - 5.6 * u * lv - u ** 2
    return np.stack([udot, vdot])


def rk4_step(cfg, y, p, h):
    k1 = f(cfg, y, p)
    k2 = f(cfg, y + 0.5 * h * k1, p)
    k3 = f(cfg, y + 0.5 * h * k2, p)
    k4 = f(cfg, y + h * k3, p)
    return y + h / 6 * (k1 + 2 * (k2 + k3) + k4)


def do_step(cfg, y, p, substeps=None):
    n = cfg.substeps if substeps is None else substeps
    h = cfg.dt / n
    y = np.asarray(y, dtype=np.float64)
    for _ in range(n):
        y = rk4_step(cfg, y, p, h)
    return y


def box_sums(cfg, field):
    return np.array([field[cfg.box(s)].sum() for s in range(cfg.S)])


def featurize(cfg, y, prev_state=None):
    scale = 1.0 / (4 * (2 * cfg.hw + 1))
    w = int(np.floor(cfg.window_size / 2))
    rows = []
    for sp in range(2):
        sens = (box_sums(cfg, y[sp]) * scale).reshape(cfg.Sy, cfg.Sx)
        for i in range(-w, w + 1):
            for j in range(-w, w + 1):
                rows.append(np.roll(np.roll(sens, i, axis=0), j, axis=1).reshape(-1)[cfg.a2s])
    result = np.stack(rows)
    if cfg.temporal_steps > 1:
        if prev_state is None:
            result = np.concatenate([result] * cfg.temporal_steps)
        else:
            keep = prev_state.shape[0] - result.shape[0]
            result = np.concatenate([result, prev_state[:keep]])
    return result


def reward_function(cfg, y, action, delta_action):
    d = box_sums(cfg, y[0] - 1.0)[cfg.a2s] / (2 * cfg.hw + 1)
    r = -np.abs(d ** 2 / 800)
    a, da = np.asarray(action)[0, :], np.asarray(delta_action)[0, :]
    return r - cfg.action_punish * a ** 2 - cfg.delta_action_punish * da ** 2


def prepare_action(cfg, action):
    a = np.asarray(action, dtype=np.float64)[0, :]
    p = np.zeros((cfg.ny, cfg.nx))
    for i, s in enumerate(cfg.a2s):
        p[cfg.box(s)] += cfg.agent_power * a[i]
    return p
