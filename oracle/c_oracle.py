"""ctypes wrapper of oracle/c/pdec_oracle.c (built into oracle/_build/libpdec_oracle.so).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): checker for tests/ and the timed CPU
baseline ("port") of bench.py."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libpdec_oracle.so")
_lib = None


class EnvTabs(C.Structure):
    _fields_ = [("N", C.c_int), ("S", C.c_int), ("A", C.c_int), ("window", C.c_int),
                ("max_value", C.c_double), ("agent_power", C.c_double), ("action_punish", C.c_double),
                ("delta_action_punish", C.c_double), ("G", C.c_void_p), ("Ga", C.c_void_p), ("a2s", C.c_void_p)]


def load(build_if_missing=True):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO) and build_if_missing:
        subprocess.run(["make", "-C", os.path.join(HERE, "c")], check=True)
    lib = C.CDLL(SO)
    lib.ks_plan_create.restype = C.c_void_p
    lib.ks_plan_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_double]
    lib.ks_plan_destroy.argtypes = [C.c_void_p]
    lib.ks_step.argtypes = [C.c_void_p] * 4
    lib.ks_env_step_batch.argtypes = [C.c_void_p, C.POINTER(EnvTabs), C.c_int] + [C.c_void_p] * 6
    lib.agent_create.restype = C.c_void_p
    lib.agent_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.agent_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.agent_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.agent_nparams.argtypes = [C.c_void_p, C.c_int]
    lib.agent_act.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]
    lib.agent_ddpg_update.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.oracle_set_threads.argtypes = [C.c_int]
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class KSPlan:
    def __init__(self, N, Lx, dt=0.1, K=30, mu=0.0):
        self.lib = load()
        self.N = N
        self.h = self.lib.ks_plan_create(N, Lx, dt, K, mu)
        if not self.h:
            raise ValueError(f"N={N}: unsupported prime factor")

    def step(self, y, p):
        y, p = np.ascontiguousarray(y, dtype=np.float64), np.ascontiguousarray(p, dtype=np.float64)
        out = np.empty(self.N)
        self.lib.ks_step(self.h, _p(y), _p(p), _p(out))
        return out


class KSBatchEnv:
    """Batched fp64 KS env step (prepare_action, CNAB2, reward, featurize, blow-up flag)"""

    def __init__(self, cfg):
        """cfg: oracle.ks.KSConfig"""
        self.cfg = cfg
        self.plan = KSPlan(cfg.nx, cfg.Lx, cfg.dt, cfg.oversampling, cfg.mu)
        self.G = np.ascontiguousarray(cfg.gaussians)
        self.Ga = np.ascontiguousarray(cfg.gaussians_actuators)
        self.a2s = np.ascontiguousarray(cfg.actuators_to_sensors - 1, dtype=np.int32)
        self.t = EnvTabs(cfg.nx, len(cfg.sensor_positions), len(cfg.actuator_positions), cfg.window_size,
                         cfg.max_value, cfg.agent_power, cfg.action_punish, cfg.delta_action_punish,
                         self.G.ctypes.data, self.Ga.ctypes.data, self.a2s.ctypes.data)

    def step(self, y, action, action_prev):
        """y [B,N] updated IN PLACE; returns state [B,A,ns], reward [B,A], done [B]"""
        B, A, ns = y.shape[0], self.t.A, self.t.window
        state, reward = np.empty((B, A, ns)), np.empty((B, A))
        done = np.zeros(B, dtype=np.int32)
        action, action_prev = np.ascontiguousarray(action, dtype=np.float64), np.ascontiguousarray(action_prev, dtype=np.float64)
        self.plan.lib.ks_env_step_batch(self.plan.h, C.byref(self.t), B, _p(y), _p(action), _p(action_prev),
                                        _p(state), _p(reward), _p(done))
        return state, reward, done


class Agent:
    """fp64 actor/critic/targets with ADAM + Polyak; params are lists [W1,b1,...] with W [out,in]"""

    def __init__(self, dimsA, actsA, dimsC, actsC):
        self.lib = load()
        self.dimsA, self.dimsC = list(dimsA), list(dimsC)
        ia = lambda v: np.ascontiguousarray(v, dtype=np.int32)
        self._keep = (ia(dimsA), ia(actsA), ia(dimsC), ia(actsC))
        k = self._keep
        self.h = self.lib.agent_create(len(actsA), _p(k[0]), _p(k[1]), len(actsC), _p(k[2]), _p(k[3]))

    @staticmethod
    def _flat(params):
        return np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64).ravel() for p in params]))

    def _unflat(self, flat, dims):
        out, o = [], 0
        for i in range(len(dims) - 1):
            n = dims[i] * dims[i + 1]
            out.append(flat[o:o + n].reshape(dims[i + 1], dims[i]).copy()); o += n
            out.append(flat[o:o + dims[i + 1]].copy()); o += dims[i + 1]
        return out

    def set(self, which, params):     # which: 0 A, 1 C, 2 At, 3 Ct
        f = self._flat(params)
        self.lib.agent_set(self.h, which, _p(f))

    def get(self, which):
        f = np.empty(self.lib.agent_nparams(self.h, which))
        self.lib.agent_get(self.h, which, _p(f))
        return self._unflat(f, self.dimsA if which in (0, 2) else self.dimsC)

    def act(self, state, noise, act_noise, lim):
        """state [cols, ns] -> actions [cols, na]"""
        state = np.ascontiguousarray(state, dtype=np.float64)
        cols = state.shape[0]
        out = np.empty((cols, self.dimsA[-1]))
        nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float64)
        self.lib.agent_act(self.h, _p(state), None if nz is None else _p(nz), cols, act_noise, lim, _p(out))
        return out

    def ddpg_update(self, s, a, r, t, snext, gamma, rho, quirk, eta_a, eta_c):
        c = lambda v: np.ascontiguousarray(v, dtype=np.float64)
        s, a, r, t, snext = c(s), c(a), c(r), c(t), c(snext)
        al, cl = C.c_double(), C.c_double()
        self.lib.agent_ddpg_update(self.h, _p(s), _p(a), _p(r), _p(t), _p(snext), s.shape[0], gamma, rho, int(quirk),
                                   eta_a, eta_c, C.byref(al), C.byref(cl))
        return al.value, cl.value
