"""Checkpoint I/O.

* `read_hook(path)`: read a reference `saves/hook.jld2` (written by `FileIO.save(..., "hook", hook)`,
  scripts/KS/setup/KSSetup.jl:391-402): the best / current actor weights (`hook.bestNNA`,
  `hook.currentNNA`, src/PDEhook.jl:68-75), the per-episode rewards and, when the hook collected it,
  the best episode's `bestDF` trajectory (src/PDEhook.jl:54-62).  Pure Python (jld2.py).
* `load_actor(nna_or_model, params)`: `copyto!(actor, hook.bestNNA)` (src/plotting.jl:29).
* `save_agent` / `load_agent`: native `.npz` checkpoint of the four networks plus their ADAM state
  (moments and beta powers), i.e. what `FileIO.save(".../agent.jld2", "agent", agent)` keeps of the
  learner.
* `save_agent_jld2` / `save_actor_jld2`: the same arrays as plain datasets in a JLD2 container (jld2.write_arrays), which
  JLD2.jl reads back as ordinary Julia arrays; julia/load_agent_arrays.jl rebuilds the Flux chains and ADAM states from
  them, so a build-trained agent reaches the reference's `load()` / `plot_heat` path (scripts/KS/setup/KSSetup.jl:378-402,
  src/plotting.jl:26-31).  (The reference's own files serialise whole Julia structs -- `Agent`, `PDEhook` with their
  committed datatypes; reproducing that type graph without Julia is out of reach, plain arrays are the interchange form.)"""
import numpy as np

from . import _lib
from .jld2 import JLD2File


def read_hook(path):
    f = JLD2File(path)
    out = {}
    f32 = [f.array(o) for o in f.numeric(1, 4) if o.dims]
    half = len(f32) // 2
    for pi, pref in enumerate(("best", "current")):          # file order: bestNNA then currentNNA, W,b,W,b...
        out[pref] = [np.ascontiguousarray(a) for a in f32[pi * half:(pi + 1) * half]]
    rew = [f.array(o) for o in f.numeric(1, 8) if o.dims and len(o.dims) == 1]
    refs = [o for o in f.ref_arrays() if o.dims and o.dims[0] > 5]
    if refs:                                                   # bestDF columns in insertion order (PDEhook.jl:56-60)
        n = refs[0].dims[0]
        cols = [np.stack([f.array(f.deref(r)) for r in f.array(o)]).astype(np.float64)
                for o in [o for o in refs if o.dims[0] == n][:4]]
        out["bestDF"] = dict(zip(("action", "p", "y", "reward"), cols))
        widths = {c.shape[1] for c in cols if c.ndim == 2}
        rew = [r for r in rew if r.size not in widths]
    out["rewards"] = max(rew, key=lambda r: r.size) if rew else np.zeros(0)
    return out


def load_actor(nna, params):
    """params: [W1, b1, W2, b2, ...] with W[out, in] (Flux.params order)"""
    model = getattr(nna, "model", nna)
    model.set_params([np.asarray(p) for p in params])
    return nna


def _adam_state(model):
    import ctypes as C
    n = model.num_params
    m = np.empty(n, dtype=model.np_dtype)
    v = np.empty(n, dtype=model.np_dtype)
    bp = (C.c_double * 2)()
    _lib.check(model.lib.pdec_adam_get_state(model.handle, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), bp))
    return m, v, np.array([bp[0], bp[1]])


def _json_default(o):
    """bit-generator states hold numpy scalars and, for MT19937-like generators, arrays"""
    if isinstance(o, np.ndarray):
        return {"__ndarray__": o.tolist(), "dtype": str(o.dtype)}
    if isinstance(o, np.generic):
        return o.item()
    return int(o)


def _json_hook(d):
    if "__ndarray__" in d:
        return np.asarray(d["__ndarray__"], dtype=d["dtype"])
    return d


def save_agent(path, agent, with_trajectory=True):
    """what `FileIO.save(".../agent.jld2", "agent", agent)` keeps (scripts/KS/setup/KSSetup.jl:391-402): the four networks,
    their ADAM states, the replay trajectory, plus the positions of the build's counter-based random streams -- a resumed
    run continues the exploration noise and the minibatch sampling where the saved one stopped"""
    p = agent.policy
    kw = {}
    for name in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        model = getattr(p, name).model
        for i, a in enumerate(model.params()):
            kw[f"{name}/p{i}"] = a
        m, v, bp = _adam_state(model)
        kw[f"{name}/adam_m"], kw[f"{name}/adam_v"], kw[f"{name}/adam_beta_pow"] = m, v, bp
        kw[f"{name}/dims"] = np.array(model.dims)
    kw["update_step"] = np.array(p.update_step)
    kw["act_noise"] = np.array(p.act_noise)
    kw["rho_effective"] = np.array(p.rho_effective)       # 1.0 = the targets never moved (quirk_frozen_targets), else the Polyak factor
    # exploration-noise and minibatch-sampling streams (counter-based: seed + offset is the whole state), the host rng
    kw["noise_seed_off"] = np.array([p._noise_seed, p._noise_off], dtype=np.uint64)
    kw["sample_seed_off"] = np.array([p._sample_seed, p._sample_off], dtype=np.uint64)
    # ... and the device-resident counter TrainPipeline's acting kernel advances (pdec_policy_act_rng_dev)
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()        # the acting kernel that advances it runs on the environment's stream
    ctr = C.c_uint64()
    _lib.check(p.lib.pdec_noise_counter_get(p.behavior_actor.model.handle, C.byref(ctr)))
    kw["noise_counter_dev"] = np.array([ctr.value], dtype=np.uint64)
    import json
    kw["rng_state"] = np.array(json.dumps(p.rng.bit_generator.state, default=_json_default))
    if with_trajectory:                 # the replay traces, as FileIO.save of the whole Agent keeps them
        tr = agent.trajectory
        n_sa, n_rt = min(tr.n_sa, tr.capacity + tr.stride), min(tr.n_rt, tr.capacity)
        kw["trajectory/counters"] = np.array([tr.n_sa, tr.n_rt, tr.capacity, tr.stride], dtype=np.int64)
        full = tr.n_rt >= tr.capacity
        kw["trajectory/state"] = (tr.state if full else tr.state[:n_sa]).cpu().numpy()
        kw["trajectory/action"] = (tr.action if full else tr.action[:n_sa]).cpu().numpy()
        kw["trajectory/reward"] = (tr.reward if full else tr.reward[:n_rt]).cpu().numpy()
        kw["trajectory/terminal"] = (tr.terminal if full else tr.terminal[:n_rt]).cpu().numpy()
    np.savez_compressed(path, **kw)


def load_agent(path, agent):
    import ctypes as C
    z = np.load(path)
    p = agent.policy
    for name in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        model = getattr(p, name).model
        if list(z[f"{name}/dims"]) != list(model.dims):
            raise _lib.PdecError(f"checkpoint {path}: {name} has dims {list(z[f'{name}/dims'])}, agent has {model.dims}")
        n = len(model.dims) - 1
        model.set_params([z[f"{name}/p{i}"] for i in range(2 * n)])
        m = np.ascontiguousarray(z[f"{name}/adam_m"], dtype=model.np_dtype)
        v = np.ascontiguousarray(z[f"{name}/adam_v"], dtype=model.np_dtype)
        bp = (C.c_double * 2)(*z[f"{name}/adam_beta_pow"])
        _lib.check(model.lib.pdec_adam_set_state(model.handle, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), bp))
    p.update_step = int(z["update_step"])
    p.act_noise = float(z["act_noise"])
    if "rho_effective" in z.files and float(z["rho_effective"]) != float(p.rho_effective):
        import warnings
        from .agent import TargetNetworkWarning
        warnings.warn(f"checkpoint {path} was written under rho_effective = {float(z['rho_effective']):g} "
                      f"({'frozen' if float(z['rho_effective']) == 1.0 else 'moving'} target networks), this agent runs with "
                      f"{float(p.rho_effective):g}: its target networks come from the other regime", TargetNetworkWarning, stacklevel=2)
    if "noise_seed_off" in z.files:
        p._noise_seed, p._noise_off = (int(x) for x in z["noise_seed_off"])
        p._sample_seed, p._sample_off = (int(x) for x in z["sample_seed_off"])
        if "noise_counter_dev" in z.files:
            _lib.check(p.lib.pdec_noise_counter_set(p.behavior_actor.model.handle, int(z["noise_counter_dev"][0])))
        import json
        st = json.loads(str(z["rng_state"]), object_hook=_json_hook)
        try:
            p.rng.bit_generator.state = st
        except (ValueError, TypeError):
            pass                          # a different bit generator than the one that was saved: keep the current stream
    if "trajectory/counters" in z.files:
        import torch
        tr = agent.trajectory
        n_sa, n_rt, cap, stride = (int(x) for x in z["trajectory/counters"])
        if (cap, stride) != (tr.capacity, tr.stride):
            raise _lib.PdecError(f"checkpoint {path}: trajectory capacity/stride {cap}/{stride}, agent has {tr.capacity}/{tr.stride}")
        for name in ("state", "action", "reward", "terminal"):
            a = torch.as_tensor(z[f"trajectory/{name}"], device=tr.device)
            getattr(tr, name)[:a.shape[0]].copy_(a)
        tr.n_sa, tr.n_rt = n_sa, n_rt
    return agent


def _net_arrays(prefix, model, with_adam=True):
    """{name: array} of one network in Julia shapes: `<prefix>_W1` [out, in], `<prefix>_b1`, ... (+ ADAM moments)"""
    out = {}
    params = model.params()
    for li in range(len(params) // 2):
        out[f"{prefix}_W{li + 1}"] = np.asarray(params[2 * li], dtype=np.float32)
        out[f"{prefix}_b{li + 1}"] = np.asarray(params[2 * li + 1], dtype=np.float32)
    if with_adam:
        m, v, bp = _adam_state(model)
        out[f"{prefix}_adam_m"], out[f"{prefix}_adam_v"] = m.astype(np.float32), v.astype(np.float32)
        out[f"{prefix}_adam_beta_pow"] = np.asarray(bp, dtype=np.float64)
    out[f"{prefix}_dims"] = np.asarray(model.dims, dtype=np.int64)
    out[f"{prefix}_acts"] = np.asarray(model.acts, dtype=np.int64)        # 0 identity, 1 relu, 2 tanh
    return out


def save_agent_jld2(path, agent):
    """the four networks + ADAM states of `agent` as plain arrays in a JLD2 file (see the module docstring)"""
    from .jld2 import write_arrays
    p = agent.policy
    arrays = {}
    for name in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        arrays.update(_net_arrays(name, getattr(p, name).model))
    # hyper = [gamma, rho (the policy's field p), act_limit, act_noise, eta_actor, eta_critic, rho_effective]: the last entry is the
    # Polyak factor the update kernels RECEIVED -- 1.0 under quirk_frozen_targets (the targets in this file are then the initial
    # networks, as in the reference's own agent.jld2), rho otherwise (a consumer can tell the two regimes apart)
    arrays["hyper"] = np.array([p.y, p.p, p.act_limit, p.act_noise, p.behavior_actor.optimizer.eta,
                                p.behavior_critic.optimizer.eta, p.rho_effective], dtype=np.float64)
    write_arrays(path, arrays)


def save_actor_jld2(path, nna, name="bestNNA"):
    """one actor (e.g. PDEhook.bestNNA, src/PDEhook.jl:68-75) as `<name>_W1`, `<name>_b1`, ..."""
    from .jld2 import write_arrays
    write_arrays(path, _net_arrays(name, getattr(nna, "model", nna), with_adam=False))
