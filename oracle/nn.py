"""Weight-shared ("convolutional") actor/critic MLPs, DDPG update, ADAM, Polyak -- NumPy
restatement.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Single operations are pinned by
finite-difference checks and the committed torch-autograd golden (tests/golden/nn_torch_golden.npz,
generator tests/golden/make_nn_golden.py); the update as a whole is pinned by the reference's own
training artifacts (round 5): the ADAM step count / Float64 beta powers of agent.jld2 bit for bit and the
learning curves of hook.jld2 in a band over seeds (tests/test_gpu_training.py), which is how the
frozen-target behaviour below was found.

The reference's Polyak loop (src/PDEagent.jl:415-417) never runs: Flux.params([At, Ct]) is empty because
src/custom_nna.jl:20 defines a `functor` of its own.  Callers reproduce the reference as it runs by passing
rho = 1 to ddpg_update / polyak (dest = 1 * dest + 0 * src), the loop as written with rho = 0.995.

Layout follows Flux: W[out, in], b[out]; a batch is a matrix [features, columns] and a Dense
layer applied to it shares weights across columns (src/PDEagent.jl:14-56)."""
import numpy as np

RELU, TANH, IDENT = 1, 2, 0


def act(z, kind):
    if kind == RELU:
        return np.maximum(z, 0)
    if kind == TANH:
        return np.tanh(z)
    return z


def act_grad(z, a, kind):
    if kind == RELU:
        return (z > 0).astype(z.dtype)
    if kind == TANH:
        return 1 - a * a
    return np.ones_like(z)


def layer_sizes(ns, na, nna_scale, is_actor, drop_middle_layer):
    """create_NNA, src/PDEagent.jl:14-44."""
    if is_actor:
        h = int(np.floor(10 * nna_scale))
        dims = [ns, h, na] if drop_middle_layer else [ns, h, h, na]
        acts = [RELU, TANH] if drop_middle_layer else [RELU, RELU, TANH]
    else:
        h = int(np.floor(20 * nna_scale))
        dims = [ns + na, h, 1] if drop_middle_layer else [ns + na, h, h, 1]
        acts = [RELU, IDENT] if drop_middle_layer else [RELU, RELU, IDENT]
    return dims, acts


def glorot_uniform(rng, dims, dtype=np.float32):
    """Flux.glorot_uniform: U(+-sqrt(6/(in+out))), zero bias (src/PDEagent.jl:66)."""
    params = []
    for i in range(len(dims) - 1):
        lim = np.sqrt(6.0 / (dims[i] + dims[i + 1]))
        params.append(rng.uniform(-lim, lim, (dims[i + 1], dims[i])).astype(dtype))
        params.append(np.zeros(dims[i + 1], dtype=dtype))
    return params


def forward(params, acts, x, keep=False):
    """Chain(Dense...)(x), x [in, cols] (src/custom_nna.jl:13)."""
    zs, as_ = [], [x]
    a = x
    for li, kind in enumerate(acts):
        W, b = params[2 * li], params[2 * li + 1]
        z = W @ a + b[:, None]
        a = act(z, kind)
        zs.append(z)
        as_.append(a)
    return (a, zs, as_) if keep else a


def backward(params, acts, zs, as_, dy):
    """Returns (grads list like params, dx)."""
    grads = [None] * len(params)
    d = dy
    for li in reversed(range(len(acts))):
        dz = d * act_grad(zs[li], as_[li + 1], acts[li])
        grads[2 * li] = dz @ as_[li].T
        grads[2 * li + 1] = dz.sum(axis=1)
        d = params[2 * li].T @ dz
    return grads, d


class Adam:
    """Flux.Optimise.ADAM (0.13): per-parameter (mt, vt, beta_powers); constants recovered
    from scripts/KS/KS22/saves/agent.jld2: beta=(0.9,0.999), eps=1e-8."""

    def __init__(self, params, eta, beta=(0.9, 0.999), eps=1e-8):
        self.eta, self.beta, self.eps = eta, beta, eps
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]
        self.bp = [np.array(beta, dtype=np.float64) for _ in params]

    def step(self, params, grads):
        b1, b2 = self.beta
        for i, (p, g) in enumerate(zip(params, grads)):
            dt = p.dtype.type
            self.m[i] = dt(b1) * self.m[i] + dt(1 - b1) * g
            self.v[i] = dt(b2) * self.v[i] + dt(1 - b2) * g * g
            bp = self.bp[i]
            delta = self.m[i] / dt(1 - bp[0]) / (np.sqrt(self.v[i] / dt(1 - bp[1])) + dt(self.eps)) * dt(self.eta)
            self.bp[i] = bp * np.array(self.beta)
            params[i] = (p - delta).astype(p.dtype)
        return params


def polyak(dst, src, rho):
    """src/PDEagent.jl:415-417: dest = rho*dest + (1-rho)*src."""
    return [(d.dtype.type(rho) * d + d.dtype.type(1 - rho) * s).astype(d.dtype) for d, s in zip(dst, src)]


def ddpg_losses_and_grads(A, C, At, Ct, acts_a, acts_c, s, a, r, t, snext, gamma, quirk=True):
    """One DDPG update's gradients (src/PDEagent.jl:363-409) WITHOUT applying them.
    s,snext [ns,Bu]; a [na,Bu]; r [Bu] (the reference holds it as 1xBu); t [Bu].
    quirk=True reproduces the reference's (1xBu) .+ (Bu) broadcast at :388/:393: the loss is
    mean_{i,j}(r_j + gamma(1-t_i)qt_i - q_i)^2; quirk=False is the usual
    diagonal TD loss.  Returns dict with critic grads, the actor grads computed with the
    critic passed in `C_after` semantics handled by ddpg_update."""
    dt = s.dtype
    Bu = s.shape[1]
    anext = forward(At, acts_a, snext)                                        # :385
    qt = forward(Ct, acts_c, np.concatenate([snext, anext])).reshape(-1)      # :386
    tgt_i = dt.type(gamma) * (1 - t.astype(dt)) * qt                          # per-sample part of :388
    q, zs, as_ = forward(C, acts_c, np.concatenate([s, a]), keep=True)        # :392
    q = q.reshape(-1)
    if quirk:
        d = tgt_i - q
        if Bu <= 2048:
            e = r[None, :] + d[:, None]                                       # [i,j]
            closs = np.mean(e ** 2)                                           # :393
        else:
            # the same double mean without the Bu x Bu matrix (8.6 GB at Bu = 32 768):
            # mean_ij (r_j + d_i)^2 = mean(r^2) + 2 mean(r) mean(d) + mean(d^2); fp64 accumulation
            r64, d64 = r.astype(np.float64), d.astype(np.float64)
            closs = dt.type(np.mean(r64 * r64) + 2.0 * r64.mean() * d64.mean() + np.mean(d64 * d64))
        dq = -(2.0 / Bu) * (r.mean() + tgt_i - q)
    else:
        e = r + tgt_i - q
        closs = np.mean(e ** 2)
        dq = -(2.0 / Bu) * e
    gC, _ = backward(C, acts_c, zs, as_, dq[None, :].astype(dt))
    return dict(critic_loss=closs, gC=gC, qt=qt, q=q)


def actor_grads(A, C, acts_a, acts_c, s):
    """src/PDEagent.jl:402-409: loss = -mean(C(vcat(s, A(s))))."""
    dt = s.dtype
    Bu = s.shape[1]
    ns = s.shape[0]
    aout, zsa, asa = forward(A, acts_a, s, keep=True)
    q, zs, as_ = forward(C, acts_c, np.concatenate([s, aout]), keep=True)
    aloss = -np.mean(q)
    dq = np.full((1, Bu), -1.0 / Bu, dtype=dt)
    _, dx = backward(C, acts_c, zs, as_, dq)
    gA, _ = backward(A, acts_a, zsa, asa, dx[ns:])
    return dict(actor_loss=aloss, gA=gA)


def ddpg_update(A, C, At, Ct, optA, optC, acts_a, acts_c, s, a, r, t, snext, gamma, rho, quirk=True):
    """Full update, src/PDEagent.jl:363-418: critic grad+ADAM, THEN actor grad through the
    UPDATED critic + ADAM, then Polyak of both targets.  Lists are updated in place."""
    out = ddpg_losses_and_grads(A, C, At, Ct, acts_a, acts_c, s, a, r, t, snext, gamma, quirk)
    C[:] = optC.step(C, out["gC"])                                            # :400
    out2 = actor_grads(A, C, acts_a, acts_c, s)
    A[:] = optA.step(A, out2["gA"])                                           # :412
    At[:] = polyak(At, A, rho)                                                # :415-417
    Ct[:] = polyak(Ct, C, rho)
    out.update(out2)
    return out


def policy_act(A, acts_a, state, noise, act_noise, act_limit, learning=True, memory_size=0):
    """(policy::CustomDDPGPolicy)(env), src/PDEagent.jl:183-207.  The reference promotes Float32 weights to the Float64
    state (acting path is fp64).  memory_size > 0: noise on actions[1:end-memory_size, :] only (:201)."""
    P = [p.astype(np.float64) for p in A]
    actions = forward(P, acts_a, np.asarray(state, dtype=np.float64))         # :189
    if learning:
        actions = np.array(actions, dtype=np.float64)
        k = actions.shape[0] - int(memory_size)
        actions[:k] = actions[:k] + np.asarray(noise)[:k] * act_noise         # :201
    return np.clip(actions, -act_limit, act_limit)                            # :202-204
