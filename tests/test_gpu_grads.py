"""The fused MFMA passes' GRADIENTS against the oracle, directly.

`pdec_ddpg_critic_grads` / `pdec_ddpg_actor_grads` leave dL/dθ of src/PDEagent.jl:385-409 in the networks' flat gradient
buffers (`pdec_mlp_grad_buffer`, layout per layer `[W row-major [out][in] | b]`) -- the quantity that crosses the all-reduce.
The post-ADAM parameter checks of test_gpu_mlp.py are blind to a per-parameter gradient scale (first ADAM step = eta*sign(g)),
so a wrong 1/Bu, a factor in dq or a mis-applied grad_scale would pass them; here every gradient ARRAY is compared with the
fp64 restatement `oracle.nn.ddpg_losses_and_grads(...)["gC"]` / `actor_grads(...)["gA"]` on the same fp32 inputs.

Tolerance: fp32 gradients <= 1e-4 relative; applied per parameter array (W1, b1, W2, ...) against that array's
largest entry, which is stricter than "of the largest entry of the buffer"."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_mlp import make_net
from util import to_dev

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 1e-4


def flat_of(grads):
    """oracle gradient list [W1 [out,in], b1, ...] -> the library's flat order"""
    return np.concatenate([np.asarray(g, dtype=np.float64).ravel() for g in grads])


def read_grads(pkg, net):
    ptr, n = net.grad_buffer()
    assert n == net.num_params
    torch.cuda.synchronize()
    t = torch.as_tensor(pkg.distributed._DevArray(ptr, n, "<f4"), device=net.device)
    return t.cpu().numpy().astype(np.float64)


def assert_arrays_close(got_flat, want_list, what):
    o = 0
    worst = 0.0
    for i, w in enumerate(want_list):
        w = np.asarray(w, dtype=np.float64)
        g = got_flat[o:o + w.size].reshape(w.shape)
        o += w.size
        scale = np.abs(w).max()
        assert scale > 0, (what, i)
        err = np.abs(g - w).max() / scale
        worst = max(worst, err)
        assert err <= TOL, f"{what}: parameter array {i} {w.shape} off by {err:.3e} of its largest entry {scale:.3e}"
    assert o == got_flat.size
    return worst


# (name, ns, actor scale, critic scale, drop_middle_layer): the shapes of the fused passes
PAIRS = {
    "c2_3layer": (3, 1.6, 7.0, False),        # 3->16->16->1 / 4->140->140->1: ddpg_critic_fused_kernel / ddpg_actor_fused_kernel
    "c4_kseg2d": (36, 2.0, 17.0, True),       # 36->20->1 / 37->340->1: ddpg2_* (BASELINE configs[3])
    "kseg10_16": (12, 2.0, 17.0, True),       # 12->20->1 / 13->340->1 (Keller-Segel10_16.jl)
    "fluid": (9, 1.8, 17.0, True),            # 9->18->1 / 10->340->1 (FluidSetup.jl)
    "ks22": (1, 0.6, 7.0, True),              # 1->6->1 / 2->140->1 (KS22.jl)
}

# Bu = 131 072: config C3's update (1 024 workgroups; the batch-mean reward reduced by its own launch above 256 x 128 columns);
# Bu = 100 352: config C4's update (2-layer passes walking FOUR chunks of 128 columns per workgroup)
CASES = [("c2_3layer", 77), ("c2_3layer", 4096), ("c2_3layer", 32768), ("c2_3layer", 131072),
         ("c4_kseg2d", 1500), ("c4_kseg2d", 100352 // 8), ("c4_kseg2d", 100352), ("kseg10_16", 777), ("fluid", 300), ("fluid", 4096),
         ("ks22", 4100)]


def _inputs(rng, ns, Bu):
    s = rng.standard_normal((ns, Bu)).astype(np.float32)
    sn = rng.standard_normal((ns, Bu)).astype(np.float32)
    a = rng.uniform(-1, 1, (1, Bu)).astype(np.float32)
    r = -rng.uniform(0, 1, Bu).astype(np.float32)
    t = (rng.uniform(0, 1, Bu) < 0.1).astype(np.float32)
    return s, a, r, t, sn


def _away_from_relu_kinks(nn, PA, PC, aa, ac, s, a, r, t, sn, margin=2e-5):
    """A gradient is discontinuous where a ReLU pre-activation crosses zero: a column whose z sits within rounding of 0 may take
    the other branch in fp32 than in the fp64 oracle, and ONE such column moves a whole row of dW by |dq| |w| |x| -- at
    Bu = 100 352 (34 M pre-activations per forward) that happened (hidden unit 221: 1.2e-4 of the array's largest entry, every
    other entry <= 2e-7).  The comparison is about the arithmetic, not about that measure-zero set: columns with a ReLU
    pre-activation closer to 0 than `margin` in any forward the gradients go through -- C(s, a), A(s), C(s, A(s)) -- are
    replaced by copies of safe columns (same Bu)."""
    f64 = lambda P: [p.astype(np.float64) for p in P]
    S, A_ = s.astype(np.float64), a.astype(np.float64)

    def near(P, acts, x):
        _, zs, _ = nn.forward(f64(P), acts, x, keep=True)
        bad = np.zeros(x.shape[1], dtype=bool)
        for z, k in zip(zs, acts):
            if k == nn.RELU:
                bad |= (np.abs(z) < margin).any(axis=0)
        return bad

    bad = near(PC, ac, np.concatenate([S, A_])) | near(PA, aa, S)
    bad |= near(PC, ac, np.concatenate([S, nn.forward(f64(PA), aa, S)]))
    good = np.flatnonzero(~bad)
    assert good.size >= 0.9 * bad.size
    src = np.arange(bad.size)
    src[bad] = good[np.arange(int(bad.sum())) % good.size]
    return s[:, src], a[:, src], r[src], t[src], sn[:, src], int(bad.sum())


@pytest.mark.parametrize("grad_scale", [1.0, 0.5])
@pytest.mark.parametrize("quirk", [1, 0])
@pytest.mark.parametrize("name,Bu", CASES)
def test_fused_pass_gradients_match_the_oracle(pkg, name, Bu, quirk, grad_scale):
    """critic gradient (src/PDEagent.jl:385-400) and actor gradient through the critic (:402-409) of ONE update, read from
    the flat buffers right behind the fused passes, against the fp64 oracle on the same fp32 inputs and parameters"""
    from oracle import nn
    ns, sa, sc, drop = PAIRS[name]
    rng = np.random.default_rng(100 + Bu % 97 + quirk)
    da, aa = nn.layer_sizes(ns, 1, sa, True, drop)
    dc, ac = nn.layer_sizes(ns, 1, sc, False, drop)
    dtype = torch.float32
    A, PA = make_net(pkg, rng, da, aa, dtype, Bu)
    Cn, PC = make_net(pkg, rng, dc, ac, dtype, Bu)
    At, PAt = make_net(pkg, rng, da, aa, dtype, Bu)
    Ct, PCt = make_net(pkg, rng, dc, ac, dtype, Bu)
    s, a, r, t, sn = _inputs(rng, ns, Bu)
    s, a, r, t, sn, _ = _away_from_relu_kinks(nn, PA, PC, aa, ac, s, a, r, t, sn)
    f64 = lambda P: [p.astype(np.float64) for p in P]
    g32 = np.float64(np.float32(0.99))
    out = nn.ddpg_losses_and_grads(f64(PA), f64(PC), f64(PAt), f64(PCt), aa, ac, s.astype(np.float64), a.astype(np.float64),
                                   r.astype(np.float64), t.astype(np.float64), sn.astype(np.float64), g32, bool(quirk))
    out2 = nn.actor_grads(f64(PA), f64(PC), aa, ac, s.astype(np.float64))
    L = pkg._lib
    ds, da_, dr, dt_, dsn = (to_dev(s.T, dtype), to_dev(a.T, dtype), to_dev(r, dtype), to_dev(t, dtype), to_dev(sn.T, dtype))
    losses = torch.zeros(2, dtype=dtype, device="cuda:0")
    L.check(A.lib.pdec_ddpg_critic_grads(A.handle, Cn.handle, At.handle, Ct.handle, L.ptr(ds), L.ptr(da_), L.ptr(dr), L.ptr(dt_),
                                         L.ptr(dsn), Bu, 0.99, quirk, grad_scale, C.c_void_p(losses.data_ptr())))
    gC = read_grads(pkg, Cn)
    L.check(A.lib.pdec_ddpg_actor_grads(A.handle, Cn.handle, L.ptr(ds), Bu, grad_scale, C.c_void_p(losses.data_ptr() + 4)))
    gA = read_grads(pkg, A)
    assert np.isfinite(gC).all() and np.isfinite(gA).all()
    assert_arrays_close(gC, [grad_scale * g for g in out["gC"]], f"critic gradient {name} Bu={Bu} quirk={quirk} scale={grad_scale}")
    assert_arrays_close(gA, [grad_scale * g for g in out2["gA"]], f"actor gradient {name} Bu={Bu} scale={grad_scale}")
    # a scale error cannot hide: the norms agree as well (1e-4), and grad_scale really scaled the buffer
    assert abs(np.linalg.norm(gC) / np.linalg.norm(grad_scale * flat_of(out["gC"])) - 1.0) <= TOL
    assert abs(np.linalg.norm(gA) / np.linalg.norm(grad_scale * flat_of(out2["gA"])) - 1.0) <= TOL
    lv = losses.cpu().numpy()
    assert abs(lv[0] - out["critic_loss"]) <= 2e-5 * max(1.0, abs(out["critic_loss"]))
    assert abs(lv[1] - out2["actor_loss"]) <= 2e-5 * max(1.0, abs(out2["actor_loss"]))


def test_fused_and_generic_paths_leave_the_same_gradient(pkg, monkeypatch):
    """the 4-launch single-GPU form reduces the SAME slabs inside its finish kernel: after pdec_ddpg_critic_grads the flat
    buffer holds exactly what the finish kernel of pdec_ddpg_update_async would have applied -- checked through ADAM's first
    step, whose size is eta * g / (|g| + eps'): parameters move against the sign of the oracle gradient wherever |g| is not
    tiny"""
    from oracle import nn
    rng = np.random.default_rng(8)
    ns, Bu = 3, 4096
    da, aa = nn.layer_sizes(ns, 1, 1.6, True, False)
    dc, ac = nn.layer_sizes(ns, 1, 7.0, False, False)
    dtype = torch.float32
    A, PA = make_net(pkg, rng, da, aa, dtype, Bu)
    Cn, PC = make_net(pkg, rng, dc, ac, dtype, Bu)
    At, PAt = make_net(pkg, rng, da, aa, dtype, Bu)
    Ct, PCt = make_net(pkg, rng, dc, ac, dtype, Bu)
    s, a, r, t, sn = _inputs(rng, ns, Bu)
    f64 = lambda P: [p.astype(np.float64) for p in P]
    out = nn.ddpg_losses_and_grads(f64(PA), f64(PC), f64(PAt), f64(PCt), aa, ac, s.astype(np.float64), a.astype(np.float64),
                                   r.astype(np.float64), t.astype(np.float64), sn.astype(np.float64),
                                   np.float64(np.float32(0.99)), True)
    L = pkg._lib
    ds, da_, dr, dt_, dsn = (to_dev(s.T, dtype), to_dev(a.T, dtype), to_dev(r, dtype), to_dev(t, dtype), to_dev(sn.T, dtype))
    L.check(A.lib.pdec_ddpg_update_async(A.handle, Cn.handle, At.handle, Ct.handle, L.ptr(ds), L.ptr(da_), L.ptr(dr), L.ptr(dt_),
                                         L.ptr(dsn), Bu, 0.99, 0.995, 1, 5e-4, 1e-3, None))
    torch.cuda.synchronize()
    moved = [new.astype(np.float64) - old.astype(np.float64) for new, old in zip(Cn.params(), PC)]
    for dlt, g in zip(moved, out["gC"]):
        big = np.abs(g) > 1e-3 * np.abs(g).max()
        assert big.any()
        assert np.array_equal(np.sign(dlt[big]), -np.sign(g[big]))
        assert np.abs(np.abs(dlt[big]) - 1e-3).max() <= 2e-5       # first Flux-ADAM step: eta * g / (|g| + eps*sqrt(1-b2))
