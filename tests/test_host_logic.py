"""CPU tests of the host-side mirror: C-ABI symbol export, trajectory glue, run-loop stop
conditions, sharding, and the world_size-2 gradient all-reduce over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol(pkg):
    """the shared library loads without a GPU and exports every function include/pdeconv.h declares -- and the unit-test /
    measurement entry points of include/pdeconv_debug.h, which are NOT part of the drop-in surface: no
    `pdec_debug_*` name is left in the public header, and the Julia glue binds none of them"""
    lib = ctypes.CDLL(pkg._lib.LIB_PATH)
    for header, sigs, extra in (("pdeconv.h", pkg._lib.SIGNATURES, {"pdec_last_error"}),
                                ("pdeconv_debug.h", pkg._lib.DEBUG_SIGNATURES, set())):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        declared = set(re.findall(r"\b(pdec_[a-z0-9_]+)\s*\(", hdr))
        assert len(declared) >= (40 if header == "pdeconv.h" else 4)
        missing = [n for n in sorted(declared) if not hasattr(lib, n)]
        assert not missing, missing
        # and the ctypes binding covers the same set
        bound = set(sigs) | extra
        assert declared == bound, declared ^ bound
        assert all(n.startswith("pdec_debug_") for n in declared) == (header == "pdeconv_debug.h")
        assert not any(n.startswith("pdec_debug_") for n in declared) or header == "pdeconv_debug.h"
    for jl in os.listdir(os.path.join(ROOT, "julia")):
        assert "pdec_debug_" not in open(os.path.join(ROOT, "julia", jl)).read(), jl


def _c_decls():
    hdr = open(os.path.join(ROOT, "include", "pdeconv.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(pdec_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return hdr, decls


def test_julia_glue_matches_the_c_abi(pkg):
    """julia/PDEenvHIP.jl cannot be executed here (no Julia in the image), so its contract with include/pdeconv.h is
    checked textually: the EnvCfg struct mirror has the C struct's fields in order and type (and so has the ctypes
    mirror), the keyword constructor passes exactly those fields in that order, every `ccall` names a declared
    symbol and passes as many arguments as the C prototype has parameters."""
    hdr, decls = _c_decls()
    body = re.search(r"typedef struct pdec_env_cfg \{(.*?)\} pdec_env_cfg;", hdr, flags=re.S).group(1)
    c_fields = re.findall(r"\b(int|double)\s+([A-Za-z_0-9]+)\s*;", body)
    assert len(c_fields) == 30
    jl = open(os.path.join(ROOT, "julia", "PDEenvHIP.jl")).read()
    jbody = re.search(r"^struct EnvCfg\n(.*?)^end", jl, flags=re.S | re.M).group(1)
    j_fields = re.findall(r"^\s*([A-Za-z_0-9]+)::(Cint|Cdouble)\s*$", jbody, flags=re.M)
    assert [(n, {"Cint": "int", "Cdouble": "double"}[t]) for n, t in j_fields] == [(n, t) for t, n in c_fields]
    py_fields = [(n, "int" if t is ctypes.c_int else "double") for n, t in pkg._lib.EnvCfg._fields_]
    assert py_fields == [(n, t) for t, n in c_fields]
    assert ctypes.sizeof(pkg._lib.EnvCfg) == 12 * 4 + 12 * 8 + 2 * 4 + 8 + 3 * 4 + 4      # (+ 4 bytes of tail padding)
    # the keyword constructor forwards every field positionally in the struct's order
    ctor = re.search(r"function EnvCfg\(;.*?\n    EnvCfg\((.*?)\)\nend", jl, flags=re.S).group(1)
    assert [a.strip() for a in ctor.replace("\n", " ").split(",")] == [n for _, n in c_fields]
    # every ccall: declared symbol, matching arity (argument-type tuple and actual arguments)
    calls = re.findall(r"ccall\(\(:(pdec_[a-z0-9_]+), LIB\),\s*(\w+),\s*\((.*?)\)\s*(?:,|\))", jl, flags=re.S)
    assert len(calls) >= 20
    for name, ret, tup in calls:
        assert name in decls, name
        n = len([a for a in tup.replace("\n", " ").split(",") if a.strip()])
        assert n == decls[name], (name, n, decls[name])
        assert ret == ("Cstring" if name == "pdec_last_error" else "Cint"), name
    for extra in ("KSSetupHIP.jl",):
        for name in re.findall(r":(pdec_[a-z0-9_]+)", open(os.path.join(ROOT, "julia", extra)).read()):
            assert name in decls


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two fresh rank processes and prints ONE line with
    the rank count it observed (--launch-check: rendezvous only, so it runs without a GPU)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    import json
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_observed"] == 2 and d["self_launched"] is True


def test_product_does_not_import_oracle():
    """the product path must not route through the oracle or any CPU fallback"""
    pdir = os.path.join(ROOT, "distributedconvrl-pde-control_amd")
    for dp, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_no_gpu_means_loud_failure(pkg):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.PdecError):
        pkg.PDEenv(pkg.KSSetup.KS22(), B=1, device="cpu")
    with pytest.raises(pkg.PdecError):
        pkg._lib.init(0)


def test_trajectory_layout_and_sampling(pkg):
    """push/pop/sample follow src/PDEagent.jl:237-340 (next state at +stride, dummy rows popped)"""
    A = 4
    tr = pkg.CircularArraySARTTrajectory(10 * A, 2, 1, A, torch.device("cpu"))
    for ep in range(2):
        if len(tr) > 0 and tr.n_sa > tr.n_rt:
            tr.pop_sa(A)                                        # PRE_EPISODE
        for t in range(3):
            s = torch.full((A, 2), float(10 * ep + t)) + torch.arange(A)[:, None] * 0.1
            tr.push_sa(s, torch.full((A, 1), float(t)))         # PRE_ACT
            tr.push_rt(torch.full((A,), -float(t)), torch.full((A,), float(t == 2)))   # POST_ACT
        tr.push_sa(torch.full((A, 2), 99.0), torch.zeros((A, 1)))   # POST_EPISODE dummy
    assert len(tr) == 6 * A and tr.n_sa == 6 * A + A
    b = tr.sample(np.random.default_rng(0), 64)
    # next_state belongs to the same actuator one env step later (or the next episode's start, masked by terminal)
    same_ep = b["terminal"] == 0
    assert torch.allclose(b["next_state"][same_ep][:, 0], b["state"][same_ep][:, 0] + 1.0)
    assert torch.allclose(b["reward"], -b["action"][:, 0])
    # wrap-around keeps the four traces aligned
    tr2 = pkg.CircularArraySARTTrajectory(2 * A, 1, 1, A, torch.device("cpu"))
    for t in range(7):
        tr2.push_sa(torch.full((A, 1), float(t)), torch.full((A, 1), float(t)))
        tr2.push_rt(torch.full((A,), float(t)), torch.zeros(A))
    tr2.push_sa(torch.full((A, 1), 7.0), torch.zeros((A, 1)))
    b = tr2.sample(np.random.default_rng(1), 32)
    assert torch.equal(b["state"][:, 0], b["reward"]) and torch.equal(b["next_state"][:, 0], b["reward"] + 1)
    assert set(b["reward"].tolist()) <= {5.0, 6.0} - {6.0} | {5.0}


def test_stop_conditions_and_sharding(pkg):
    class E:
        def __init__(self): self.t = False
        def is_terminated(self): return self.t
    e = E()
    sc = pkg.StopAfterEpisodeWithMinSteps(3)
    assert [sc(None, e) for _ in range(2)] == [False, False]
    e.t = True
    assert sc(None, e) is True                       # first episode end at/after step 3 (StopCondition.jl:31)
    se = pkg.StopAfterEpisode(2)
    assert se(None, e) is False and se(None, e) is True
    tot = 0
    for r in range(8):
        lo, hi = pkg.distributed.shard_range(4096, 8, r)
        assert hi - lo == 512 and lo == tot
        tot = hi
    assert [pkg.distributed.shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


_WORKER = r'''
import os, sys, importlib
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4])
dist.init_process_group("gloo", rank=rank, world_size=world)
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
from oracle import nn
rng = np.random.default_rng(5)
ns, na, Bu = 3, 1, 64
da, aa = nn.layer_sizes(ns, na, 1.6, True, False); dc, ac = nn.layer_sizes(ns, na, 7.0, False, False)
mk = lambda d: nn.glorot_uniform(rng, d, np.float64)
A, C, At, Ct = mk(da), mk(dc), mk(da), mk(dc)
s, sn = rng.standard_normal((ns, Bu)), rng.standard_normal((ns, Bu))
a, r, t = rng.uniform(-1, 1, (na, Bu)), -rng.uniform(0, 1, Bu), np.zeros(Bu)
full = nn.ddpg_losses_and_grads(A, C, At, Ct, aa, ac, s, a, r, t, sn, 0.99, quirk=False)["gC"]
lo, hi = pkg.distributed.shard_range(Bu, world, rank)
# data-parallel: each rank differentiates the mean over ITS shard, scaled by 1/world (equal shards)
loc = nn.ddpg_losses_and_grads(A, C, At, Ct, aa, ac, s[:, lo:hi], a[:, lo:hi], r[lo:hi], t[lo:hi], sn[:, lo:hi], 0.99, quirk=False)["gC"]
red = pkg.distributed.all_reduce_host_grads([g / world for g in loc])
err = max(np.abs(x - y).max() for x, y in zip(red, full))
# the policy gradient (the north-star's only exchange): same identity for the actor
gA_full = nn.actor_grads(A, C, aa, ac, s)["gA"]
gA_loc = nn.actor_grads(A, C, aa, ac, s[:, lo:hi])["gA"]
redA = pkg.distributed.all_reduce_host_grads([g / world for g in gA_loc])
err = max(err, max(np.abs(x - y).max() for x, y in zip(redA, gA_full)))
# identical ADAM step on every rank keeps the replicas bit-identical
opt = nn.Adam([c.copy() for c in C], 1e-3); Cn = opt.step([c.copy() for c in C], red)
chk = torch.tensor([float(sum(np.sum(c) for c in Cn))], dtype=torch.float64)
gathered = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(gathered, chk)
same = all(torch.equal(g, gathered[0]) for g in gathered)
print(f"RESULT {err:.3e} {int(same)}")
dist.destroy_process_group()
'''


def test_world2_gradient_allreduce_equals_single_rank(tmp_path):
    """N=2 over gloo: sharded batch + summed gradients == single-rank gradient of the full batch"""
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
        m = re.search(r"RESULT (\S+) (\d)", o)
        assert m and float(m.group(1)) <= 1e-12 and m.group(2) == "1", o


def test_fluid_setup_box_tables_reproduce_dense_kernels(pkg):
    """the BW x BH boxes handed to pdec_fluid_env_create are exactly the reference's thresholded bumps
    (scripts/Fluid/setup/FluidSetup.jl:139-161), and the product setup agrees with the oracle's"""
    import numpy as np
    from oracle import fluid
    n, spa = 32, 4
    setup = pkg.FluidSetup(nx=n, sensors_per_axis=spa, variance=0.08)
    cfg = fluid.FluidConfig(nx=n, sensors_per_axis=spa, variance=0.08)
    assert np.abs(setup.gaussians - cfg.gaussians).max() == 0.0
    assert np.abs(setup.gaussians_actuators - cfg.gaussians_actuators).max() == 0.0
    sb, so, ab, ao, BH, BW, a2s = setup.box_tables()
    for boxes, org, dense in ((sb, so, setup.gaussians), (ab, ao, setup.gaussians_actuators)):
        for s in range(spa * spa):
            rebuilt = np.zeros((n, n))
            for dj in range(BW):
                for di in range(BH):
                    rebuilt[(org[s, 1] + di) % n, (org[s, 0] + dj) % n] += boxes[s, dj, di]
            assert np.abs(rebuilt - dense[s]).max() == 0.0
    assert setup.oversampling == int(np.floor(16 * n * 0.02)) and setup.state_shape == (9, 16)
    c = setup.env_cfg(2, 1)
    assert (c.ifpad, c.sensors_per_axis, c.N, c.S, c.A) == (1, spa, n, 16, 16)


def test_read_hook_matches_committed_golden(pkg):
    """checkpoint.read_hook on the reference's own saves/hook.jld2 (row F3) reproduces the committed fixture;
    runs only where the reference tree is mounted (the build container)."""
    import os
    import numpy as np
    import pytest
    ref = "/root/reference/scripts/KS/KS22/saves/hook.jld2"
    if not os.path.exists(ref):
        pytest.skip("reference tree not mounted")
    from util import load_golden
    g = load_golden("ks22_hook.npz")
    h = pkg.checkpoint.read_hook(ref)
    for a, b in zip(h["best"], (g["best_W1"], g["best_b1"], g["best_W2"], g["best_b2"])):
        assert np.array_equal(a, b)
    assert np.array_equal(h["bestDF"]["y"], g["y"]) and np.array_equal(h["bestDF"]["action"], g["action"])
    assert np.array_equal(h["rewards"], g["episode_rewards"])


def test_jld2_writer_roundtrip_and_checksums(pkg, tmp_path):
    """row F3, write side: plain arrays in a JLD2 container.  The metadata checksum (Jenkins lookup3) is pinned by the
    reference's own file -- the 44 superblock bytes of scripts/KS/KS22/saves/hook.jld2 hash to the checksum stored behind
    them -- the container carries the JLD2 text header and a version-2 superblock at base address 512 as the reference's
    files do, and every array survives a round trip through the reader that also reads the reference's artifacts."""
    import importlib
    import struct
    jl = importlib.import_module("distributedconvrl-pde-control_amd.jld2")
    sb = bytes.fromhex("894844460d0a1a0a020808000002000000000000ffffffffffffffffe1fa0200000000004cf8020000000000")
    assert jl.lookup3(sb) == 0x627460C6
    rng = np.random.default_rng(0)
    arrs = {"bestNNA_W1": rng.standard_normal((6, 1)).astype(np.float32), "bestNNA_b1": rng.standard_normal(6).astype(np.float32),
            "bestNNA_W2": rng.standard_normal((1, 6)).astype(np.float32), "trace": rng.standard_normal((3, 30000)),
            "bestNNA_dims": np.array([1, 6, 1], dtype=np.int64)}
    path = str(tmp_path / "a.jld2")
    jl.write_arrays(path, arrs)
    raw = open(path, "rb").read()
    assert raw.startswith(b"HDF5-based Julia Data Format, version 0.1.1") and raw[512:520] == b"\x89HDF\r\n\x1a\n"
    assert raw[520] == 2 and struct.unpack_from("<Q", raw, 524)[0] == 512 and struct.unpack_from("<Q", raw, 540)[0] == len(raw)
    assert jl.lookup3(raw[512:556]) == struct.unpack_from("<I", raw, 556)[0]
    f = jl.JLD2File(path)
    for off, o in f.objs.items():            # every object header carries a valid checksum
        flags = raw[off + 5]
        nsz = 1 << (flags & 3)
        chunk = int.from_bytes(raw[off + 6:off + 6 + nsz], "little")
        end = off + 6 + nsz + chunk
        assert jl.lookup3(raw[off:end]) == struct.unpack_from("<I", raw, end)[0]
    back = jl.read_arrays(path)
    assert set(back) == set(arrs)
    for k, a in arrs.items():
        assert back[k].dtype == a.dtype and np.array_equal(back[k], a), k
    ref = "/root/reference/scripts/KS/KS22/saves/hook.jld2"
    if os.path.exists(ref):                  # the reader's view of a reference dataset header == what the writer emits
        fr = jl.JLD2File(ref)
        o = next(o for o in (fr.objs[k] for k in fr.order) if o.cls == 1 and o.size == 4 and o.dims == (1, 6))
        w = jl.JLD2File(path)
        mine = next(o for o in w.objs.values() if o.dims == (1, 6))
        for (ta, pa, sa), (tb, pb, sb_) in zip(o.msgs[:3], mine.msgs[:3]):
            assert ta == tb and fr.buf[pa:pa + sa] == w.buf[pb:pb + sb_]


def test_optional_featurize_branches_are_refused_or_restated(pkg):
    """memory_size > 0 (KSSetup.jl:220-226, PDEagent.jl:201): built for the per-actuator setups of the reference (KS, Keller-Segel, fluid) since round 4 (shapes
    here, GPU parity in tests/test_gpu_memory.py), refused -- an error, not silence -- by the global agent, the 2-D Keller-Segel grid and
    together with the reward-based blow-up test; temporal_steps > 1 is accepted (the step kernels' general featurize path) and
    the oracle restates both branches of KSSetup.jl:209-218 and the memory rows of :220-226"""
    from oracle import ks
    for mk in (lambda **k: pkg.KSSetup.KS22_global(**k), lambda **k: pkg.KellerSegel2DSetup(**k), lambda **k: pkg.KSSetup.KS22(check_max_value="reward", **k),
               lambda **k: pkg.KSSetup.KS22(**{**k, "memory_size": -1})):
        with pytest.raises(pkg.PdecError, match="memory_size"):
            mk(memory_size=2)
    sm = pkg.KSSetup.KS22(window_size=3, temporal_steps=2, memory_size=2)
    assert sm.state_shape == (8, 8) and sm.action_shape == (3, 8) and sm.env_cfg(1, 0).memory_size == 2
    fm = pkg.FluidSetup(nx=32, sensors_per_axis=4, memory_size=2)
    assert fm.state_shape == (11, 16) and fm.action_shape == (3, 16) and fm.env_cfg(1, 1).memory_size == 2
    km = pkg.KellerSegelSetup(memory_size=1)
    assert km.state_shape == (13, km.n_actuators) and km.action_shape == (2, km.n_actuators) and km.env_cfg(1, 0).memory_size == 1
    cm = ks.KSConfig(192, 22.0, np.arange(1, 193, 24), sigma_sensors=0.7, sigma_actuators=0.7, window_size=3, temporal_steps=2)
    cm.memory_size = 2
    y_ = np.sin(np.arange(192) * 0.1)
    act_ = np.arange(24.0).reshape(3, 8)
    s0 = ks.featurize(cm, y_)                                   # reset form: fresh rows twice, zeros (:211-214, :222)
    assert s0.shape == (8, 8) and np.array_equal(s0[:3], s0[3:6]) and not s0[6:].any()
    s1 = ks.featurize(cm, 2 * y_, s0, act_)                     # env form: the newest block of s0 without its memory rows (:216, :224)
    assert np.array_equal(s1[3:6], s0[:3]) and np.array_equal(s1[6:], act_[1:]) and np.allclose(s1[:3], 2 * s0[:3])
    with pytest.raises(pkg.PdecError, match="temporal_steps"):
        pkg.KSSetup.KS22_global(temporal_steps=2)
    s = pkg.KSSetup.KS22(window_size=3, temporal_steps=2)
    assert s.state_shape == (6, 8) and s.env_cfg(1, 0).temporal_steps == 2
    cfg = ks.KSConfig(192, 22.0, np.arange(1, 193, 24), sigma_sensors=0.7, sigma_actuators=0.7, window_size=3, temporal_steps=2)
    y = np.sin(np.arange(192) / 7.0)
    f0 = ks.featurize(cfg, y)
    assert f0.shape == (6, 8) and np.array_equal(f0[:3], f0[3:])
    f1 = ks.featurize(cfg, 2 * y, prev_state=f0)
    assert np.array_equal(f1[3:], f0[:3]) and np.allclose(f1[:3], 2 * f0[:3])


def test_graph_capture_phase_reachability(pkg):
    """which (chunk, ring phase) pairs the graph capture can ever meet -- host logic of
    pipeline.TrainPipeline._reachable, no GPU: E = 26 / 27 / 30 with chunk 24 miss phases, E = 51 (the bench) meets all"""
    TP = pkg.TrainPipeline
    def phases(E, c):
        p = object.__new__(TP)
        p.E, p.tick, p._first_tick, p.LAG, p.ep_start = E, 8, 0, 2, 0
        return [pos for pos in range(6) if p._reachable(c, pos)]
    assert phases(51, 24) == list(range(6)) and phases(17, 6) == list(range(6))
    assert phases(26, 24) == [1, 3, 5] and phases(27, 24) == [1, 2, 4, 5] and phases(30, 24) == [1, 2, 3, 4, 5]
    assert phases(26, 6) == list(range(6)) and phases(0, 24) == list(range(6))


def test_fluid_error_detection_matches_the_restated_reference(pkg):
    """scripts/Fluid/setup/FluidSetup.jl:263-273 (-> src/PDEhook.jl:78-82): neighbouring cells of real(ifft(y)) more than 10
    apart along either axis.  FluidSetup.error_detection (torch, single field or batch) against oracle.fluid.error_detection
    on smooth fields, on a field with one 10.5 jump along each axis, and just below the threshold; make_hook hands it to
    PDEhook and the hook files the episode as errored only when it ended early."""
    import torch
    from oracle import fluid as ofl
    setup = pkg.FluidSetup.Fluid_8(nx=32)
    rng = np.random.default_rng(3)
    smooth = np.fft.fft2(rng.standard_normal((32, 32)))
    cases = []
    for axis, amp in ((0, 10.5), (1, 10.5), (0, 9.9), (1, 9.9)):
        w = np.zeros((32, 32))
        idx = [slice(None), slice(None)]
        idx[axis] = slice(5, 6)
        w[tuple(idx)] = amp
        cases.append((np.fft.fft2(w), amp > 10.0))
    cases.append((smooth, ofl.error_detection(smooth)))
    for yhat, want in cases:
        assert ofl.error_detection(yhat) == want
        assert setup.error_detection(yhat) == want
        assert setup.error_detection(torch.as_tensor(yhat)) == want
    batch = np.stack([cases[2][0], cases[0][0], cases[3][0]])          # one errored trajectory in the batch
    assert setup.error_detection(batch) and not setup.error_detection(batch[[0, 2]])
    hook = setup.make_hook(collect_NNA=False, collect_bestDF=False)
    assert hook.error_detection(cases[0][0]) and not hook.error_detection(cases[2][0])

    class _Env:                       # the fields PDEhook reads at POST_EPISODE_STAGE (src/PDEhook.jl:65-97)
        te, steps = 6.0, 3
    class _Agent:
        class policy:
            behavior_actor = None
    for t, y, errored in ((2.0, cases[0][0], True), (6.0, cases[0][0], False), (2.0, cases[2][0], False)):
        hook = setup.make_hook(collect_NNA=False, collect_bestDF=False)
        env = _Env(); env.time, env.y = t, y
        hook(pkg.POST_EPISODE_STAGE, _Agent(), env)
        assert (hook.errored_episodes == [1]) == errored, (t, hook.errored_episodes)


def test_recorded_steps_never_contain_object_teardown(pkg):
    """a recording (pipeline._eager) logs the library calls of one control step; pdec_destroy reaches the library from __del__
    of unrelated objects whenever the garbage collector runs -- it must not be logged (a replay would name a dead handle)"""
    lib = pkg._lib.load()
    calls = []
    lib.record_into(calls)
    try:
        h = pkg._lib.Handle(0)
        lib.pdec_destroy(h)                 # bad handle: returns an error code, which is fine here
        lib.pdec_version()
    finally:
        lib.record_into(None)
    names = [getattr(f, "__name__", str(f)) for f, _a in calls]
    assert not any("destroy" in n for n in names) and len(calls) == 1


_PROTO_WORKER = r'''
import importlib, os, sys, json
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world, scenario = int(sys.argv[2]), int(sys.argv[3]), sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4])
dist.init_process_group("gloo", rank=rank, world_size=world)
pkg = importlib.import_module("distributedconvrl-pde-control_amd")


class StubLib:
    """stands in for libpdeconv's communicator entry points (no GPU, no RCCL): what fails is chosen by the scenario"""
    def __init__(self):
        self.destroyed = 0
        self.err = b""
    def pdec_comm_unique_id(self, buf):
        if scenario == "id_fails":
            self.err = b"ncclGetUniqueId failed: stub"
            return -5
        return 0
    def pdec_comm_create_timeout(self, handle_ref, nranks, rk, buf, timeout_ms):
        assert nranks == world and rk == rank and timeout_ms > 0
        if scenario == "create_fails_on_1" and rank == 1:
            self.err = b"ncclCommInitRank(&C->comm, nranks, id, rank) failed: stub"
            return -5
        if scenario == "create_times_out_on_0" and rank == 0:
            self.err = b"ncclCommInitRank(nranks=2, rank=0) did not return within 1 ms (a rank missing from the rendezvous?)"
            return -5
        return 0
    def pdec_last_error(self):
        return self.err
    def pdec_destroy(self, h):
        self.destroyed += 1
        return 0


lib = StubLib()
out = {"rank": rank}
try:
    red = pkg.distributed.NativeGradReducer(lib, reduce_critic=True, timeout_s=0.001 if scenario.startswith("create_times") else 5.0)
    out["created"] = True
except pkg.PdecError as e:
    out["created"], out["error"] = False, str(e)
out["destroyed"] = lib.destroyed
# every rank is still in step with the others: one more collective completes
got = [None] * world
dist.all_gather_object(got, out)
if rank == 0:
    print("RESULT " + json.dumps(got))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("scenario", ["ok", "id_fails", "create_fails_on_1", "create_times_out_on_0"])
def test_native_communicator_bring_up_is_a_collective_protocol(tmp_path, scenario):
    """NativeGradReducer's constructor makes the same collective calls on every rank whatever fails -- rank 0
    always broadcasts (ok, id, error); every rank enters the (bounded) rendezvous; the verdicts are all-gathered -- so either
    every rank holds a communicator or every rank raised PdecError, nobody is left inside a mismatched collective, and a rank
    that did create its communicator while another did not destroys it.  World size 2 over gloo with a stub library."""
    import json
    script = tmp_path / "p.py"
    script.write_text(_PROTO_WORKER)
    port = str(31500 + (os.getpid() + hash(scenario)) % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port, scenario], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    try:
        outs = [p.communicate(timeout=120) for p in procs]
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        raise
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    res = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("RESULT ")][0][7:])
    created = [r["created"] for r in res]
    assert created == ([True, True] if scenario == "ok" else [False, False]), res
    if scenario == "id_fails":
        assert all("rank 0 could not create the ncclUniqueId" in r["error"] for r in res) and all(r["destroyed"] == 0 for r in res)
    if scenario == "create_fails_on_1":
        assert all("ncclCommInitRank" in r["error"] and "1:" in r["error"] for r in res)
        assert res[0]["destroyed"] == 1 and res[1]["destroyed"] == 0          # rank 0 had one and dropped it
    if scenario == "create_times_out_on_0":
        assert all("did not return within" in r["error"] for r in res) and res[1]["destroyed"] == 1


def test_negate_policy_is_the_fluid_scripts_baseline_controller(pkg):
    """scripts/Fluid/setup/FluidSetup.jl:277-322: result[i] = -env.state[i] over the first length(action) entries of the state in
    column-major order, clamped to +-1, behind `start_steps` steps of the start policy; POST_EPISODE resets the counter"""
    torch = pytest.importorskip("torch")

    class Env:
        pass

    B, A, ns, na = 2, 5, 9, 1
    env = Env()
    env.state = torch.arange(B * A * ns, dtype=torch.float64).reshape(B, A, ns) / 7.0 - 3.0
    env._ashape, env.dtype, env.device = (B, A, na), torch.float64, torch.device("cpu")

    class Setup:
        action_shape = (na, A)

    agent = pkg.create_agent_negate(setup=Setup(), start_steps=2)
    assert float(agent(env).abs().max()) == 0.0 and float(agent(env).abs().max()) == 0.0      # the start policy (zeros) twice
    a = agent(env)
    for b in range(B):
        st_julia = env.state[b].numpy().T                          # [ns, A]
        want = np.clip(-st_julia.reshape(-1, order="F")[:na * A], -1.0, 1.0).reshape((na, A), order="F")
        assert np.array_equal(a[b].numpy().T, want)
    agent(pkg.POST_EPISODE_STAGE, env)
    assert agent.policy.update_step == 0 and float(agent(env).abs().max()) == 0.0


def test_rlcore_wrap_shift_is_zero_until_the_traces_wrap(pkg):
    """tests/util.py rlcore_wrap_shift (study code, outside the product): index i of an RLCore CircularArrayBuffer with G pushes and
    len = min(G, capacity) frames is logical row G - len + i; state / action hold capacity + 1 frames, reward / terminal
    `capacity`, and pde_sample (src/PDEagent.jl:317-340) indexes all four with one index -> A - 1 rows apart at the PRE_ACT update
    once both have wrapped.  The product's own trajectory stays aligned (shift 0 in its slots at every fill level)."""
    import torch
    from types import SimpleNamespace
    from importlib import import_module
    from util import emulate_rlcore_wrap, rlcore_wrap_shift
    agent = import_module("distributedconvrl-pde-control_amd.agent")
    A, cap = 8, 64
    tr = agent.CircularArraySARTTrajectory(cap, 1, 1, A, torch.device("cpu"))
    plain = agent.CircularArraySARTTrajectory(cap, 1, 1, A, torch.device("cpu"))
    assert tr.capacity == cap and rlcore_wrap_shift(tr) == 0
    holder = SimpleNamespace(trajectory=tr, policy=SimpleNamespace(sampling="device"))
    emulate_rlcore_wrap(pkg, holder)
    assert holder.policy.sampling == "host" and type(tr) is not agent.CircularArraySARTTrajectory
    seen = []
    for step in range(20):
        for t in (tr, plain):
            t.push_sa(torch.full((A, 1), float(step)), torch.zeros(A, 1))          # PRE_ACT push ...
        seen.append(rlcore_wrap_shift(tr))                                         # ... update samples here
        if len(tr) > A:
            i_s, i_rt, i_sn = tr.sample_slots_many(np.random.default_rng(step), 3, 2)
            assert ((i_sn - i_s) % (cap + A) == A).all()
            j_s, j_rt, j_sn = plain.sample_slots_many(np.random.default_rng(step), 3, 2)
            assert (j_rt == i_rt).all() and ((i_s - j_s) % (cap + A) == seen[-1] % (cap + A)).all()      # same draw, shifted (s, a, s')
        for t in (tr, plain):
            t.push_rt(torch.zeros(A), torch.zeros(A))                              # POST_ACT
    # the state trace (65 frames) overflows at the 9th push (72 rows), the reward trace (64) holds 64 rows then
    assert seen == [0] * 8 + [A - 1] * 12


def test_target_network_regime_is_per_experiment_family_and_a_mismatch_warns(pkg):
    """VERDICT r5 item 5 / ADVICE r5 (medium): `quirk_frozen_targets` defaults per setup to the regime under which this path
    reproduces the reference's saved runs -- frozen for KS (src/custom_nna.jl:20 as committed: the Polyak loop of
    src/PDEagent.jl:415-417 runs over an empty list), moving for Keller-Segel and the fluid (HISTORY.md 5.1) -- and asking
    create_agent for the other one raises a TargetNetworkWarning that a caller sees.  Same table in the Julia glue."""
    import warnings
    from importlib import import_module
    agent = import_module("distributedconvrl-pde-control_amd.agent")
    want = {pkg.KSSetup.KS22(): True, pkg.KSSetup.bench_C2(256): True, pkg.KellerSegelSetup(): False,
            pkg.KellerSegel2DSetup(nx=64, ny=64): False, pkg.FluidSetup(nx=64): False}
    for setup, frozen in want.items():
        assert setup.reproduces_reference_with == ("frozen" if frozen else "moving")
        with warnings.catch_warnings():
            warnings.simplefilter("error")                           # the default and the matching explicit request are silent
            assert agent.resolve_target_networks(setup) is frozen
            assert agent.resolve_target_networks(setup, frozen) is frozen
        with pytest.warns(pkg.TargetNetworkWarning, match="reproduced only with " + ("frozen" if frozen else "moving")):
            assert agent.resolve_target_networks(setup, not frozen) is (not frozen)       # honoured, loudly
    src = open(os.path.join(ROOT, "distributedconvrl-pde-control_amd", "agent.py")).read()
    assert "resolve_target_networks(setup, overrides.get(\"quirk_frozen_targets\")" in src       # create_agent goes through it
    jl = open(os.path.join(ROOT, "julia", "PDEenvHIP.jl")).read()
    table = re.search(r"REPRODUCES_REFERENCE_WITH = Dict\((.*?)\)\n", jl, flags=re.S).group(1)
    assert dict(re.findall(r"(\w+) => :(\w+)", table)) == {"KS_CNAB2": "frozen", "KS_RK4_FD": "frozen", "KSEG_RK4": "moving",
                                                           "KSEG2D_RK4": "moving", "FLUID_RK4": "moving"}
    assert jl.count("note_experiment_family!(cfg.pde_kind)") == 3


def test_native_reducer_agrees_on_failure_through_an_exchange_callable(pkg):
    """NativeGradReducer with `exchange` (no process group) must run the agreement step too -- two threads as ranks,
    a fake library whose rendezvous fails on rank 1: BOTH constructors raise and rank 0 drops its communicator; an exchange
    that cannot gather is refused before anything collective starts"""
    import threading
    from importlib import import_module
    D = import_module("distributedconvrl-pde-control_amd.distributed")
    L = import_module("distributedconvrl-pde-control_amd._lib")

    class FakeLib:
        def __init__(self, rank, fail):
            self.rank, self.fail, self.destroyed = rank, fail, 0
        def pdec_comm_unique_id(self, buf):
            return 0
        def pdec_comm_create_timeout(self, comm, n, rank, buf, ms):
            return -1 if self.fail else 0
        def pdec_last_error(self):
            return b"rendezvous timed out (fake)"
        def pdec_destroy(self, comm):
            self.destroyed += 1
            return 0

    class Exchange:                                   # a 2-party board: broadcast from rank 0, gather in rank order
        def __init__(self):
            self.cv, self.box, self.round = threading.Condition(), {}, {}
        def make(self, rank, n=2):
            def ex(payload, op):
                with self.cv:
                    k = self.round.get(rank, 0)
                    self.round[rank] = k + 1
                    self.box.setdefault(k, {})[rank] = payload
                    self.cv.notify_all()
                    self.cv.wait_for(lambda: len(self.box[k]) == n, timeout=20)
                    got = [self.box[k][r] for r in range(n)]
                return got[0] if op == "broadcast" else got
            return ex

    for fail1 in (True, False):
        board, libs, errs, reds = Exchange(), [FakeLib(0, False), FakeLib(1, fail1)], [None, None], [None, None]
        def work(r):
            try:
                reds[r] = D.NativeGradReducer(libs[r], rank=r, world_size=2, exchange=board.make(r), timeout_s=1.0)
            except L.PdecError as e:
                errs[r] = str(e)
        th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
        [t.start() for t in th]; [t.join(30) for t in th]
        if fail1:
            assert all(e and "rank(s) 1" in e for e in errs), errs
            assert libs[0].destroyed == 1 and libs[1].destroyed == 0 and reds == [None, None]
        else:
            assert errs == [None, None] and all(r is not None and r.comm is not None for r in reds)
    with pytest.raises(L.PdecError, match="gather"):
        D.NativeGradReducer(FakeLib(0, False), rank=0, world_size=2, exchange=lambda payload: payload)
