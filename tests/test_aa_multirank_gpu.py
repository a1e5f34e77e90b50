"""N > 1 on ONE GPU (the box gpurun gives has a single MI355X): two rank processes share cuda:0, rendezvous over gloo
on 127.0.0.1, and run the REAL data-parallel code path -- HIP gradient passes into the library's flat gradient
buffers, `GradReducer.all_reduce` on those device buffers, identical ADAM / Polyak on every rank.

This file sorts first on purpose: the pytest process itself never touches the GPU here (only the rank processes it
starts do), and starting programs from a process that has already initialised the GPU is not allowed on the GPU
boxes -- so these tests must run before any other `-m gpu` test initialises it in this process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, json, importlib
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world, mode = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[2]
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
setup = pkg.KSSetup.bench_C2(256)
A_n, ns = setup.n_actuators, setup.state_shape[0]
Bg = 8 * world                                   # global batch of trajectories, 8 per rank
cols_g = Bg * A_n
g = torch.Generator().manual_seed(7)
full = dict(state=torch.randn(cols_g, ns, generator=g), action=torch.rand(cols_g, 1, generator=g) * 2 - 1,
            reward=-torch.rand(cols_g, generator=g), terminal=(torch.rand(cols_g, generator=g) < 0.05).float(),
            next_state=torch.randn(cols_g, ns, generator=g))
lo, hi = pkg.distributed.shard_range(Bg, world, rank)
shard = {k: v[lo * A_n:hi * A_n].cuda().contiguous() for k, v in full.items()}

def make(reducer, B):
    return pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), device="cuda:0", reducer=reducer,
                            quirk_target_broadcast=False, max_update_cols=cols_g)

flat = lambda nna: torch.as_tensor(np.concatenate([p.ravel() for p in nna.params()]))
if mode == "native":
    # the library's own RCCL communicator (NativeGradReducer -> pdec_comm_create / pdec_allreduce_grads), the 128-byte id
    # travelling over the gloo group.  RCCL refuses a communicator whose ranks share one device ("Duplicate GPU"), so on
    # a single-GPU box this reports the refusal (the error path of pdec_comm_create); with >= 2 devices it runs the three
    # updates and compares with the torch.distributed reducer bit for bit.
    ndev = torch.cuda.device_count()
    if ndev >= world:
        torch.cuda.set_device(rank)
    out = {"rank": rank, "mode": mode, "devices": ndev}
    try:
        nred = pkg.distributed.NativeGradReducer(pkg._lib.init(torch.cuda.current_device()), reduce_critic=True)
        out["created"] = True
    except pkg.PdecError as e:
        out["created"], out["error"] = False, str(e)
    flags = [None] * world
    dist.all_gather_object(flags, out["created"])
    if all(flags):
        dev = f"cuda:{torch.cuda.current_device()}"
        mk = lambda red: pkg.create_agent(setup=setup, B=hi - lo, rng=np.random.default_rng(1), device=dev, reducer=red,
                                          quirk_target_broadcast=False, max_update_cols=cols_g)
        sh = {k: v.to(dev) for k, v in shard.items()}
        a_nat, a_tor = mk(nred), mk(pkg.distributed.GradReducer(reduce_critic=True))
        for _ in range(3):
            a_nat.policy.update(sh); a_tor.policy.update(sh)
        torch.cuda.synchronize()
        out["native_equals_torch"] = all(torch.equal(flat(getattr(a_nat.policy, n)), flat(getattr(a_tor.policy, n)))
                                         for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"))
    dist.barrier()
    if rank == 0:
        print("RESULT " + json.dumps(out))
    dist.destroy_process_group()
    sys.exit(0)
if mode == "grads":
    # what crosses the all-reduce, against the ORACLE: each rank's fused passes on its shard (grad_scale = 1/world), summed by
    # GradReducer.all_reduce on the library's flat buffers == the fp64 oracle's gradient of the FULL batch (diagonal TD target,
    # equal shards: the mean over the global batch is the mean of the shard means); src/PDEagent.jl:385-409
    import ctypes as C
    from oracle import nn
    red = pkg.distributed.GradReducer(reduce_critic=True)
    agent = make(red, hi - lo)
    pol = agent.policy
    A, Cn, At, Ct = (pol.behavior_actor.model, pol.behavior_critic.model, pol.target_actor.model, pol.target_critic.model)
    L, P_ = pkg._lib, pkg._lib.ptr
    s, a, r, t, sn = (shard[k] for k in ("state", "action", "reward", "terminal", "next_state"))
    Bu = s.shape[0]
    losses = torch.zeros(2, device="cuda:0")
    view = lambda m: torch.as_tensor(pkg.distributed._DevArray(*m.grad_buffer(), "<f4"), device="cuda:0")
    L.check(pol.lib.pdec_ddpg_critic_grads(A.handle, Cn.handle, At.handle, Ct.handle, P_(s), P_(a), P_(r), P_(t), P_(sn), Bu, 0.99, 0,
                                           1.0 / world, C.c_void_p(losses.data_ptr())))
    red.all_reduce(Cn)
    torch.cuda.synchronize()
    gC = view(Cn).cpu().clone()
    L.check(pol.lib.pdec_ddpg_actor_grads(A.handle, Cn.handle, P_(s), Bu, 1.0 / world, C.c_void_p(losses.data_ptr() + 4)))
    red.all_reduce(A)
    torch.cuda.synchronize()
    gA = view(A).cpu().clone()
    out = {"rank": rank, "mode": mode}
    for k, v in (("gC", gC), ("gA", gA)):
        got = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(got, v)
        out[k + "_identical"] = all(torch.equal(x, got[0]) for x in got)
    if rank == 0:
        f64 = lambda m: [p.astype(np.float64) for p in m.params()]
        n64 = lambda x: x.numpy().astype(np.float64)
        acts_a, acts_c = [nn.RELU, nn.RELU, nn.TANH], [nn.RELU, nn.RELU, nn.IDENT]
        o = nn.ddpg_losses_and_grads(f64(A), f64(Cn), f64(At), f64(Ct), acts_a, acts_c, n64(full["state"]).T, n64(full["action"]).T,
                                     n64(full["reward"]), n64(full["terminal"]), n64(full["next_state"]).T,
                                     np.float64(np.float32(0.99)), False)
        o2 = nn.actor_grads(f64(A), f64(Cn), acts_a, acts_c, n64(full["state"]).T)
        def worst(flat, want):
            off, w = 0, 0.0
            for g in want:
                w = max(w, float(np.abs(flat[off:off + g.size].reshape(g.shape) - g).max() / np.abs(g).max()))
                off += g.size
            assert off == flat.size
            return w
        out["gC_vs_oracle"] = worst(gC.numpy().astype(np.float64), o["gC"])
        out["gA_vs_oracle"] = worst(gA.numpy().astype(np.float64), o2["gA"])
        # one rank's own shard alone is NOT the full-batch gradient (the all-reduce really happened)
        o_sh = nn.ddpg_losses_and_grads(f64(A), f64(Cn), f64(At), f64(Ct), acts_a, acts_c, n64(s.cpu()).T, n64(a.cpu()).T, n64(r.cpu()),
                                        n64(t.cpu()), n64(sn.cpu()).T, np.float64(np.float32(0.99)), False)
        out["shard_vs_full"] = float(max(np.abs(x - y).max() / np.abs(y).max() for x, y in zip(o_sh["gC"], o["gC"])))
    dist.barrier()
    if rank == 0:
        print("RESULT " + json.dumps(out))
    dist.destroy_process_group()
    sys.exit(0)
if mode == "pipeline":
    # VERDICT r5 item 1: the two-stream training pipeline of each rank on its shard of the trajectories, gradient exchange
    # (policy only) over gloo -- once with the all-reduce ON the update stream, once on a third stream beside the next critic
    # half (TrainPipeline(stream_ar=...)); both orders must leave the same networks and fields bit for bit on every rank, and
    # the actors must be identical replicas across the ranks.
    B = 16
    y0_all = setup.generate_random_init(np.random.default_rng(0), B * world) * 0.15
    def run(off_chain):
        s_env, s_upd, s_ar = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        env = pkg.PDEenv(setup, B=B, dtype=torch.float32, y0=y0_all[rank * B:(rank + 1) * B], stream=s_env, autoreset=False)
        agent = pkg.create_agent(setup=setup, B=B, rng=np.random.default_rng(1), dtype=torch.float32, stream=s_upd, start_steps=-1,
                                 noise_seed=7 + rank, trajectory_length=1, reducer=pkg.distributed.GradReducer(reduce_critic=False))
        agent.policy.act_noise = 0.3
        torch.cuda.synchronize()
        p = pkg.TrainPipeline(env, agent, lag=2, episode_steps=11, stream_env=s_env, stream_upd=s_upd, use_graphs=False,
                              noise_seed=99 + rank, stream_ar=s_ar, ar_off_chain=off_chain)
        assert p.multi_rank and p.ar_off_chain == off_chain
        p.run(30); p.sync()
        torch.cuda.synchronize()
        return p
    pc, ps = run(False), run(True)
    out = {"rank": rank, "mode": mode, "recorded": bool(ps._progs) and bool(pc._progs)}
    same = torch.equal(pc.y, ps.y) and bool(torch.isfinite(ps.y).all())
    for n in ("behavior_actor", "behavior_critic", "target_actor", "target_critic"):
        same = same and torch.equal(flat(getattr(pc.policy, n)), flat(getattr(ps.policy, n)))
    flags = [None] * world
    dist.all_gather_object(flags, bool(same))
    out["orders_identical_on_every_rank"] = all(flags)
    for k, v in (("actor", flat(ps.policy.behavior_actor)), ("critic", flat(ps.policy.behavior_critic))):
        got = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(got, v)
        out[k + "_identical"] = all(torch.equal(x, got[0]) for x in got)
    dist.barrier()
    if rank == 0:
        print("RESULT " + json.dumps(out))
    dist.destroy_process_group()
    sys.exit(0)
red = pkg.distributed.GradReducer(reduce_critic=(mode == "all"))
assert red.world_size == world
agent = make(red, hi - lo)
for _ in range(3):
    agent.policy.update(shard)
torch.cuda.synchronize()
pol = agent.policy
mine = {"actor": flat(pol.behavior_actor), "target_actor": flat(pol.target_actor), "critic": flat(pol.behavior_critic)}
out = {"rank": rank, "mode": mode}
for k, v in mine.items():
    got = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(got, v)
    out[k + "_identical"] = all(torch.equal(x, got[0]) for x in got)
    out[k + "_finite"] = bool(torch.isfinite(v).all())
if rank == 0 and mode == "all":
    # single-process reference: the same three updates on the FULL batch (diagonal TD target, so the mean over the
    # global batch equals the mean of the shard means)
    ref = make(None, Bg)
    fb = {k: v.cuda().contiguous() for k, v in full.items()}
    for _ in range(3):
        ref.policy.update(fb)
    torch.cuda.synchronize()
    for k, nna in (("actor", ref.policy.behavior_actor), ("critic", ref.policy.behavior_critic)):
        r = flat(nna)
        out[k + "_vs_single_rank"] = float((mine[k] - r).abs().max() / r.abs().max())
    # the all-reduce really happened: one rank's own shard alone gives a different critic
    solo = make(None, hi - lo)
    for _ in range(3):
        solo.policy.update(shard)
    torch.cuda.synchronize()
    out["critic_vs_unreduced"] = float((mine["critic"] - flat(solo.policy.behavior_critic)).abs().max())
dist.barrier()
if rank == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


def _run_ranks(tmp_path, mode, world=2):
    script = tmp_path / "rank.py"
    script.write_text(_WORKER)
    port = str(29600 + os.getpid() % 1500)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, mode], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=300 if mode == "native" else 600) for p in procs]
    except subprocess.TimeoutExpired:
        for p in procs:                      # the exact processes started above, nothing else
            p.kill()
        raise
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("RESULT ")]
    assert line, outs[0][0]
    return json.loads(line[0][7:])


def test_two_ranks_policy_gradient_only(tmp_path):
    """north-star exchange: only the ACTOR gradient is all-reduced (each rank trains its own critic on its shard):
    actors and target actors stay bit-identical replicas, the critics differ"""
    r = _run_ranks(tmp_path, "policy")
    assert r["actor_identical"] and r["target_actor_identical"] and r["actor_finite"] and r["critic_finite"]
    assert not r["critic_identical"]


def test_two_ranks_all_gradients(tmp_path):
    """SURVEY.md §8e form: actor and critic gradients all-reduced -> every network bit-identical on every rank, and
    equal (to fp32 summation-order tolerance, 2e-5 relative) to ONE rank updating on the whole batch"""
    r = _run_ranks(tmp_path, "all")
    assert r["actor_identical"] and r["target_actor_identical"] and r["critic_identical"]
    assert r["actor_vs_single_rank"] <= 2e-5 and r["critic_vs_single_rank"] <= 2e-5, r
    assert r["critic_vs_unreduced"] > 1e-6, r


def test_two_ranks_all_reduced_gradient_is_the_oracles_full_batch_gradient(tmp_path):
    """the buffer that comes out of the all-reduce (critic and actor) equals the fp64 oracle's gradient of
    the whole batch, per parameter array <= 1e-4 of its largest entry, bit-identical on both ranks"""
    r = _run_ranks(tmp_path, "grads")
    assert r["gC_identical"] and r["gA_identical"], r
    assert r["gC_vs_oracle"] <= 1e-4 and r["gA_vs_oracle"] <= 1e-4, r
    assert r["shard_vs_full"] > 1e-3, r


def test_two_ranks_allreduce_off_the_update_chain_is_bit_identical(tmp_path):
    """two ranks on the one GPU, gradient exchange over gloo, the training pipeline with the all-reduce +
    ADAM(actor) on a third stream beside the next critic half vs on the update stream: same networks and PDE state bit for bit
    on both ranks; actors identical replicas, critics local (policy-gradient-only exchange)"""
    r = _run_ranks(tmp_path, "pipeline")
    assert r["orders_identical_on_every_rank"] and r["recorded"], r
    assert r["actor_identical"] and not r["critic_identical"], r


def test_native_rccl_reducer_with_two_ranks(tmp_path):
    """the gradient exchange through the library's own RCCL communicator with nranks = 2.  On a box with
    >= 2 GPUs the two ranks run three data-parallel updates through NativeGradReducer and through the torch.distributed
    reducer: every network bit-identical.  On the single-GPU boxes of this pool RCCL itself refuses the communicator
    (two ranks on one device -- `ncclCommInitRank`: invalid usage / duplicate GPU); then the test pins exactly that: the
    refusal surfaces as a PdecError naming the RCCL call, on every rank, without a hang."""
    r = _run_ranks(tmp_path, "native")
    if r["created"]:
        assert r["devices"] >= 2 and r["native_equals_torch"], r
    else:
        assert r["devices"] < 2 and "ncclCommInitRank" in r["error"], r


def test_bench_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` (no launcher): the parent starts two fresh ranks before touching the GPU; with
    PDEC_BENCH_BACKEND=gloo they share the single device.  One JSON line, n_gpus = 2, both ranks observed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PDEC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "4",
                        "--batch", "64", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_observed"] == 2 and d["collective_backend"] == "gloo"
    assert d["config"]["global_batch"] == 128 and d["checks"]["finite"] and d["value"] > 0
    # N > 1 issues its interior steps from recorded call lists too (the all-reduce is one more recorded call) and the
    # completion events ride on the launches that apply the updates, exactly as with one rank
    assert d["issue"]["recorded_step_replay"] and d["issue"]["stop_events"] and d["collective_issued_by"] == "torch.distributed"


def test_bench_c4_two_ranks_on_one_gpu():
    """`bench.py --config C4 --gpus 2` (reduced grid 64 x 64, B = 8 per rank): the data-parallel RL step of the 2-D
    Keller-Segel config through the same self-launcher -- env + act + update per step, gradients all-reduced"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PDEC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C4", "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--batch", "8", "--nx", "64", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["n_ranks_observed"] == 2 and d["config"]["global_batch"] == 16
    assert d["checks"]["finite"] and d["value"] > 0 and d["roofline"]["kernel"] and d["unit"] == "env-steps/s"


def test_bench_c5_two_ranks_on_one_gpu():
    """`bench.py --config C5 --gpus 2` (reduced grid 128 x 128, B = 2 per rank): the data-parallel RL
    step of the 2-D fluid config (scripts/Fluid/setup/FluidSetup.jl:163-261) through the self-launcher -- fp64 env + act +
    update per step, gradients all-reduced over gloo"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PDEC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C5", "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--nx", "128", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["n_ranks_observed"] == 2 and d["config"]["global_batch"] == 4 and d["dtype"] == "f64"
    assert d["checks"]["finite"] and d["value"] > 0 and d["roofline"]["kernel"].startswith("fluid_") and d["unit"] == "env-steps/s"


def test_bench_split_update_on_one_rank():
    """`bench.py --split-update` end to end: the N > 1 launch sequence with a 1-rank native RCCL communicator (created,
    verified against the closed-form sum) in the timed pipeline; reports through the same JSON line"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--split-update", "--dp-sync", "all", "--steps", "20", "--warmup", "5",
                        "--batch", "64", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    sel = d["collective_selection"]
    assert sel["native_created"] and sel["native_verified"] and d["collective_issued_by"].startswith("libpdeconv")
    assert d["checks"]["finite"] and d["value"] > 0 and "split-update" in d["config"]["workload"] and "variants" not in d


@pytest.mark.slow
def test_self_launcher_kills_its_ranks_at_the_deadline():
    """a rank that never finishes (here: --deadline-s shorter than the import + set-up) must not hang the caller: the parent
    kills exactly its own children and exits non-zero"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PDEC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3000000", "--warmup", "4", "--batch", "64",
                        "--no-cpu-baseline", "--deadline-s", "20"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and ("did not finish within" in r.stderr or "not finished after" in r.stderr), (r.returncode, r.stderr[-2000:])


def test_bench_8gpu_line_carries_config_c3_as_a_variant():
    """`bench.py --gpus 8` appends BASELINE.json configs[2] (KS N = 1024, 512 trajectories per GPU, gradient all-reduce) as
    variants.C3, run on the same ranks with the reducer the headline verified.  Here the same code path with two ranks on the one
    GPU (PDEC_BENCH_C3_VARIANT=1 switches it on below 8 ranks), small batch: the line keeps its headline and gains the variant."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PDEC_BENCH_BACKEND="gloo", PDEC_BENCH_C3_VARIANT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "4", "--batch", "32",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and "config C2" not in d["config"]["workload"]   # custom batch
    v = d["variants"]["C3"]
    assert "error" not in v, v
    assert v["n_gpus"] == 2 and v["global_batch"] == 64 and "N=1024" in v["workload"] and v["finite"] and v["value"] > 0
    assert v["collective_issued_by"] == "torch.distributed"
