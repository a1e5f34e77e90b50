"""The library's streams (include/pdeconv.h: pdec_stream_create / pdec_stream_destroy) and the part streams an environment
takes from its caller (pdec_env_part_streams / pdec_env_set_part_streams).  The reference runs on one stream; these calls exist
because of where hardware queues land on the GPU's compute pipes ("streams and compute pipes") -- WHICH stream a
part of the batch runs on never changes a result, and that is what is checked here, bit for bit."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _release(pkg, streams):
    gc.collect()
    torch.cuda.synchronize()
    for s in streams:
        pkg.destroy_stream(s)


def test_levels_are_clamped_and_streams_are_owned(pkg):
    s = [pkg.make_stream(lv) for lv in (-7, -1, 0, 1, 9)]
    assert [x.priority for x in s] == [-1, -1, 0, 1, 1]
    assert len({x.cuda_stream for x in s}) == 5
    x = torch.zeros(1 << 16, device="cuda:0")
    torch.cuda.synchronize()
    for st in s:                                     # they are ordinary streams for torch
        with torch.cuda.stream(st):
            x.add_(1.0)
        st.synchronize()
    assert float(x.sum()) == 5.0 * (1 << 16)
    with pytest.raises(pkg.PdecError):
        pkg.destroy_stream(torch.cuda.Stream())      # not one of the library's
    with pytest.raises(pkg.PdecError):
        pkg.make_streams((0, 0, 0, 0, 0))            # five busy streams cannot sit on four pipes
    _release(pkg, s)
    with pytest.raises(pkg.PdecError):
        pkg.destroy_stream(s[0])                     # released already


def test_ks_environment_has_no_parts_and_ignores_part_streams(pkg):
    env = pkg.PDEenv(pkg.KSSetup.bench_C2(256), B=4, dtype=torch.float32)
    assert env.n_part_streams == 0
    s = pkg.make_streams((1,))
    env.set_part_streams(s)
    assert env.n_part_streams == 0
    del env
    _release(pkg, s)


def test_kseg2d_parts_on_the_callers_streams_bit_identical(pkg):
    """config C4's grid, B = 96 (1 536 tiles: three parts by default): the library's own part streams, the caller's two, the
    caller's one (two parts), none (unsplit) -- the same fields, states and rewards bit for bit after three control steps"""
    setup = pkg.KellerSegel2DSetup(nx=256, ny=256, substeps=4)
    B = 96
    rng = np.random.default_rng(2)
    y0 = np.ascontiguousarray(np.moveaxis(setup.generate_random_init(rng, B), 1, -1))
    acts = [torch.from_numpy(rng.uniform(-1, 1, (B,) + tuple(reversed(setup.action_shape))).astype(np.float32)).cuda() for _ in range(3)]

    def run(stream=None, part_streams=None, expect=None):
        env = pkg.PDEenv(setup, B=B, dtype=torch.float32, y0=y0, stream=stream, autoreset=False, part_streams=part_streams)
        assert env.n_part_streams == expect
        torch.cuda.synchronize()
        for a in acts:
            env(a)
        torch.cuda.synchronize()
        return env.y.clone(), env.state.clone(), env.reward.clone()

    ref = run(expect=2)
    s = pkg.make_streams((-1, 1, 1))
    for kw, n in ((dict(stream=s[0], part_streams=s[1:]), 2), (dict(stream=s[0], part_streams=s[1:2]), 1),
                  (dict(stream=s[0], part_streams=[]), 0), (dict(part_streams=s[1:]), 2)):
        got = run(expect=n, **kw)
        for g, r in zip(got, ref):
            assert torch.equal(g, r), (n, kw.keys())
    assert bool(torch.isfinite(ref[0]).all())
    _release(pkg, s)


def test_fluid_children_on_the_callers_stream_bit_identical(pkg):
    """fluid, 512-point padded grid (n = 256 with the 3/2 rule is below the threshold; n = 512 splits), B = 8: two child
    environments; the second one's stream from the caller -- same spectra bit for bit; too few streams are refused"""
    setup = pkg.FluidSetup(nx=512, sensors_per_axis=16, variance=0.04)
    B = 8

    def make(**kw):
        env = pkg.PDEenv(setup, B=B, dtype=torch.float64, autoreset=False, **kw)
        y0 = setup.random_init_device(env, np.random.default_rng(4))
        env.set_y0(y0)
        return env

    a = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (B,) + tuple(reversed(setup.action_shape)))).cuda()
    e0 = make()
    if e0.n_part_streams == 0:
        pytest.skip("this fluid configuration runs unsplit")
    assert e0.n_part_streams == 1
    with pytest.raises(pkg.PdecError):
        e0.set_part_streams([])
    torch.cuda.synchronize()
    e0(a)
    torch.cuda.synchronize()
    s = pkg.make_streams((-1, 1))
    with torch.cuda.stream(s[0]):
        e1 = make(stream=s[0], part_streams=s[1:])
        torch.cuda.synchronize()
        e1(a)
    torch.cuda.synchronize()
    assert torch.equal(e0.y, e1.y) and torch.equal(e0.state, e1.state) and torch.equal(e0.reward, e1.reward)
    assert bool(torch.isfinite(e0.y).all())
    del e0, e1
    _release(pkg, s)


def test_part_streams_are_not_made_under_stream_capture(pkg):
    """the library's part streams / fork-join events of a split 2-D Keller-Segel step are made at the first split step;
    while the environment's stream is being captured into a HIP graph that is illegal (it would invalidate the capture), so the
    step refuses with an error -- and the same capture succeeds once one eager step has made them."""
    import ctypes as C
    L = pkg._lib
    setup = pkg.KellerSegel2DSetup(substeps=2)
    st = torch.cuda.Stream()
    B = 64                                              # 1 024 tiles: two batch parts
    lib = L.load()
    env = pkg.PDEenv(setup, B=B, dtype=torch.float32, stream=st)
    assert env.n_part_streams == 1
    y = torch.ones((B, 256, 256, 2), dtype=torch.float32, device="cuda:0")
    p = torch.zeros((B, 256, 256), dtype=torch.float32, device="cuda:0")
    out, want = torch.zeros_like(y), torch.zeros_like(y)
    flags = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    step = lambda dst: lib.pdec_pde_step(env.handle, L.ptr(y), L.ptr(p), L.ptr(dst), L.ptr(flags))
    torch.cuda.synchronize()
    L.check(lib.pdec_capture_begin(env.handle))
    try:
        with pytest.raises(pkg.PdecError, match="captured"):
            L.check(step(out))
    finally:
        h = L.Handle()
        if lib.pdec_capture_end(env.handle, C.byref(h)) == 0:
            lib.pdec_destroy(h)
    torch.cuda.synchronize()
    L.check(step(want))                                 # eager: makes the streams
    torch.cuda.synchronize()
    L.check(lib.pdec_capture_begin(env.handle))
    try:
        L.check(step(out))                              # now capturable: nothing left to create
    finally:
        h = L.Handle()
        L.check(lib.pdec_capture_end(env.handle, C.byref(h)))
    out.zero_()
    torch.cuda.synchronize()
    L.check(lib.pdec_graph_launch(h, None))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(want).all()) and torch.equal(out, want)
    lib.pdec_destroy(h)
