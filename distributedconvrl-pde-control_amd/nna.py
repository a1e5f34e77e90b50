"""CustomNeuralNetworkApproximator / create_NNA / create_chain mirrors
(src/custom_nna.jl:7-27, src/PDEagent.jl:14-56,427-460) on top of libpdeconv's MLP handle.

`HipMLP` plays the role of the Flux `Chain(Dense...)` model: callable on a matrix whose
columns share the weights (= the convolution over actuators), parameters exposed as Flux
would (`params()` -> [W1, b1, W2, b2, ...] with W[out, in])."""
import copy
import ctypes as C

import numpy as np
import torch

from . import _lib

_ACT = {"relu": _lib.ACT_RELU, "tanh": _lib.ACT_TANH, "identity": _lib.ACT_IDENTITY, None: _lib.ACT_IDENTITY}


def glorot_uniform(rng, dims):
    """Flux.glorot_uniform(rng): U(+-sqrt(6/(in+out))) Float32 weights, zero bias
    (src/PDEagent.jl:66).  rng: numpy Generator (StableRNG streams are not reproducible
    outside Julia; only the distribution is)."""
    params = []
    for i in range(len(dims) - 1):
        lim = np.sqrt(6.0 / (dims[i] + dims[i + 1]))
        params.append(rng.uniform(-lim, lim, (dims[i + 1], dims[i])).astype(np.float32))
        params.append(np.zeros(dims[i + 1], dtype=np.float32))
    return params


class HipMLP:
    def __init__(self, dims, acts, params=None, dtype=torch.float32, device="cuda:0", max_cols=1, stream=None):
        self.dims = [int(d) for d in dims]
        self.acts = [a if isinstance(a, int) else _ACT[a] for a in acts]
        assert len(self.acts) == len(self.dims) - 1
        self.dtype, self.device = dtype, torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PdecError("HipMLP runs on the GPU only (no CPU fallback)")
        self.max_cols = int(max_cols)
        self.lib = _lib.init(self.device.index or 0)
        self._h = _lib.Handle()
        d = (C.c_int32 * len(self.dims))(*self.dims)
        a = (C.c_int32 * len(self.acts))(*self.acts)
        _lib.check(self.lib.pdec_mlp_create(C.byref(self._h), _lib.dtype_code(dtype), len(self.acts), d, a, None,
                                            self.max_cols))
        self.stream = stream
        if stream is not None:
            _lib.check(self.lib.pdec_set_stream(self._h, C.c_void_p(stream.cuda_stream)))
        if params is not None:
            self.set_params(params)

    @property
    def handle(self):
        return self._h

    @property
    def np_dtype(self):
        return np.float64 if self.dtype == torch.float64 else np.float32

    @property
    def num_params(self):
        return sum(self.dims[i] * self.dims[i + 1] + self.dims[i + 1] for i in range(len(self.acts)))

    def _flatten(self, params):
        flat = []
        for li in range(len(self.acts)):
            W, b = np.asarray(params[2 * li]), np.asarray(params[2 * li + 1])
            assert W.shape == (self.dims[li + 1], self.dims[li]), (W.shape, self.dims)
            flat.append(W.astype(self.np_dtype).ravel(order="F"))   # Julia column-major [out, in]
            flat.append(b.astype(self.np_dtype).ravel())
        return np.ascontiguousarray(np.concatenate(flat))

    def _unflatten(self, flat):
        out, o = [], 0
        for li in range(len(self.acts)):
            i, n = self.dims[li], self.dims[li + 1]
            out.append(flat[o:o + i * n].reshape((n, i), order="F").copy())
            o += i * n
            out.append(flat[o:o + n].copy())
            o += n
        return out

    def set_params(self, params):
        """Flux.loadparams!(model, params) (src/custom_nna.jl:26-27)"""
        flat = self._flatten(params)
        _lib.check(self.lib.pdec_mlp_set_params(self._h, flat.ctypes.data_as(C.c_void_p)))

    def params(self):
        """Flux.params(model) -> host arrays [W1, b1, ...]"""
        flat = np.empty(self.num_params, dtype=self.np_dtype)
        _lib.check(self.lib.pdec_mlp_get_params(self._h, flat.ctypes.data_as(C.c_void_p)))
        return self._unflatten(flat)

    def __call__(self, x):
        """x: [cols, in] device tensor (memory of Julia [in, cols]) -> [cols, out]"""
        x = x.to(self.dtype).contiguous()
        cols = x.numel() // self.dims[0]
        y = torch.empty((cols, self.dims[-1]), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.pdec_mlp_forward(self._h, _lib.ptr(x), cols, _lib.ptr(y)))
        return y

    def backward(self, x, dy, want_dx=True):
        """gradient of sum(dy .* model(x)): returns (grads [W1,b1,...] host, dx device or None)"""
        x, dy = x.to(self.dtype).contiguous(), dy.to(self.dtype).contiguous()
        cols = x.numel() // self.dims[0]
        dx = torch.empty_like(x) if want_dx else None
        g = torch.empty(self.num_params, dtype=self.dtype, device=self.device)
        _lib.check(self.lib.pdec_mlp_backward(self._h, _lib.ptr(x), _lib.ptr(dy), cols, _lib.ptr(dx), _lib.ptr(g)))
        return self._unflatten(g.cpu().numpy()), dx

    def grad_buffer(self):
        """(device pointer, n) of the flat internal gradient buffer (for the all-reduce)"""
        p, n = C.c_void_p(), C.c_int()
        _lib.check(self.lib.pdec_mlp_grad_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def clone(self, dtype=None, max_cols=None):
        m = HipMLP(self.dims, self.acts, None, dtype or self.dtype, self.device, max_cols or self.max_cols, self.stream)
        _lib.check(self.lib.pdec_mlp_copy(m._h, self._h))
        return m

    def close(self):
        if self._h is not None:
            self.lib.pdec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ADAM:
    """Flux.ADAM(eta) hyper-parameters (state lives with the model's handle)"""

    def __init__(self, eta=0.001, beta=(0.9, 0.999), epsilon=1e-8):
        self.eta, self.beta, self.epsilon = eta, beta, epsilon


class CustomNeuralNetworkApproximator:
    """src/custom_nna.jl:7-27"""

    def __init__(self, model, optimizer=None):
        self.model, self.optimizer = model, optimizer

    def __call__(self, x):                       # custom_nna.jl:13
        return self.model(x)

    def params(self):                            # @forward Flux.params
        return self.model.params()

    def device(self):
        return self.model.device

    def update(self, gs=None):
        """RLBase.update!(app, gs) = Flux.Optimise.update!(opt, params, gs) (custom_nna.jl:23-24).
        The gradients are the ones the last backward left in the model's device buffer."""
        o = self.optimizer
        _lib.check(self.model.lib.pdec_adam_step(self.model.handle, o.eta, o.beta[0], o.beta[1], o.epsilon))

    def copyto(self, src):
        """Base.copyto!(dest, src) = Flux.loadparams!(dest.model, params(src)) (custom_nna.jl:26-27)"""
        if isinstance(src, CustomNeuralNetworkApproximator):
            src = src.model
        if isinstance(src, HipMLP):
            _lib.check(self.model.lib.pdec_mlp_copy(self.model.handle, src.handle))
        else:
            self.model.set_params(src)
        return self

    def __deepcopy__(self, memo):
        return CustomNeuralNetworkApproximator(self.model.clone(), copy.copy(self.optimizer))


def layer_spec(ns, na, nna_scale, is_actor, drop_middle_layer, fun="relu"):
    """create_NNA's layer table, src/PDEagent.jl:14-44"""
    if is_actor:
        h = int(np.floor(10 * nna_scale))
        dims = [ns, h, na] if drop_middle_layer else [ns, h, h, na]
        acts = [fun, "tanh"] if drop_middle_layer else [fun, fun, "tanh"]
    else:
        h = int(np.floor(20 * nna_scale))
        dims = [ns + na, h, 1] if drop_middle_layer else [ns + na, h, h, 1]
        acts = [fun, None] if drop_middle_layer else [fun, fun, None]
    return dims, acts


def create_chain(*, na, ns, is_actor, init_rng, nna_scale, drop_middle_layer, fun="relu", dtype=torch.float32,
                 device="cuda:0", max_cols=1, stream=None):
    """src/PDEagent.jl:427-460"""
    dims, acts = layer_spec(ns, na, nna_scale, is_actor, drop_middle_layer, fun)
    return HipMLP(dims, acts, glorot_uniform(init_rng, dims), dtype, device, max_cols, stream)


def create_NNA(*, na, ns, is_actor, init_rng, nna_scale, drop_middle_layer, learning_rate=0.001, fun="relu",
               copyfrom=None, dtype=torch.float32, device="cuda:0", max_cols=1, stream=None):
    """src/PDEagent.jl:14-56"""
    model = create_chain(na=na, ns=ns, is_actor=is_actor, init_rng=init_rng, nna_scale=nna_scale,
                         drop_middle_layer=drop_middle_layer, fun=fun, dtype=dtype, device=device,
                         max_cols=max_cols, stream=stream)
    nna = CustomNeuralNetworkApproximator(model, ADAM(learning_rate))
    if copyfrom is not None:
        nna.copyto(copyfrom)
    return nna
