import sys, json, ctypes as C, importlib, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
L = pkg._lib
lib = L.init(0)
setup = pkg.FluidSetup(nx=512, sensors_per_axis=16, variance=0.04)
env = pkg.PDEenv(setup, B=int(os.environ.get("B", "16")), dtype=torch.float64, device="cuda:0", autoreset=False)
y0 = setup.random_init_device(env, np.random.default_rng(0)); env.set_y0(y0)
zero = torch.zeros_like(env.y)
L.check(lib.pdec_prof_reset(env.handle)); L.check(lib.pdec_prof_enable(env.handle, 4))
for _ in range(5): env.rhs(env.y, zero)
torch.cuda.synchronize()
out = {}
for lab in ("fluid_k1", "fluid_k2", "fluid_k3"):
    ms, n = C.c_double(), C.c_int()
    L.check(lib.pdec_prof_get(env.handle, lab.encode(), C.byref(ms), C.byref(n)))
    out[lab] = round(ms.value * 1e3, 1)
print(os.environ.get("PDEC_K2P_DBG", "0"), os.environ.get("PDEC_FLUID_K2P", "-"), out)
