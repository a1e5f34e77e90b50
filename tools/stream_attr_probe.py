"""what torch's pool streams and the library's streams really are: priority range, priority and flags of each"""
import ctypes as C, importlib, sys, os
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("distributedconvrl-pde-control_amd")
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(), C.c_int()
print("range rc", hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), "least", lo.value, "greatest", hi.value)
def attrs(name, s):
    p, f = C.c_int(99), C.c_uint(99)
    r1 = hip.hipStreamGetPriority(C.c_void_p(s.cuda_stream), C.byref(p))
    r2 = hip.hipStreamGetFlags(C.c_void_p(s.cuda_stream), C.byref(f))
    print(f"{name:28s} ptr {s.cuda_stream:#x} prio {p.value} (rc {r1}) flags {f.value} (rc {r2}) torch.priority {getattr(s, 'priority', None)}")
attrs("torch Stream()", torch.cuda.Stream())
attrs("torch Stream(priority=-1)", torch.cuda.Stream(priority=-1))
attrs("torch Stream(priority=-2)", torch.cuda.Stream(priority=-2))
attrs("torch Stream(priority=1)", torch.cuda.Stream(priority=1))
for lv in (-1, 0, 1):
    attrs(f"make_stream({lv})", pkg.make_stream(lv))
print(torch.cuda.Stream.priority_range())
