"""Back-compat shim: the JLD2 reader now lives in the package (checkpoint I/O)."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
JLD2File = importlib.import_module("distributedconvrl-pde-control_amd.jld2").JLD2File
