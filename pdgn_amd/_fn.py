"""autograd.Function base of the package's custom Functions.

torch 2.x wraps `Function.apply` in Python: per call it unwraps dead functorch wrappers from every argument, asks whether a
functorch transform is active and inspects `setup_context` (~4 us; ~300 applies per training step, tools/host_prof.py).  Nothing
here runs under torch.func transforms or defines `setup_context`, so `apply` goes straight to the C++ implementation."""
from torch.autograd import Function as _TorchFunction


import os


class Function(_TorchFunction):
    if os.environ.get("PDGN_FAST_APPLY", "1") == "1":           # 0: torch's Python wrapper (A/B, tools/host_time.py)
        @classmethod
        def apply(cls, *args):
            return super(_TorchFunction, cls).apply(*args)
