"""ctypes loader for libpdgn_hip.so (the C ABI declared in include/pdgn_hip.h).

There is NO CPU fallback: if the library is missing or a tensor is not on a ROCm device the
call raises.  Building is explicit (`python -m pdgn_amd.build` / `__graft_entry__.build()`).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libpdgn_hip.so")
ABI_VERSION = 26
_lib = None


class PdgnHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle of libpdgn_hip.so."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise PdgnHipError(
                "libpdgn_hip.so is not built (%s missing): run `python -m pdgn_amd.build`; "
                "pdgn_amd has no CPU fallback" % SO_PATH)
        handle = ctypes.CDLL(SO_PATH)
        handle.pdgn_abi_version.restype = ctypes.c_int
        got = handle.pdgn_abi_version()
        if got != ABI_VERSION:
            raise PdgnHipError("libpdgn_hip.so ABI %d != expected %d: rebuild" % (got, ABI_VERSION))
        _lib = handle
        if handle.pdgn_gemm_set_mode(-1) == 2:                     # the default mode: a ring on the current device now (raw ctypes callers);
            ensure_scale_slots(handle=handle)                      # the wrappers move it if the contractions run on another one
    return _lib


_SCALE_SLOTS = {}               # device index -> the ring handed to the library (one process drives one GPU; kept alive here)


def ensure_scale_slots(device=None, handle=None):
    """Two-part contractions (csrc/gemm_x3.hip): the arena of the library's own operand-maxima scans is the caller's memory -- the
    library never allocates.  64 MB on the device the contractions run on, used round-robin (a scanned operand takes 4 bytes per
    kernel-side row: at most 0.3 MB; an iteration makes a few dozen scans inside the library and at most two iterations are in
    flight); called by the contraction wrappers, so that the arena lives on the device in use whatever device was current when the
    library was loaded."""
    if not torch.cuda.is_available():
        return
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    if _SCALE_SLOTS.get("attached") == idx:
        return
    if idx not in _SCALE_SLOTS:
        _SCALE_SLOTS[idx] = torch.zeros(1 << 24, dtype=torch.int32, device=torch.device("cuda", idx))
    t = _SCALE_SLOTS[idx]
    check((handle or lib()).pdgn_gemm_set_scale_slots(ctypes.c_void_p(t.data_ptr()), ctypes.c_longlong(t.numel() * 4)),
          "pdgn_gemm_set_scale_slots")
    _SCALE_SLOTS["attached"] = idx


def stream_of(t):
    """hipStream_t of torch's current stream on t's device (raw handle: this runs ~3000 times per training step)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(t.device.index))


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def check(rc, what):
    if rc != 0:
        if rc == -1:
            raise PdgnHipError("%s: argument outside the supported range" % what)
        raise PdgnHipError("%s: HIP launch failed with hipError_t %d" % (what, rc))


def require(t, name, dtype, dim=None):
    """Reference preconditions (pointops.py:129-130, knnquery_cuda.cpp:12-14) + dtype/device."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise PdgnHipError("%s must live on a ROCm device (pdgn_amd has no CPU path)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    assert t.is_contiguous(), "%s must be contiguous" % name
    if dim is not None and t.dim() != dim:
        raise ValueError("%s must have %d dims, got %d" % (name, dim, t.dim()))
    return t


def gemm_mode():
    """'x3' (fp32 products as six bf16 MFMA products, csrc/gemm_x3.hip -- the default) or 'fp32' (the fp32 matrix instructions,
    csrc/gemm_nt.hip): what pdgn_gemm_nt / _nn / _nt_ex / _tn_big launch.  PDGN_GEMM in the environment sets the initial value."""
    return _MODE_NAMES[lib().pdgn_gemm_set_mode(-1)]


_MODE_NAMES = {0: "fp32", 1: "x3", 2: "x2"}
DEFAULT_GEMM_MODE = os.environ.get("PDGN_GEMM") or "x2"          # what the library starts in (it reads the same variable at first use)


def matrix_core_mode():
    """True in the two modes that multiply split operands on the bf16 / fp16 matrix cores (csrc/gemm_x3.hip)."""
    return gemm_mode() != "fp32"


def set_gemm_mode(mode):
    """Select the arithmetic of the dense contractions for this process: "x3" (bf16 matrix cores, three parts per value; the
    process default's matrix instruction per instance class), "x3_16" / "x3_32" (every class on v_mfma_f32_16x16x32_bf16 /
    _32x32x16_bf16), "x2" (fp16 matrix cores, two scaled parts per value), "fp32" (fp32 matrix instructions); returns the previous
    mode's name ("x3" | "x2" | "fp32").  Pre-split weights (fused.split_planes) belong to the mode they were made in."""
    m = str(mode)
    if m == "x2":
        ensure_scale_slots()
    old = lib().pdgn_gemm_set_mode(0 if m.startswith("f") else 2 if m == "x2" else 1)
    if not m.startswith("f"):                                      # (any matrix-core mode without a suffix: the process default's instruction per class)
        lib().pdgn_gemm_set_shape(16 if m.endswith("_16") else 32 if m.endswith("_32") else -1)
    return _MODE_NAMES[old]


def set_gemm_shape(shape):
    """Select the bf16 matrix instruction of the dense contractions (32: v_mfma_f32_32x32x16_bf16, 16: v_mfma_f32_16x16x32_bf16)
    for this process; returns the previous shape.  Measurement / tests (PDGN_X3_SHAPE sets the process default)."""
    return lib().pdgn_gemm_set_shape(int(shape))


def set_gemm_config(cfg):
    """Force a tile configuration of pdgn_gemm_nt / _nn (0 .. 3; None or -1: the launch model's pick).  Measurement / tests.
    Returns the previous value (-1: automatic)."""
    return lib().pdgn_gemm_set_config(-1 if cfg is None else int(cfg))
