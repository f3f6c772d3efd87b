"""Build libpdgn_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

The library is built IN-TREE (pdgn_amd/libpdgn_hip.so) so that it travels with the
repository snapshot to the GPU box; hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libpdgn_hip.so")
ARCH = "gfx950"
# -ffp-contract=off: the distance / interpolation expressions are written with explicit
# __fmaf_rn chains that reproduce the reference's contracted arithmetic bit for bit.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]
# gemm_x3.hip: a k chunk is ONE fully unrolled straight line of up to 192 matrix instructions with the conversion, load and LDS
# work placed between them; the default budget of `#pragma unroll` (16 K IR instructions) refuses the 16x16x32 form's chunk
EXTRA_FLAGS = {"gemm_x3.hip": ["-mllvm", "-pragma-unroll-threshold=200000"],
               "gemm_x3_16.hip": ["-mllvm", "-pragma-unroll-threshold=200000"],
               "gemm_x3_h2.hip": ["-mllvm", "-pragma-unroll-threshold=200000"]}
EXTRA_DEPS = {"gemm_x3_16.hip": ["gemm_x3.hip"], "gemm_x3_h2.hip": ["gemm_x3.hip"]}          # sources a translation unit #includes besides the headers


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "pdgn_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not (force or _stale()):
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (not force) and os.path.exists(obj) and os.path.getmtime(obj) > max(
                os.path.getmtime(src), *(os.path.getmtime(os.path.join(CSRC, d)) for d in EXTRA_DEPS.get(os.path.basename(src), [])),
                *(os.path.getmtime(os.path.join(CSRC, h)) for h in os.listdir(CSRC) if h.endswith(".h")),
                os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "pdgn_hip.h"))):
            continue
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", SO] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
