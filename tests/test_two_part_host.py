"""CPU: the arithmetic of the two-part contractions (csrc/gemm_x3.hip NP = 2) restated in numpy -- the
power-of-two scale from the operand's largest magnitude, x 2^e = h + l in fp16, three partial products -- against its stated
bounds.  (The kernels themselves are compared with fp64 on the GPU: tests/test_gpu_deconv.py.)"""
import numpy as np
import pytest


def exponent(maxbits):
    """x2_exponent (csrc/gemm_shared.h): e = 14 - floor(log2 max |x|) from the maximum's bit pattern, clamped to +-126."""
    e = 14 - ((int(maxbits) >> 23) - 127)
    return max(-126, min(126, e))


def split(x):
    """(e, h, l): x 2^e = h + l + err, h and l fp16 (round to nearest even, as v_cvt_pk_f16_f32)."""
    x = np.asarray(x, dtype=np.float32)
    e = exponent(np.abs(x).max().view(np.uint32))
    xs = (x * np.float32(2.0) ** e).astype(np.float32)              # exact: a power of two
    h = xs.astype(np.float16)
    l = (xs - h.astype(np.float32)).astype(np.float16)              # the remainder is exact in fp32
    return e, h, l


@pytest.mark.parametrize("scale", [1.0, 3e-30, 5e25, 1e-33])     # (a largest magnitude below 2^-112 saturates the exponent at 126: fewer bits, no overflow)
def test_split_bounds(scale):
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(200000) * rng.uniform(0, 3, 200000) * scale).astype(np.float32)
    x[:5] = [0.0, np.abs(x).max(), -np.abs(x).max(), np.float32(scale) * np.float32(2.0) ** -20, np.float32(scale) * np.float32(2.0) ** -30]
    e, h, l = split(x)
    mx = float(np.abs(x).max())
    assert np.isfinite(h.astype(np.float64)).all() and 2.0 ** 14 <= mx * 2.0 ** e < 2.0 ** 15            # nothing overflows fp16
    err = np.abs((h.astype(np.float64) + l.astype(np.float64)) * 2.0 ** -e - x.astype(np.float64))
    near = np.abs(x) >= mx * 2.0 ** -16
    assert (err[near] <= 2.0 ** -23 * np.abs(x[near])).all()        # full 22 + 1 bits while within 2^-16 of the largest value (l still normal)
    assert (err <= np.maximum(2.0 ** -23 * np.abs(x), 2.0 ** -39 * mx)).all()      # below: fp16's subnormal spacing, an absolute bound


def test_exponent_special_cases():
    assert exponent(np.float32(0).view(np.uint32)) == 126 and exponent(np.float32(1e-45).view(np.uint32)) == 126
    assert exponent(np.float32(np.inf).view(np.uint32)) == -114 and exponent(np.float32(1.0).view(np.uint32)) == 14
    assert exponent(np.float32(3.4e38).view(np.uint32)) == -113 and exponent(np.float32(65504.0).view(np.uint32)) == -1


@pytest.mark.parametrize("K", [8, 128, 5120])
def test_three_products_against_fp64(K):
    """sum_k (al wh + ah wl + ah wh) 2^-(ea + ew) against the fp64 product: per product the dropped al wl and the two
    representation errors, <= ~2^-21 |a w|; here with exact accumulation, so the bound is the whole error."""
    rng = np.random.default_rng(K)
    a = (rng.standard_normal((64, K)) * rng.uniform(0, 3, (64, 1))).astype(np.float32)
    w = (rng.standard_normal((48, K)) * 0.3).astype(np.float32)
    ea, ah, al = split(a)
    ew, wh, wl = split(w)
    f = lambda t: t.astype(np.float64)
    got = (f(al) @ f(wh).T + f(ah) @ f(wl).T + f(ah) @ f(wh).T) * 2.0 ** -(ea + ew)
    ref = f(a) @ f(w).T
    mag = np.abs(f(a)) @ np.abs(f(w)).T
    assert (np.abs(got - ref) <= 2.0 ** -21 * mag).all()
    assert np.abs(got - ref).max() / mag.max() < 1.5e-7             # in a sum the signs average: far below the fp32 accumulation's own error
