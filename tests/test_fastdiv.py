"""The min/max codecs divide every residual by its channel's scale in fp16 arithmetic (reference compress_quantize.py:465, :560: fp16
tensors, i.e. the correctly rounded fp16 quotient).  The kernels' `hdiv_r` (csrc/cfx_device.h) replaces the IEEE fp32 division with
t = a * rcp(b); q = t + (a - t * b) * rcp(b) and rounds q to fp16.  This file shows on the CPU that the replacement is exact: for EVERY
pair of fp16 significands, and for a reciprocal that is off by one unit in the last place either way (v_rcp_f32's error bound), q is
within half an fp32 ulp of a / b and its fp16 rounding equals the correctly rounded quotient.  (Scaling by powers of two is exact, so the
significands cover every pair of normal operands; results below the fp16 normal range round to code 0 whatever their last bit, and zero /
infinite / NaN operands are handled by v_div_fixup_f32 - the GPU parity tests hold those against the oracle.)"""
import numpy as np
import pytest


def _fma(a, b, c):
    # products of two fp32 values fit a float64 exactly; the sum's float64 rounding error is far below the fp32 rounding that follows
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


@pytest.mark.parametrize("ulps", [-1, 0, 1])
def test_newton_step_quotient_rounds_to_the_correct_fp16(ulps):
    a = np.arange(0x3c00, 0x4400, dtype=np.uint16).view(np.float16).astype(np.float32)       # [1, 4): quotients below and above 1
    b = np.arange(0x3c00, 0x4000, dtype=np.uint16).view(np.float16).astype(np.float32)       # [1, 2)
    A, B = np.meshgrid(a, b, indexing="ij")
    exact = A.astype(np.float64) / B.astype(np.float64)
    want = exact.astype(np.float16)
    assert np.array_equal((A / B).astype(np.float16).view(np.uint16), want.view(np.uint16))   # fp32 division then fp16: no double-rounding harm
    rb = ((np.float32(1) / B).view(np.int32) + ulps).view(np.float32)
    t = A * rb
    q = _fma(_fma(-t, B, A), rb, t)
    err = np.abs(q.astype(np.float64) - exact) / np.spacing(np.abs(q)).astype(np.float64)
    assert err.max() <= 0.5 + 1e-9
    assert np.array_equal(q.astype(np.float16).view(np.uint16), want.view(np.uint16))
    if ulps:
        # (the step is needed: a * rcp(b) alone misses some roundings)
        assert not np.array_equal(t.astype(np.float16).view(np.uint16), want.view(np.uint16))
