"""The C oracle (oracle/cfx_oracle.c, used for cpu_baseline timing and big-size checks) must equal the
numpy oracle bit for bit.  CPU only."""
import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import ref_np as R

F16 = np.float16


@pytest.fixture(autouse=True, params=["generic", "f16c"])
def oracle_build(request):
    """Every test runs on both builds of the C oracle: software fp16 conversions and the F16C instructions."""
    if request.param == "f16c" and not CO.cpu_has_f16c():
        pytest.skip("CPU without f16c / avx2")
    saved = CO._lib
    CO._lib = CO.load(request.param)
    assert CO._lib.oracle_uses_f16c() == (1 if request.param == "f16c" else 0)
    yield request.param
    CO._lib = saved


CASES = [("binary", 0), ("int2", 0), ("int4", 0), ("int8", 0), ("topk", 1), ("topk", 2), ("topk", 4), ("topk", 8), ("topk", 16)]


def _inputs(seed, N, C):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((N, C)).astype(F16)
    x = (base.astype(np.float32) + 0.1 * rng.standard_normal((N, C)).astype(np.float32)).astype(F16)
    return x, base


def _same(a, b):
    a = np.asarray(a).view(np.uint16).reshape(-1)
    b = np.asarray(b).view(np.uint16).reshape(-1)
    na, nb = (a & 0x7fff) > 0x7c00, (b & 0x7fff) > 0x7c00
    return bool(((a == b) | (na & nb)).all())


def test_f16_conversion_exhaustive():
    """every fp16 bit pattern round-trips, and fp32->fp16 rounding equals numpy's for a dense sample."""
    import ctypes
    L = CO.load()
    allh = np.arange(65536, dtype=np.uint16)
    # identity codec: residual 0 'topk m=1' keeps every element: val == x
    x = allh.reshape(64, 1024)
    pkt, nb = CO.compress("topk", x, None, 64, 1024, 1)
    assert _same(pkt[:65536], allh)
    # rounding: x + base with random operands exercises f2h on sums (checked vs numpy float16 add)
    rng = np.random.default_rng(1)
    a = rng.integers(0, 65536, size=(64, 1024), dtype=np.uint16)
    b = rng.integers(0, 65536, size=(64, 1024), dtype=np.uint16)
    fin = lambda t: (t & 0x7c00) != 0x7c00
    m = fin(a) & fin(b)
    a = np.where(m, a, 0).astype(np.uint16)
    b = np.where(m, b, 0).astype(np.uint16)
    out = CO.decompress("topk", np.concatenate([a.reshape(-1), np.zeros(65536 // 4, np.uint16)]), b, 64, 1024, 1)
    with np.errstate(over="ignore", invalid="ignore"):
        want = (a.view(F16) + b.view(F16)).astype(F16)
    assert _same(out, R.bits(want))


@pytest.mark.parametrize("codec,param", CASES)
@pytest.mark.parametrize("shape", [(64, 256), (130, 1024), (256, 1152), (2, 512)])
def test_c_oracle_equals_numpy(codec, param, shape):
    N, C = shape
    if codec == "topk" and (N * C) % 1024:
        pytest.skip("needs N*C % 1024 == 0")
    x, base = _inputs(N * 7 + C, N, C)
    pkt_np, nb_np = R.residual_compress(codec, x, base, param)
    pkt_c, nb_c = CO.compress(codec, x, base, N, C, param)
    assert _same(pkt_c, pkt_np), "packet"
    assert _same(nb_c, R.bits(nb_np)), "new_base"
    rec_c = CO.decompress(codec, pkt_c, base, N, C, param)
    assert _same(rec_c, R.bits(nb_np)), "recon"
    # residual 0
    pkt_np0, _ = R.compress(codec, x, None, param)
    pkt_c0, _ = CO.compress(codec, x, None, N, C, param, update=False)
    assert _same(pkt_c0, pkt_np0)
    assert _same(CO.decompress(codec, pkt_c0, None, N, C, param), R.bits(R.decompress(codec, pkt_np0, N, C, param)))


@pytest.mark.parametrize("codec,param", CASES[:5])
def test_c_oracle_edge_cases(codec, param):
    N, C = 32, 512
    x, base = _inputs(3, N, C)
    x2 = x.copy(); b2 = base.copy()
    x2[:, 7] = b2[:, 7] + F16(0.5)
    x2[3, :] = F16(60000.0); x2[4, :] = F16(-60000.0)
    x2[5, ::2] = F16(-0.0); b2[5, :] = F16(0.0)
    x2[6, :] = F16(6e-8); b2[6, :] = F16(0.0)
    for xx, bb in ((base.copy(), base), (x2, b2)):
        pkt_np, nb_np = R.residual_compress(codec, xx, bb, param)
        pkt_c, nb_c = CO.compress(codec, xx, bb, N, C, param)
        assert _same(pkt_c, pkt_np)
        assert _same(nb_c, R.bits(nb_np))
