"""Pin the oracle (oracle/ref_np.py) against golden vectors captured from the reference by import
(tests/golden/make_golden.py).  CPU only.

Contract checked here (oracle/ref_np.py docstring):
  * bit / integer outputs: exact;
  * fp16 outputs given the golden scale vectors: exact;
  * the oracle's own scale vectors vs the reference's: <= 1e-3 relative, never more than 1 fp16 ulp apart
    (the reference accumulates in fp32 in an unspecified order; the oracle sums exactly).
"""
import numpy as np
import pytest

import _golden as G
from oracle import ref_np as R

F16 = np.float16
SCALE_TOL = 1e-3          # north-star tolerance for the fp error-feedback state / scales

FAST = [(64, 256), (256, 1152), (128, 3072)]
SLOW = [(64, 256), (256, 1152)]
SEEDS = [42, 43, 44]


def _scale_close(mine, gold, what):
    mine = np.ascontiguousarray(mine).view(np.uint16).reshape(-1)
    gold = np.ascontiguousarray(gold).view(np.uint16).reshape(-1)
    n, mx = G.ulp_diff_count(mine, gold)
    assert mx <= 1, f"{what}: scale differs by {mx} ulp"
    assert G.rel_err(mine, gold) < SCALE_TOL, what
    return n


@pytest.mark.parametrize("shape", FAST)
@pytest.mark.parametrize("seed", SEEDS)
def test_g1_binary_fastpath(shape, seed):
    fn = "g1_binary_fastpath_eager.npz"
    N, C = shape
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    gu, gv = G.get(fn, f"{tag}/u"), G.get(fn, f"{tag}/v")
    packed, u, v, nb = R.binary_quant_fastpath(x, base, -1, True)
    G.check(fn, f"{tag}/packed", packed, "binary packed bits")
    assert u.shape == (N, 1) and v.shape == (C, 1)
    _scale_close(u, gu, "U")
    _scale_close(v, gv, "V")
    # EF state and reconstruction given the reference's scale vectors: bit-exact
    bits01 = R.unpack_bits_1(packed)
    nb_given = R.binary_apply(R.as_f16(base), bits01, R.as_f16(gu).reshape(-1), R.as_f16(gv).reshape(-1))
    G.check(fn, f"{tag}/new_base", R.bits(nb_given), "new_base | golden scales")
    rec = R.binary_dequant_fastpath(packed, gu.reshape(-1), gv.reshape(-1), base)
    G.check(fn, f"{tag}/recon", R.bits(rec), "recon | golden scales")
    if G.stored(fn, f"{tag}/new_base"):
        assert G.rel_err(R.bits(nb), G.get(fn, f"{tag}/new_base")) < SCALE_TOL
    # update_cache=False returns no new_base
    assert R.binary_quant_fastpath(x, base, -1, False)[3] is None


@pytest.mark.parametrize("shape", FAST)
@pytest.mark.parametrize("seed", SEEDS)
def test_g2_int2_fastpath(shape, seed):
    fn = "g2_int2_fastpath_eager.npz"
    N, C = shape
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    gu, gv = G.get(fn, f"{tag}/u"), G.get(fn, f"{tag}/v")
    packed, u, v, nb = R.int2_quant_fastpath(x, base, True, -1)
    _scale_close(u, gu, "tok")
    _scale_close(v, gv, "chan")
    gp = G.get(fn, f"{tag}/packed")
    mism = float((packed != gp).mean())
    assert mism <= 1e-3, f"packed mismatch {mism}"      # the reference's own budget (compress_fastpath_test.py:133-134)
    # given the reference's scales: codes, EF state and reconstruction are bit-exact
    d = (R.as_f16(x) - R.as_f16(base)).astype(F16)
    idx, thr = R.int2_codes(d, R.as_f16(gu).reshape(-1), R.as_f16(gv).reshape(-1))
    G.check(fn, f"{tag}/packed", R.pack_int2(idx), "int2 packed | golden scales")
    nb_given = (R.as_f16(base) + R.int2_levels(idx, thr)).astype(F16)
    G.check(fn, f"{tag}/new_base", R.bits(nb_given), "new_base | golden scales")
    rec = R.int2_dequant_fastpath(gp, gu, gv, base)
    G.check(fn, f"{tag}/recon", R.bits(rec), "recon | golden scales")
    if G.stored(fn, f"{tag}/new_base"):
        assert G.rel_err(R.bits(nb), G.get(fn, f"{tag}/new_base")) < 0.02   # INT2_FASTPATH_TOL of the reference test


@pytest.mark.parametrize("shape", SLOW)
@pytest.mark.parametrize("seed", SEEDS)
def test_g3_1bit_slowpath(shape, seed):
    fn = "g3_g6_slowpath_codecs_eager.npz"
    N, C = shape
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    delta = (R.as_f16(x) - R.as_f16(base)).astype(F16)
    p, u, v = R.quantize_1bit(delta, -1)
    G.check(fn, f"{tag}/b1/packed", p, "1-bit packed")
    gu, gv = G.get(fn, f"{tag}/b1/u"), G.get(fn, f"{tag}/b1/v")
    assert u.shape == (N, 1) and v.shape == (1, C)
    nu = _scale_close(u, gu, "u")
    nv = _scale_close(v, gv, "v")
    deq = R.dequantize_1bit(p, gu, gv)
    G.check(fn, f"{tag}/b1/deq", R.bits(deq), "dequantize_1bit | golden scales")
    sim = R.sim_binary(delta, -1)
    if G.stored(fn, f"{tag}/b1/sim"):
        g = G.get(fn, f"{tag}/b1/sim")
        if nu == 0 and nv == 0:
            assert np.array_equal(R.bits(sim), g)
        assert G.rel_err(R.bits(sim), g) < SCALE_TOL
    if G.stored(fn, f"{tag}/i2mm/sim"):
        G.check(fn, f"{tag}/i2mm/sim", R.bits(R.sim_int2_minmax(delta)), "sim_int2_minmax")


@pytest.mark.parametrize("shape", SLOW)
@pytest.mark.parametrize("seed", SEEDS)
def test_g4_int8_eager_exact(shape, seed):
    fn = "g3_g6_slowpath_codecs_eager.npz"
    N, C = shape
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    delta = (R.as_f16(x) - R.as_f16(base)).astype(F16)
    q, s, zp = R.quantize_int8(delta)
    G.check(fn, f"{tag}/i8/scale", R.bits(s), "int8 scale")
    G.check(fn, f"{tag}/i8/zp", zp, "int8 zero point")
    G.check(fn, f"{tag}/i8/q", q, "int8 q")
    G.check(fn, f"{tag}/i8/deq", R.bits(R.dequantize_int8(q, s, zp)), "int8 dequant")


@pytest.mark.parametrize("shape", SLOW)
@pytest.mark.parametrize("seed", SEEDS)
def test_g5_int4_eager_exact(shape, seed):
    fn = "g3_g6_slowpath_codecs_eager.npz"
    N, C = shape
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    delta = (R.as_f16(x) - R.as_f16(base)).astype(F16)
    q, s, mn = R.quantize_int4(delta)
    G.check(fn, f"{tag}/i4/scale", R.bits(s), "int4 scale")
    G.check(fn, f"{tag}/i4/min", R.bits(mn), "int4 min")
    G.check(fn, f"{tag}/i4/q", q, "int4 packed")
    G.check(fn, f"{tag}/i4/deq", R.bits(R.dequantize_int4(q, s, mn)), "int4 dequant")
    G.check(fn, f"{tag}/i4/sim", R.bits(R.sim_int4(delta, 0)), "sim_int4")


@pytest.mark.parametrize("shape", SLOW)
@pytest.mark.parametrize("seed", SEEDS)
def test_g6_int2_slowpath(shape, seed):
    fn = "g3_g6_slowpath_codecs_eager.npz"
    N, C = shape
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    delta = (R.as_f16(x) - R.as_f16(base)).astype(F16)
    q, chan, tok = R.quantize_int2(delta)
    gc, gt = G.get(fn, f"{tag}/i2/chan"), G.get(fn, f"{tag}/i2/tok")
    assert chan.shape == (1, C) and tok.shape == (N, 1)
    _scale_close(chan, gc, "chan")
    _scale_close(tok, gt, "tok")
    idx, _ = R.int2_codes(delta, R.as_f16(gt).reshape(-1), R.as_f16(gc).reshape(-1))
    G.check(fn, f"{tag}/i2/q", R.pack_int2(idx), "int2 packed | golden scales")
    if G.stored(fn, f"{tag}/i2/q"):
        gq = G.get(fn, f"{tag}/i2/q")
        G.check(fn, f"{tag}/i2/deq", R.bits(R.dequantize_int2(gq, gc, gt)), "dequantize_int2 | golden scales")
        assert float((q != gq).mean()) <= 1e-3
        assert G.rel_err(R.bits(R.sim_int2(delta)), G.get(fn, f"{tag}/i2/sim")) < 0.02


@pytest.mark.parametrize("tag,N,C", [("64x256_s42", 64, 256), ("64x256_s43", 64, 256), ("32x1024_s42", 32, 1024)])
@pytest.mark.parametrize("m", [1, 2, 4, 8, 16])
def test_g7_topk_exact(tag, N, C, m):
    fn = "g7_topk_eager.npz"
    x, base = G.get(fn, f"{tag}/x"), G.get(fn, f"{tag}/base")
    delta = (R.as_f16(x) - R.as_f16(base)).astype(F16)
    val, idx = R.topk_compress(delta.reshape(-1, 1024), m)
    G.check(fn, f"{tag}/m{m}/val", R.bits(val), "topk val")
    G.check(fn, f"{tag}/m{m}/idx", idx, "topk idx")
    G.check(fn, f"{tag}/m{m}/dec", R.bits(R.topk_decompress(val, idx, m).reshape(N, C)), "topk decompress")
    # sim_topk uses torch.topk, whose choice among equal magnitudes is unspecified (it picked the LAST of two
    # equal |x| in block (50, 232..239) here) while the real kernel path (tl.argmax) and the oracle take the FIRST;
    # compare the simulator only on blocks without a tie for the maximum.
    sim = R.bits(R.sim_topk(delta, m)).reshape(-1, m)
    gold = G.get(fn, f"{tag}/m{m}/sim").reshape(-1, m)
    a = np.abs(delta.reshape(-1, m).astype(np.float32))
    tie = (a == a.max(axis=1, keepdims=True)).sum(axis=1) > 1
    assert np.array_equal(sim[~tie], gold[~tie])
    assert tie.mean() < 0.01


@pytest.mark.parametrize("m", [2, 4, 8])
def test_g7_topk_ties(m):
    fn = "g7_topk_eager.npz"
    x = G.get(fn, "ties/x")
    val, idx = R.topk_compress(R.as_f16(x), m)
    G.check(fn, f"ties/m{m}/val", R.bits(val), "tie val")
    G.check(fn, f"ties/m{m}/idx", idx, "tie idx")


def test_packet_roundtrip_and_sizes():
    """Wire layout arithmetic (main.py:285-293, slowpath.py:111-135) and residual codec round trips."""
    rng = np.random.default_rng(0)
    N, C = 64, 1024
    base = rng.standard_normal((N, C)).astype(F16)
    x = (base.astype(np.float32) + 0.1 * rng.standard_normal((N, C))).astype(F16)
    for codec, param in (("binary", 0), ("int2", 0), ("int4", 0), ("int8", 0), ("topk", 8)):
        pkt, nb = R.residual_compress(codec, x, base, param)
        assert pkt.dtype == np.uint16 and pkt.size == R.packet_halves(codec, N, C, param)
        rec = R.residual_decompress(codec, pkt, base, N, C, param)
        assert np.array_equal(R.bits(rec), R.bits(nb)), codec      # sender EF state == receiver reconstruction
    # 1-bit wire: 15.5x at FLUX shape
    assert R.packet_halves("binary", 544, 3072) * 2 == 216128
    assert R.packet_halves("int2", 544, 3072) == 208896 + 544 + 3072
    assert R.packet_halves("int8", 4096, 1152) * 2 == 4723200
    assert R.packet_halves("int4", 1024, 1152) * 2 == 594432


# ---- the reference AS WRITTEN runs these three quantisers under @torch.compile; inductor keeps fp32 between fused ops, so
# ---- its bits differ from its own eager run.  The eager-semantics oracle is checked against the compiled
# ---- captures only to the tolerances the reference's own tests use (compress_slowpath_test.py: INT4_TOL 0.05, INT2_TOL 0.02).
@pytest.mark.parametrize("shape", SLOW)
@pytest.mark.parametrize("seed", SEEDS)
def test_compiled_mode_captures_within_reference_tolerances(shape, seed):
    fn = "g3_g6_slowpath_codecs_compiled.npz"
    if fn not in G.manifest():
        pytest.skip("compiled-mode golden vectors not generated")
    N, C = shape
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    delta = (R.as_f16(x) - R.as_f16(base)).astype(F16)
    # scale vectors have no fused arithmetic in front of them that matters at fp16: within 1 ulp
    q8, s8, z8 = R.quantize_int8(delta)
    assert G.ulp_diff_count(R.bits(s8), G.get(fn, f"{tag}/i8/scale"))[1] <= 1
    assert int(np.abs(z8.astype(np.int32) - G.get(fn, f"{tag}/i8/zp").astype(np.int32)).max()) <= 1
    q4, s4, m4 = R.quantize_int4(delta)
    assert G.ulp_diff_count(R.bits(s4), G.get(fn, f"{tag}/i4/scale"))[1] <= 1
    assert np.array_equal(R.bits(m4), G.get(fn, f"{tag}/i4/min"))
    if G.stored(fn, f"{tag}/i8/q"):
        gq = G.get(fn, f"{tag}/i8/q").astype(np.int32)
        d = np.abs(q8.astype(np.int32) - gq)
        assert d.max() <= 2 and (d != 0).mean() < 0.10          # survey: ~6 % of int8 codes move by 1-2 between the two modes
        assert G.rel_err(R.bits(R.dequantize_int8(q8, s8, z8)), G.get(fn, f"{tag}/i8/deq")) < 0.02
        assert float((q4 != G.get(fn, f"{tag}/i4/q")).mean()) < 0.02
        assert G.rel_err(R.bits(R.dequantize_int4(q4, s4, m4)), G.get(fn, f"{tag}/i4/deq")) < 0.05
        q2, c2, t2 = R.quantize_int2(delta)
        assert float((q2 != G.get(fn, f"{tag}/i2/q")).mean()) < 5e-3
        assert G.rel_err(R.bits(R.dequantize_int2(q2, c2, t2)), G.get(fn, f"{tag}/i2/deq")) < 0.02
