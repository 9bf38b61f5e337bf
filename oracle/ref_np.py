"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement, in numpy, of the arithmetic of CompactFusion's residual-compressed
activation-exchange path (`xfuser/compact/*`).  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s cpu_baseline leg may import this module.  Every function cites the
reference file:line it restates.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here against
golden vectors captured by importing the reference in the build container
(`tests/golden/make_golden.py`, groups G1-G10):
  * integer / bit outputs (packed bits, int codes, indices): bit-exact, except where they
    depend on a scale that moved by one fp16 ulp (see below);
  * fp16 outputs *given the same scale vectors*: bit-exact;
  * scale vectors themselves: within 1e-3 relative (north-star tolerance).

Numerics contract
-----------------
* fp16 arithmetic "as torch eager does it": every elementwise op takes fp16 operands,
  computes in fp32 and rounds once to fp16.  numpy's float16 ops do exactly that, and by
  Figueroa's theorem (24 >= 2*11+2) this equals a native correctly-rounded fp16 op, which
  is what the HIP kernels use (v_pk_add_f16, v_pk_mul_f16, ...).
* Reductions (`torch.mean` over rows/columns of |delta|): torch accumulates in fp32 in an
  unspecified order, so the reference's own result is order-dependent in the last fp32
  bits.  The oracle (and the HIP kernels) use the ORDER-INDEPENDENT EXACT sum instead:
  fp16 values are multiples of 2^-24, so they are summed as 64-bit integers in units of
  2^-24, converted once to fp32 (round-to-nearest-even), divided by the count in fp32 and
  rounded to fp16 - the same "sum_fp32 / N -> fp16" shape as ATen's CPU mean
  (aten/src/ATen/native/ReduceOps.cpp mean_out: cast_fp32 -> sum -> div -> cast_fp16).
  Versus any fp32 accumulation order this differs by <= ~1e-6 relative before the fp16
  rounding, i.e. the fp16 result is identical except when the value sits within that
  distance of a rounding boundary (a one-ulp, 4.9e-4 relative, flip on ~0.1 % of scale
  entries) - inside the 1e-3 tolerance and the same class of difference the reference
  shows between its own CPU and GPU runs.
* `@torch.compile`d reference functions (int8 / int4 / int2 slowpath) are restated with
  their EAGER semantics (fp16 rounding after every op).  Inductor keeps fp32 between the
  fused ops, is backend-dependent, and is not a stable specification;
  compiled-mode golden vectors are checked only to the reference's own test tolerances.
* float -> integer conversion of NaN is defined as 0 (what the GPU conversion
  instructions return; the reference relies on undefined behaviour there).
"""
from __future__ import annotations

import numpy as np

F16 = np.float16
F32 = np.float32
SPARSE_LAST_DIM_SIZE = 1024  # compress_topk.py:8


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def as_f16(a) -> np.ndarray:
    """Accept fp16 arrays or uint16 bit patterns."""
    a = np.asarray(a)
    if a.dtype == np.uint16:
        return a.view(F16)
    assert a.dtype == F16, a.dtype
    return a


def bits(a) -> np.ndarray:
    return np.ascontiguousarray(as_f16(a)).view(np.uint16)


def exact_sum_units(a16: np.ndarray, axis) -> np.ndarray:
    """Exact sum of non-negative finite fp16 values, returned as int64 counts of 2^-24."""
    fx = (a16.astype(np.float64) * 16777216.0).astype(np.int64)  # exact: < 2^40
    return fx.sum(axis=axis, dtype=np.int64)


def mean16_exact(a16: np.ndarray, axis, keepdims=False) -> np.ndarray:
    """fp16( fp32(exact_sum) / fp32(count) )  - see module docstring."""
    n = a16.shape[axis] if axis is not None else a16.size
    s = exact_sum_units(a16, axis)
    s32 = s.astype(F32) * F32(2.0 ** -24)
    m = (s32 / F32(n)).astype(F16)
    if keepdims and axis is not None:
        m = np.expand_dims(m, axis)
    return m


def _nan_to_zero_int(a, dtype):
    a = np.asarray(a, dtype=np.float32)
    out = np.where(np.isnan(a), np.float32(0), a)
    return out.astype(dtype)


def abs_delta_stats(x, base):
    """Shared scale prologue of the 1-bit and 2-bit fastpaths
    (fastpath.py:150-166 and :614-625): d = x - base, a = |d|,
    colmean = mean(a, dim=0) (C,), rowmean = mean(a, dim=1) (N,), mu = mean(rowmean)."""
    x, base = as_f16(x), as_f16(base)
    d = (x - base) if base is not None else x
    a = np.abs(d)
    colmean = mean16_exact(a, 0)
    rowmean = mean16_exact(a, 1)
    mu = mean16_exact(rowmean, 0)
    return d, colmean, rowmean, mu


# --------------------------------------------------------------------------------------
# 1-bit
# --------------------------------------------------------------------------------------
def pack_bits_1(d16: np.ndarray) -> np.ndarray:
    """bit i of byte j = (d[n, 8j+i] >= 0)   (fastpath.py:62-72, compress_quantize.py:123-145).
    NaN compares false -> bit 0."""
    N, C = d16.shape
    assert C % 8 == 0
    with np.errstate(invalid="ignore"):
        b = (d16 >= 0).astype(np.uint8).reshape(N, C // 8, 8)
    sh = np.arange(8, dtype=np.uint8)
    return (b << sh).sum(axis=2).astype(np.uint8)


def unpack_bits_1(packed: np.ndarray) -> np.ndarray:
    N, C8 = packed.shape
    sh = np.arange(8, dtype=np.uint8)
    return ((packed[:, :, None] >> sh) & 1).reshape(N, C8 * 8).astype(np.uint8)


def binary_scales_mean(d16: np.ndarray):
    """rank == -1 scales (fastpath.py:156-166 / compress_quantize.py:37-49):
    v = mean|d| over rows (C,), u = rowmean / mean(rowmean) (N,)  - no epsilon."""
    a = np.abs(d16)
    v = mean16_exact(a, 0)
    um = mean16_exact(a, 1)
    mu = mean16_exact(um, 0)
    with np.errstate(invalid="ignore", divide="ignore"):
        u = (um / mu).astype(F16)
    return u, v


def binary_apply(base16, bits01, u16, v16):
    """new_base / recon = base + (2b-1) * fp16(u[n]*v[c])   (fastpath.py:109-116, :328-363), K == 1."""
    with np.errstate(invalid="ignore", over="ignore"):
        scale = (u16.reshape(-1, 1) * v16.reshape(1, -1)).astype(F16)
        recv = np.where(bits01.astype(bool), scale, -scale).astype(F16)
        if base16 is None:
            return recv
        return (base16 + recv).astype(F16)


def binary_rank_scale(u16, vt16):
    """Rank-K scale matrix of the 1-bit codec: scale[n, c] = fp16(sum_k fp16(U[n,k] * V[c,k]))   (fastpath.py:109: tl.sum of the fp16
    products; Triton adds them in fp16 in an unspecified tree order - here fp32 in index order, one final rounding: equal for K = 1,
    within an ulp otherwise).  u16 (N, K), vt16 (C, K)."""
    u16, vt16 = as_f16(u16), as_f16(vt16)
    acc = np.zeros((u16.shape[0], vt16.shape[0]), dtype=F32)
    with np.errstate(invalid="ignore", over="ignore"):
        for k in range(u16.shape[1]):
            acc = acc + (u16[:, k:k + 1] * vt16[:, k:k + 1].T).astype(F16).astype(F32)
        return acc.astype(F16)


def binary_rank_apply(base16, bits01, u16, vt16):
    """out = base + (2b-1) * scale   (fastpath.py:109-116, :328-363 with K >= 1)."""
    with np.errstate(invalid="ignore", over="ignore"):
        scale = binary_rank_scale(u16, vt16)
        recv = np.where(bits01.astype(bool), scale, -scale).astype(F16)
        return recv if base16 is None else (as_f16(base16) + recv).astype(F16)


def binary_rank_quant_fastpath(x, base, rank, init_q, update_cache=True):
    """binary_quant_fastpath with rank >= 1 (fastpath.py:186-200): scales = subspace_iter(|x - base|, rank, 2) -> U (N,K), V (C,K).
    init_q: the (C, rank) start matrix (the reference draws torch.randn and orthonormalises it; only its span matters)."""
    x = as_f16(x)
    base = None if base is None else as_f16(base)
    d = x if base is None else (x - base).astype(F16)
    packed = pack_bits_1(d)
    U, V, _ = subspace_iter(np.abs(d), rank, 2, init_q=np.linalg.qr(np.asarray(init_q, dtype=F32))[0])
    vt = np.ascontiguousarray(V.T)
    nb = binary_rank_apply(base, unpack_bits_1(packed), U, vt) if update_cache else None
    return packed, U, vt, nb


def binary_quant_fastpath(x, base, rank=-1, update_cache=True):
    """fastpath.py:124-228 (`binary_quant_fastpath`, rank == -1 only).
    Returns packed (N, C/8) u8, u (N,1) f16, v (C,1) f16, new_base (N,C) f16 | None."""
    assert rank == -1, "rank >= 1 (subspace-iteration scales) is deprecated in the reference (main.py:188-189)"
    x, base = as_f16(x), as_f16(base)
    d = (x - base).astype(F16)
    packed = pack_bits_1(d)
    u, v = binary_scales_mean(d)
    nb = binary_apply(base, unpack_bits_1(packed), u, v) if update_cache else None
    return packed, u.reshape(-1, 1), v.reshape(-1, 1), nb


def binary_dequant_fastpath(packed, u, v, base):
    """fastpath.py:371-438."""
    return binary_apply(None if base is None else as_f16(base), unpack_bits_1(np.asarray(packed)), as_f16(u), as_f16(v))


def quantize_1bit(x, rank=-1):
    """compress_quantize.py:7-90 (rank == -1): packed (N,C/8), u (N,1), v (1,C)."""
    assert rank == -1
    x = as_f16(x)
    packed = pack_bits_1(x)
    u, v = binary_scales_mean(x)
    return packed, u.reshape(-1, 1), v.reshape(1, -1)


def dequantize_1bit(packed, u, v):
    """compress_quantize.py:154-225: S = fp16(U @ V) (K == 1: one product), out = +-S."""
    return binary_apply(None, unpack_bits_1(np.asarray(packed)), as_f16(u), as_f16(v))


def sim_binary(x, rank=-1):
    """compress_quantize.py:300-335 (rank == -1); zeros quantise to +1."""
    assert rank == -1
    x = as_f16(x)
    u, v = binary_scales_mean(x)
    with np.errstate(invalid="ignore", over="ignore"):
        scale = (v.reshape(1, -1) * u.reshape(-1, 1)).astype(F16)  # chan_scale * tok_scale
        sgn = np.sign(x).astype(F16)
        sgn = np.where(sgn == 0, F16(1), sgn)
        return (sgn * scale).astype(F16)


# --------------------------------------------------------------------------------------
# 2-bit (sign / magnitude)
# --------------------------------------------------------------------------------------
def int2_scales(d16):
    """fastpath.py:614-625 / compress_quantize.py:671-683:
    chan = mean|d| over rows (1,C); tok = rowmean / (mean(rowmean) + 1e-6) (N,1)."""
    a = np.abs(d16)
    chan = mean16_exact(a, 0)
    tm = mean16_exact(a, 1)
    mu = mean16_exact(tm, 0)
    mu_eps = (mu.astype(F32) + F32(1e-6)).astype(F16)
    with np.errstate(invalid="ignore", divide="ignore"):
        tok = (tm / mu_eps).astype(F16)
    return tok, chan


def int2_codes(d16, tok16, chan16):
    """idx = (d>=0)<<1 | (|d| > fp16(chan*tok))   (fastpath.py:536-543)."""
    with np.errstate(invalid="ignore", over="ignore"):
        thr = (chan16.reshape(1, -1) * tok16.reshape(-1, 1)).astype(F16)
        s = (d16 >= 0).astype(np.uint8)
        m = (np.abs(d16) > thr).astype(np.uint8)
    return (s << 1) | m, thr


def pack_int2(idx):
    N, C = idx.shape
    assert C % 4 == 0
    sh = (np.arange(4, dtype=np.uint8) * 2)
    return (idx.reshape(N, C // 4, 4) << sh).sum(axis=2).astype(np.uint8)


def unpack_int2(packed):
    N, C4 = packed.shape
    sh = (np.arange(4, dtype=np.uint8) * 2)
    return ((packed[:, :, None] >> sh) & 3).reshape(N, C4 * 4).astype(np.uint8)


def int2_levels(idx, thr16):
    """recv = (+-1) * (mag ? 2.0*thr : 0.5*thr)   (fastpath.py:565-573, :729-734)."""
    with np.errstate(invalid="ignore", over="ignore"):
        small = (F16(0.5) * thr16).astype(F16)
        large = (F16(2.0) * thr16).astype(F16)
        lvl = np.where((idx & 1) == 0, small, large).astype(F16)
        sgn = ((idx >> 1).astype(F16) * F16(2.0) - F16(1.0)).astype(F16)
        return (sgn * lvl).astype(F16)


def int2_quant_fastpath(x, base, update_cache=True, rank=-1):
    """fastpath.py:584-669.  Returns packed (N,C/4), u=tok (N,1), v=chan (C,1), new_base | None."""
    assert rank == -1
    x, base = as_f16(x), as_f16(base)
    d = (x - base).astype(F16)
    tok, chan = int2_scales(d)
    idx, thr = int2_codes(d, tok, chan)
    packed = pack_int2(idx)
    nb = None
    if update_cache:
        with np.errstate(invalid="ignore", over="ignore"):
            nb = (base + int2_levels(idx, thr)).astype(F16)
    return packed, tok.reshape(-1, 1), chan.reshape(-1, 1), nb


def int2_dequant_fastpath(packed, u, v, base):
    """fastpath.py:745-811."""
    tok, chan = as_f16(u).reshape(-1), as_f16(v).reshape(-1)
    idx = unpack_int2(np.asarray(packed))
    with np.errstate(invalid="ignore", over="ignore"):
        thr = (chan.reshape(1, -1) * tok.reshape(-1, 1)).astype(F16)
        recv = int2_levels(idx, thr)
        if base is None:
            return recv
        return (as_f16(base) + recv).astype(F16)


def quantize_int2(x):
    """compress_quantize.py:642-704 (eager): packed (N,C/4), chan (1,C), tok (N,1)."""
    x = as_f16(x)
    tok, chan = int2_scales(x)
    idx, _ = int2_codes(x, tok, chan)
    return pack_int2(idx), chan.reshape(1, -1), tok.reshape(-1, 1)


def dequantize_int2(packed, chan, tok):
    """compress_quantize.py:706-753 (eager)."""
    return int2_dequant_fastpath(packed, as_f16(tok).reshape(-1, 1), as_f16(chan).reshape(-1, 1), None)


def sim_int2(x):
    """compress_quantize.py:338-384 (eager): same levels, assigned with torch.where chains."""
    x = as_f16(x)
    tok, chan = int2_scales(x)
    with np.errstate(invalid="ignore", over="ignore"):
        thr = (chan.reshape(1, -1) * tok.reshape(-1, 1)).astype(F16)
        out = np.zeros_like(x)
        out = np.where(x < -thr, (F16(-2.0) * thr).astype(F16), out)
        out = np.where((x >= -thr) & (x < 0), (F16(-0.5) * thr).astype(F16), out)
        out = np.where((x >= 0) & (x <= thr), (F16(0.5) * thr).astype(F16), out)
        out = np.where(x > thr, (F16(2.0) * thr).astype(F16), out)
    return out.astype(F16)


# --------------------------------------------------------------------------------------
# min/max affine codecs (int8, int4, int2-minmax)
# --------------------------------------------------------------------------------------
def _minmax_scale(x16, levels_minus_1_plus_eps: float):
    """scale = fp16( fp16(max - min) / fp32(levels-1+1e-6) ), min, max over rows (dim 0)."""
    mn = x16.min(axis=0, keepdims=True)
    mx = x16.max(axis=0, keepdims=True)
    with np.errstate(invalid="ignore", over="ignore"):
        rng = (mx - mn).astype(F16)
        scale = (rng.astype(F32) / F32(levels_minus_1_plus_eps)).astype(F16)
    return scale, mn.astype(F16)


def quantize_int8(x):
    """compress_quantize.py:428-471 (eager semantics): q int8 (N,C), scale f16 (1,C), zp int16 (1,C)."""
    x = as_f16(x)
    scale, mn = _minmax_scale(x, 255 + 1e-6)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        r = np.rint((mn / scale).astype(F16)).astype(F16)
        zpf = (F16(-128.0) - r).astype(F16)
        zpf = np.clip(zpf, F16(-128), F16(127))           # NaN propagates
        zp = _nan_to_zero_int(zpf, np.int16)
        t = (x / scale).astype(F16)
        t = (t + zp.astype(F16)).astype(F16)
        q = np.clip(np.rint(t).astype(F16), F16(-128), F16(127))
        q = _nan_to_zero_int(q, np.int8)
    return q, scale, zp


def dequantize_int8(q, scale, zp):
    """compress_quantize.py:473-484: (q.half() - zp.half()) * scale."""
    scale = as_f16(scale).reshape(1, -1)
    with np.errstate(invalid="ignore", over="ignore"):
        t = (np.asarray(q).astype(F16) - np.asarray(zp).reshape(1, -1).astype(F16)).astype(F16)
        return (t * scale).astype(F16)


def int4_codes(x16, scale16, min16):
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        t = ((x16 - min16).astype(F16) / scale16).astype(F16)
        q = np.clip(np.rint(t).astype(F16), F16(0), F16(15))
    return _nan_to_zero_int(q, np.uint8)


def quantize_int4(x):
    """compress_quantize.py:522-583 (eager): packed (N/2, C) u8 [lo nibble = even row], scale (1,C), min (1,C)."""
    x = as_f16(x)
    N, C = x.shape
    assert N % 2 == 0
    scale, mn = _minmax_scale(x, 15 + 1e-6)
    q = int4_codes(x, scale, mn).reshape(N // 2, 2, C)
    packed = (q[:, 0, :] & 0x0F) | ((q[:, 1, :] & 0x0F) << 4)
    return packed.astype(np.uint8), scale, mn


def unpack_int4(packed):
    N2, C = packed.shape
    out = np.empty((N2 * 2, C), dtype=np.uint8)
    out[0::2] = packed & 0x0F
    out[1::2] = (packed >> 4) & 0x0F
    return out


def dequantize_int4(packed, scale, mn):
    """compress_quantize.py:585-640: q.half() * scale + min."""
    scale, mn = as_f16(scale).reshape(1, -1), as_f16(mn).reshape(1, -1)
    q = unpack_int4(np.asarray(packed))
    with np.errstate(invalid="ignore", over="ignore"):
        return ((q.astype(F16) * scale).astype(F16) + mn).astype(F16)


def sim_int4(x, dim=0):
    """compress_quantize.py:487-520."""
    x = as_f16(x)
    if dim == 1:
        return sim_int4(x.T.copy(), 0).T.copy()
    scale, mn = _minmax_scale(x, 15 + 1e-6)
    q = int4_codes(x, scale, mn)
    with np.errstate(invalid="ignore", over="ignore"):
        return ((q.astype(F16) * scale).astype(F16) + mn).astype(F16)


def sim_int2_minmax(x):
    """compress_quantize.py:386-426."""
    x = as_f16(x)
    scale, mn = _minmax_scale(x, 3 + 1e-6)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        t = ((x - mn).astype(F16) / scale).astype(F16)
        q = np.clip(np.rint(t).astype(F16), F16(0), F16(3)).astype(F16)
        return ((q * scale).astype(F16) + mn).astype(F16)


# --------------------------------------------------------------------------------------
# 1:m block top-1 ("SPARSE")
# --------------------------------------------------------------------------------------
def _first_argmax_abs(blocks16):
    """argmax |x| along the last axis, first maximum wins (tl.argmax, compress_topk.py:82-83);
    NaN magnitudes are treated as larger than everything (max-propagating)."""
    a = np.abs(blocks16).astype(np.float32)
    a = np.where(np.isnan(a), np.float32(np.inf), a)
    return np.argmax(a, axis=-1)


def topk_compress(x2d, m):
    """compress_topk.py:11-105.  x2d: (A, 1024) fp16.  Each 2m-block is two half-blocks of m;
    per half-block keep the element of largest |x|.  val (A, 1024/m) fp16, idx (A, 512/m) u8
    with idx = (i1 << 4) | i2."""
    x2d = as_f16(x2d)
    A, L = x2d.shape
    assert L % (2 * m) == 0 and m in (1, 2, 4, 8, 16)
    B = L // (2 * m)
    blk = x2d.reshape(A, B, 2, m)
    am = _first_argmax_abs(blk)                      # (A,B,2)
    val = np.take_along_axis(blk, am[..., None], axis=-1)[..., 0]  # (A,B,2)
    idx = ((am[..., 0].astype(np.uint8) << 4) | am[..., 1].astype(np.uint8)).astype(np.uint8)
    return val.reshape(A, B * 2).astype(F16), idx


def topk_decompress(val, idx, m):
    """compress_topk.py:108-163: zeros + scatter."""
    val, idx = as_f16(val), np.asarray(idx)
    A, B = idx.shape
    out = np.zeros((A, B, 2, m), dtype=F16)
    v = val.reshape(A, B, 2)
    i1 = (idx >> 4) & 0xF
    i2 = idx & 0xF
    ai, bi = np.meshgrid(np.arange(A), np.arange(B), indexing="ij")
    out[ai, bi, 0, i1] = v[..., 0]
    out[ai, bi, 1, i2] = v[..., 1]
    return out.reshape(A, B * 2 * m)


def sim_topk(x, m):
    """compress_topk.py:221-235: keep the largest-|x| element of every m (torch.topk k=1)."""
    x = as_f16(x)
    shp = x.shape
    blk = x.reshape(-1, m)
    am = _first_argmax_abs(blk)
    out = np.zeros_like(blk)
    r = np.arange(blk.shape[0])
    out[r, am] = blk[r, am]
    return out.reshape(shp)


# --------------------------------------------------------------------------------------
# low rank (subspace iteration) - fp32 linear algebra, tolerance-checked (RNG / BLAS order dependent)
# --------------------------------------------------------------------------------------
def subspace_iter(A, rank, num_iters=2, init_q=None, rng=None):
    """compress_lowrank.py:14-61.  Returns U (m,r), V (r,n), Q (n,r) in A's dtype."""
    A16 = as_f16(A)
    Af = A16.astype(F32)
    m, n = Af.shape
    if init_q is None:
        rng = rng or np.random.default_rng(0)
        Q, _ = np.linalg.qr(rng.standard_normal((n, rank)).astype(F32))
    else:
        Q = np.asarray(init_q, dtype=F32)
    for _ in range(num_iters):
        Z = Af.T @ (Af @ Q)
        Q, _ = np.linalg.qr(Z)
    U, _ = np.linalg.qr(Af @ Q)
    V = U.T @ Af
    return U.astype(F16), V.astype(F16), Q.astype(F16)


# --------------------------------------------------------------------------------------
# wire formats
# --------------------------------------------------------------------------------------
def _as_half_words(u8: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(u8).reshape(-1)
    assert b.size % 2 == 0
    return b.view(np.uint16)


def fastpath_packet(packed_u8, u16, v16) -> np.ndarray:
    """[packed bytes viewed as half | U (N*K) | V (C*K)] as uint16 words (main.py:149-152)."""
    return np.concatenate([_as_half_words(packed_u8), bits(u16).reshape(-1), bits(v16).reshape(-1)])


def fastpath_unpacket(words, N, C, per_byte):
    """main.py:285-304."""
    words = np.asarray(words).view(np.uint16).reshape(-1)
    qn = N * (C // per_byte) // 2
    assert words.size == qn + N + C, (words.size, qn, N, C)
    packed = words[:qn].view(np.uint8).reshape(N, C // per_byte)
    u = words[qn:qn + N].view(F16)
    v = words[qn + N:].view(F16)
    return packed, u, v


def packet_halves(codec: str, N: int, C: int, param: int = 0) -> int:
    """Packet length in fp16 words for every wire codec (a3 / a11 of SURVEY.md §8 and §8d)."""
    if codec == "binary":
        return N * C // 16 + N + C
    if codec == "int2":
        return N * C // 8 + N + C
    if codec == "int4":      # [q (N/2,C) u8 | scale C | min C]
        return N * C // 4 + 2 * C
    if codec == "int8":      # [q (N,C) i8 | scale C f16 | zp C i16]
        return N * C // 2 + 2 * C
    if codec == "topk":      # [val | idx]  slowpath.py:76-79,133-135
        return N * C // param + N * C // param // 4
    if codec == "lowrank":
        return (N + C) * param
    if codec == "lowrank_q":
        return (N * param) // 4 + 2 * param + (C * param) // 4 + 2 * param
    raise ValueError(codec)


# --------------------------------------------------------------------------------------
# residual codecs as *pure functions*:  (x, base) -> packet, new_base ;  (packet, base) -> recon
# These are the units the C-ABI `cfx_compress_batch` / `cfx_decompress_batch` implement.
# --------------------------------------------------------------------------------------
def _delta(x, base):
    x = as_f16(x)
    return x if base is None else (x - as_f16(base)).astype(F16)


def _add_base(base, recv):
    if base is None:
        return recv
    with np.errstate(invalid="ignore", over="ignore"):
        return (as_f16(base) + recv).astype(F16)


def compress(codec: str, x, base, param: int = 0):
    """Returns (packet_words uint16, recv fp16 (N,C)) where recv = decompress(packet) exactly,
    so new_base = base + recv is the error-feedback update (main.py:227-233)."""
    d = _delta(x, base)
    N, C = d.shape
    if codec == "binary":        # fastpath wire layout, V as (C,1)  (main.py:130-166)
        packed = pack_bits_1(d)
        u, v = binary_scales_mean(d)
        recv = binary_apply(None, unpack_bits_1(packed), u, v)
        return fastpath_packet(packed, u, v), recv
    if codec == "int2":
        tok, chan = int2_scales(d)
        idx, thr = int2_codes(d, tok, chan)
        return fastpath_packet(pack_int2(idx), tok, chan), int2_levels(idx, thr)
    if codec == "int4":          # composition of compress_quantize.py:522-640 with main.py:227-233
        q, s, mn = quantize_int4(d)
        pkt = np.concatenate([_as_half_words(q), bits(s).reshape(-1), bits(mn).reshape(-1)])
        return pkt, dequantize_int4(q, s, mn)
    if codec == "int8":
        q, s, zp = quantize_int8(d)
        pkt = np.concatenate([_as_half_words(q.view(np.uint8)), bits(s).reshape(-1), zp.reshape(-1).view(np.uint16)])
        return pkt, dequantize_int8(q, s, zp)
    if codec == "topk":          # slowpath.py:76-79
        val, idx = topk_compress(d.reshape(-1, SPARSE_LAST_DIM_SIZE), param)
        pkt = np.concatenate([bits(val).reshape(-1), _as_half_words(idx)])
        return pkt, topk_decompress(val, idx, param).reshape(N, C)
    raise ValueError(codec)


def decompress(codec: str, packet, N: int, C: int, param: int = 0):
    """packet words -> recv (N,C) fp16 (no base add)."""
    w = np.asarray(packet).view(np.uint16).reshape(-1)
    assert w.size == packet_halves(codec, N, C, param), (w.size, packet_halves(codec, N, C, param))
    if codec == "binary":
        p, u, v = fastpath_unpacket(w, N, C, 8)
        return binary_apply(None, unpack_bits_1(p), u, v)
    if codec == "int2":
        p, u, v = fastpath_unpacket(w, N, C, 4)
        return int2_dequant_fastpath(p, u, v, None)
    if codec == "int4":
        qn = N * C // 4
        q = w[:qn].view(np.uint8).reshape(N // 2, C)
        return dequantize_int4(q, w[qn:qn + C].view(F16), w[qn + C:].view(F16))
    if codec == "int8":
        qn = N * C // 2
        q = w[:qn].view(np.int8).reshape(N, C)
        return dequantize_int8(q, w[qn:qn + C].view(F16), w[qn + C:].view(np.int16))
    if codec == "topk":
        vn = N * C // param
        val = w[:vn].view(F16).reshape(-1, SPARSE_LAST_DIM_SIZE // param)
        idx = w[vn:].view(np.uint8).reshape(-1, SPARSE_LAST_DIM_SIZE // param // 2)
        return topk_decompress(val, idx, param).reshape(N, C)
    raise ValueError(codec)


def residual_compress(codec, x, base, param=0, ef=True):
    """(packet, new_base): residual-1 flow of main.py:227-233 (+ fastpath main.py:130-166)."""
    pkt, recv = compress(codec, x, base, param)
    nb = _add_base(base, recv) if ef else as_f16(x).copy()
    return pkt, nb


def residual_decompress(codec, packet, base, N, C, param=0):
    """recon = base + decompress(packet)   (main.py:373-377, :276-319)."""
    return _add_base(base, decompress(codec, packet, N, C, param))


# --------------------------------------------------------------------------------------
# slowpath BINARY wire (V stored (K,C); identical bytes for K == 1)   slowpath.py:44-53,111-150
# --------------------------------------------------------------------------------------
def slowpath_compress_binary(x):
    p, u, v = quantize_1bit(x, -1)
    return fastpath_packet(p, u, v)


def slowpath_decompress_binary(packet, N, C):
    p, u, v = fastpath_unpacket(packet, N, C, 8)
    return dequantize_1bit(p, u, v)


# --------------------------------------------------------------------------------------
# state machine (main.py:169-270, :322-388) over a dict cache (utils.py:123-162)
# --------------------------------------------------------------------------------------
def residual2_delta(x, base, dbase):
    """Second-order residual dd = (x - base) - delta_base, one fp16 rounding per operation (main.py:247)."""
    with np.errstate(invalid="ignore", over="ignore"):
        return ((np.asarray(x, F16) - np.asarray(base, F16)).astype(F16) - np.asarray(dbase, F16)).astype(F16)


def residual2_update(base, dbase, recv, decay):
    """(new_base, new_delta_base) of main.py:250-256 / :381-384: new_base = (base + delta_base) + recv and
    new_delta_base = fp16(fp32(fp16(delta_base + recv)) * fp32(decay)) - torch multiplies a half tensor by a Python scalar
    in fp32 (main.py:272-273)."""
    with np.errstate(invalid="ignore", over="ignore"):
        b, d, r = np.asarray(base, F16), np.asarray(dbase, F16), np.asarray(recv, F16)
        nb = ((b + d).astype(F16) + r).astype(F16)
        nd = ((d + r).astype(F16).astype(np.float32) * np.float32(decay)).astype(F16)
    return nb, nd


class OracleCompact:
    """Minimal restatement of compact_compress / compact_decompress with module-global state folded
    into an object.  `codec` strings: 'warmup' | 'binary' | 'int2' | 'int4' | 'int8' | 'topk'.
    `simulate=True` reproduces main.py:117-119,126-127 (packet = dequantised tensor)."""

    def __init__(self, residual=1, ef=True, fastpath=False, simulate=False, param=0, decay=None):
        assert residual in (0, 1, 2)
        self.residual, self.ef, self.fastpath, self.simulate = residual, ef, fastpath, simulate
        self.param, self.decay = param, decay
        self.base, self.dbase = {}, {}

    @staticmethod
    def _nc(x):
        x = as_f16(x)
        if x.ndim >= 4:
            return x.reshape(-1, x.shape[-2] * x.shape[-1])
        if x.ndim == 3:
            return x.reshape(x.shape[0] * x.shape[1], x.shape[2])
        assert x.ndim == 2
        return x

    def _comp(self, codec, d):
        if self.simulate:
            recv = {"int4": lambda t: sim_int4(t, 0), "int2": sim_int2, "binary": sim_binary,
                    "topk": lambda t: sim_topk(t, self.param)}[codec](d)
            return bits(recv).reshape(-1), recv
        if codec == "binary":   # slowpath wire == fastpath wire for K == 1
            pkt, recv = compress("binary", d, None)
            return pkt, recv
        return compress(codec, d, None, self.param)

    def _decomp(self, codec, pkt, N, C):
        if self.simulate:
            return np.asarray(pkt).view(np.uint16).view(F16).reshape(N, C)
        return decompress(codec, pkt, N, C, self.param)

    def compress(self, key, x, codec, update_cache=True):
        x = self._nc(x)
        N, C = x.shape
        if codec == "warmup":
            if update_cache:
                b = self.base.get(key)
                if self.residual == 2 and b is not None and not self.fastpath:
                    self.dbase[key] = (x - b).astype(F16)
                else:
                    self.dbase[key] = None
                self.base[key] = x.copy()
            return bits(x).reshape(-1)
        if self.residual == 0:
            pkt, _ = self._comp(codec, x)
            return pkt
        base = self.base[key]
        if self.residual == 1:
            d = (x - base).astype(F16)
            pkt, recv = self._comp(codec, d)
            if update_cache:
                self.base[key] = _add_base(base, recv) if self.ef else x.copy()
                self.dbase[key] = None
            return pkt
        db = self.dbase[key]
        with np.errstate(invalid="ignore", over="ignore"):
            dd = ((x - base).astype(F16) - db).astype(F16)
            pkt, recv = self._comp(codec, dd)
            if update_cache:
                self.base[key] = ((base + db).astype(F16) + recv).astype(F16)
                self.dbase[key] = residual2_update(base, db, recv, self.decay)[1]
        return pkt

    def decompress(self, key, pkt, codec, shape, update_cache=True):
        shape = tuple(shape)
        if len(shape) >= 4:
            N, C = int(np.prod(shape[:-2])), shape[-2] * shape[-1]
        elif len(shape) == 3:
            N, C = shape[0] * shape[1], shape[2]
        else:
            N, C = shape
        if codec == "warmup":
            val = np.asarray(pkt).view(np.uint16).view(F16).reshape(N, C)
            if update_cache:
                b = self.base.get(key)
                if self.residual == 2 and b is not None and not self.fastpath:
                    self.dbase[key] = (val - b).astype(F16)
                else:
                    self.dbase[key] = None
                self.base[key] = val.copy()
            return val.reshape(shape)
        recv = self._decomp(codec, pkt, N, C)
        if self.residual == 0:
            return recv.reshape(shape)
        base = self.base[key]
        if self.residual == 1:
            rec = _add_base(base, recv)
            if update_cache:
                self.base[key] = rec
                self.dbase[key] = None
            return rec.reshape(shape)
        db = self.dbase[key]
        with np.errstate(invalid="ignore", over="ignore"):
            rec = ((base + db).astype(F16) + recv).astype(F16)
            if update_cache:
                self.base[key] = rec
                self.dbase[key] = residual2_update(base, db, recv, self.decay)[1]
        return rec.reshape(shape)
