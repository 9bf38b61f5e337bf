"""Type dispatch + flat fp16 wire format of the non-fused path - mirror of `xfuser/compact/slowpath.py`
(slowpath_compress :26-84, slowpath_decompress :86-175, sim_compress :185-239).

BINARY `[codes | U(N,1) | V(1,C)]` and SPARSE `[val | idx]` are single native launches; LOW_RANK / LOW_RANK_Q are in
`lowrank.py`.  INT2 / INT4 / INT8 / IDENTITY are not slowpath wire codecs in the reference (ValueError, :80-81);
here INT2 / INT4 / INT8 are accepted as an extension because BASELINE.json's configs use them as residual codecs."""
from __future__ import annotations

import torch

from .. import codecs
from .compress_topk import SPARSE_LAST_DIM_SIZE  # noqa: F401
from .utils import COMPACT_COMPRESS_TYPE as T

_MAP = {T.BINARY: codecs.Codec.BINARY, T.INT2: codecs.Codec.INT2, T.INT4: codecs.Codec.INT4, T.INT8: codecs.Codec.INT8,
        T.SPARSE: codecs.Codec.TOPK}


def _resolve(compress_type, rank, sparse_ratio):
    if compress_type in (T.LOW_RANK, T.LOW_RANK_Q):
        assert rank is not None and rank >= 1, "Rank must be provided for LOW_RANK compression"
        return None, rank
    if compress_type not in _MAP:
        raise ValueError(f"Invalid compress_type value: {compress_type}")
    if compress_type == T.BINARY:
        assert rank is not None and (rank >= 1 or rank == -1), "Rank must be >= 1 or -1 for BINARY compression"
        if rank != -1:
            return "binary_rank", int(rank)       # rank-K scales (deprecated in the reference, kept for its tests): wire [q | U (N,K) | V (K,C)]
    if compress_type == T.SPARSE:
        assert sparse_ratio is not None, "sparse_ratio must be provided for SPARSE compression"
        return int(_MAP[compress_type]), int(sparse_ratio)
    return int(_MAP[compress_type]), 0


def slowpath_compress(x: torch.Tensor, compress_type: T, rank: int = None, sparse_ratio: int = None):
    assert x.dtype == torch.half, f"x.dtype: {x.dtype}"
    assert x.dim() == 2
    N, C = x.shape
    cid, param = _resolve(compress_type, rank, sparse_ratio)
    if cid is None:
        from . import lowrank
        return lowrank.slowpath_compress(x.contiguous(), compress_type, param)
    if cid == "binary_rank":
        # slowpath.py:44-53: quantize_1bit(x, rank) -> cat(q, U (N,K), V (K,C))
        from .compress_quantize import quantize_1bit
        q, u, v = quantize_1bit(x, param)
        return torch.cat([q.contiguous().view(-1).view(torch.half), u.reshape(-1), v.reshape(-1)])
    pkt, _ = codecs.compress(cid, x.contiguous(), None, N, C, param, update_cache=False)
    return pkt


def slowpath_decompress(x: torch.Tensor, shape: tuple, compress_type: T, rank: int = None, sparse_ratio: int = None):
    assert x.dim() == 1 and x.dtype == torch.half
    assert len(shape) == 2
    N, C = shape
    cid, param = _resolve(compress_type, rank, sparse_ratio)
    if cid is None:
        from . import lowrank
        return lowrank.slowpath_decompress(x, (N, C), compress_type, param)
    if cid == "binary_rank":
        from .compress_quantize import dequantize_1bit
        qh = N * C // 16
        assert x.numel() == qh + (N + C) * param, "packet size does not match (N, C) and the rank"
        return dequantize_1bit(x[:qh].view(torch.uint8).view(N, C // 8), x[qh:qh + N * param].view(N, param), x[qh + N * param:].view(param, C))
    assert x.numel() == codecs.packet_halves(cid, N, C, param), "packet size does not match (N, C) and the codec"
    return codecs.decompress(cid, x, None, N, C, param)


# ---- LOW_RANK_AWL ("attention-aware" low rank; deprecated in the reference: simulate mode only, behind COMPACT_ALLOW_DEPRECATED) -------
_current_lowrank_scale_k = None      # (C,) or (N,): per-channel or per-token importance, set by compact_update_awl_scale (ring.py)
_current_lowrank_scale_v = None


def set_current_lowrank_scale(scale_k, scale_v) -> None:
    """slowpath.py:177-183"""
    global _current_lowrank_scale_k, _current_lowrank_scale_v
    _current_lowrank_scale_k, _current_lowrank_scale_v = scale_k, scale_v


def _sim_lowrank_awl(x: torch.Tensor, rank: int):
    """slowpath.py:217-237: rank-r approximation of the residual WEIGHTED by the current importance scale (rows or columns), the
    weight divided out of the factors again.  The factorisation is the function-level subspace_iter (library GEMM / QR on whatever
    device x is on): this type only exists in the reference's simulate mode, there is no wire format to be fast about."""
    from .utils import ALLOW_DEPRECATED
    from .lowrank import subspace_iter
    from .main import compact_get_current_cache_key
    assert rank is not None
    assert ALLOW_DEPRECATED, "LOW_RANK_AWL is deprecated"
    N, C = x.shape
    is_k = compact_get_current_cache_key().split("-")[-1] == "k"
    scale = _current_lowrank_scale_k if is_k else _current_lowrank_scale_v
    by_col = scale is not None and tuple(scale.shape) == (C,)
    by_row = scale is not None and tuple(scale.shape) == (N,)
    if by_col:
        x = x.float() * scale.view(1, C)
    elif by_row:
        x = x.float() * scale.view(N, 1)
    u, v, _ = subspace_iter(x, rank, 2)
    if by_col:
        v = v / scale.view(1, C)
    elif by_row:
        u = u / scale.view(N, 1)
    return torch.matmul(u, v)


def sim_compress(x: torch.Tensor, compress_type: T, sparse_ratio: int = None, rank: int = None):
    """decode(encode(x)) at full size - the reference's `simulate=True` primitive."""
    if compress_type == T.IDENTITY:
        return x
    if compress_type == T.INT2_MINMAX:
        from .compress_quantize import sim_int2_minmax
        return sim_int2_minmax(x)
    if compress_type == T.LOW_RANK_AWL:
        return _sim_lowrank_awl(x, rank)
    pkt = slowpath_compress(x.half(), compress_type, rank=rank, sparse_ratio=sparse_ratio)
    return slowpath_decompress(pkt, tuple(x.shape), compress_type, rank=rank, sparse_ratio=sparse_ratio)
