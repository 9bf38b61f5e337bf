"""Stand-alone quantisers - mirror of `xfuser/compact/compress_quantize.py` (quantize_1bit :7-90, dequantize_1bit
:154-225, quantize_int8/dequantize_int8 :428-484, quantize_int4/dequantize_int4 :522-640, quantize_int2/
dequantize_int2 :642-753, sim_binary :300-335, sim_int2 :338-384, sim_int4 :487-520), same signatures and returns.

All of them are the residual-0 (`base = NULL`) entry of the gfx950 kernels; returned parts are views into the packet.
Numerics follow the reference's EAGER semantics (fp16 rounding after every op); `@torch.compile`d reference code
differs from its own eager run and is matched to the reference's test tolerances only."""
from __future__ import annotations

import torch

from .. import codecs

K = codecs.Codec


def _nc(x):
    assert x.dtype == torch.half, "Input tensor must be FP16"
    assert x.dim() == 2, "Input tensor must be 2D"
    return x.contiguous()


def _cat_packet(*parts):
    return torch.cat([p.contiguous().view(-1).view(torch.half) if p.dtype != torch.half else p.reshape(-1) for p in parts])


# ---- 1 bit -------------------------------------------------------------------------------------------------------
def quantize_1bit(input_tensor: torch.Tensor, rank):
    """-> packed (N, C//8) uint8, scale_u (N,K), scale_v (K,C)   (rank = -1: K = 1 mean scales; rank 1..32: rank-K factors of |x|,
    compress_quantize.py:37-49 - note V is (K, C) here and (C, K) in the fastpath wire)."""
    assert rank >= 1 or rank == -1, "Rank must be >= 1 or -1"
    x = _nc(input_tensor)
    N, C = x.shape
    assert C % 8 == 0, "Channel dimension C must be divisible by 8 for packing"
    if rank != -1:
        from . import lowrank
        pkt = torch.empty(codecs.binary_rank_packet_halves(N, C, rank), dtype=torch.float16, device=x.device)
        codecs.binary_rank_compress_batch([x], [None], [None], [pkt], [lowrank._start(C, rank, x.device)], N, C, rank, update_cache=False)
        qh = N * C // 16
        return (pkt[:qh].view(torch.uint8).view(N, C // 8), pkt[qh:qh + N * rank].view(N, rank),
                pkt[qh + N * rank:].view(C, rank).t().contiguous())
    pkt, _ = codecs.compress(K.BINARY, x, None, N, C, update_cache=False)
    qh = N * C // 16
    return pkt[:qh].view(torch.uint8).view(N, C // 8), pkt[qh:qh + N].view(N, 1), pkt[qh + N:].view(1, C)


def dequantize_1bit(packed_tensor: torch.Tensor, scale_u: torch.Tensor, scale_v: torch.Tensor):
    assert packed_tensor.dtype == torch.uint8 and scale_u.dtype == torch.half and scale_v.dtype == torch.half
    N, C8 = packed_tensor.shape
    C = C8 * 8
    Kr = scale_u.shape[1]
    assert scale_u.shape == (N, Kr) and scale_v.shape == (Kr, C)
    if Kr > 1:
        out = torch.empty((N, C), dtype=torch.float16, device=packed_tensor.device)
        codecs.binary_rank_decompress_batch([_cat_packet(packed_tensor, scale_u, scale_v.t().contiguous())], [None], [out], N, C, Kr)
        return out
    return codecs.decompress(K.BINARY, _cat_packet(packed_tensor, scale_u, scale_v), None, N, C)


def sim_binary(input_tensor: torch.Tensor, rank=None):
    assert rank is not None, "Rank must be provided"
    x = _nc(input_tensor)
    return dequantize_1bit(*quantize_1bit(x, rank))


# ---- 2 bit -------------------------------------------------------------------------------------------------------
def quantize_int2(input_tensor: torch.Tensor):
    """-> packed (N, C//4) uint8, chan_scale (1,C), tok_scale (N,1)."""
    x = _nc(input_tensor)
    N, C = x.shape
    assert C % 4 == 0, f"Dimension C must be divisible by 4 for INT2 packing, got {C}"
    pkt, _ = codecs.compress(K.INT2, x, None, N, C, update_cache=False)
    qh = N * C // 8
    return pkt[:qh].view(torch.uint8).view(N, C // 4), pkt[qh + N:].view(1, C), pkt[qh:qh + N].view(N, 1)


def dequantize_int2(packed_indices: torch.Tensor, chan_scale: torch.Tensor, tok_scale: torch.Tensor):
    assert packed_indices.dtype == torch.uint8 and chan_scale.dtype == torch.half and tok_scale.dtype == torch.half
    N, C4 = packed_indices.shape
    C = C4 * 4
    assert chan_scale.shape == (1, C) and tok_scale.shape == (N, 1)
    return codecs.decompress(K.INT2, _cat_packet(packed_indices, tok_scale, chan_scale), None, N, C)


def sim_int2(input_tensor: torch.Tensor):
    return dequantize_int2(*quantize_int2(input_tensor))


def sim_int2_minmax(input_tensor: torch.Tensor):
    """4-level per-channel min/max quantise-dequantise (compress_quantize.py:386-426).  Simulation-only in the reference
    (quality studies), so this is plain tensor arithmetic on whatever device the input lives on, not a kernel."""
    x = _nc(input_tensor)
    lo = x.amin(dim=0, keepdim=True)
    step = ((x.amax(dim=0, keepdim=True) - lo) / (3 + 1e-6)).to(x.dtype)
    level = torch.clamp(torch.round((x - lo) / step), 0, 3).to(x.dtype)
    return level * step + lo


# ---- int8 --------------------------------------------------------------------------------------------------------
def quantize_int8(input_tensor: torch.Tensor):
    """-> q int8 (N,C), scale fp16 (1,C), zero_point int16 (1,C)."""
    x = _nc(input_tensor)
    N, C = x.shape
    pkt, _ = codecs.compress(K.INT8, x, None, N, C, update_cache=False)
    qh = N * C // 2
    return pkt[:qh].view(torch.int8).view(N, C), pkt[qh:qh + C].view(1, C), pkt[qh + C:].view(torch.int16).view(1, C)


def dequantize_int8(q_tensor, scale, zero_point):
    N, C = q_tensor.shape
    pkt = _cat_packet(q_tensor.view(torch.uint8), scale.reshape(-1), zero_point.reshape(-1).view(torch.half))
    return codecs.decompress(K.INT8, pkt, None, N, C)


# ---- int4 --------------------------------------------------------------------------------------------------------
def quantize_int4(input_tensor: torch.Tensor):
    """-> packed (N/2, C) uint8 [low nibble = even row], scale (1,C), min (1,C)."""
    x = _nc(input_tensor)
    N, C = x.shape
    assert N % 2 == 0, f"Dimension N (0) size must be even for INT4 packing, got {N}"
    pkt, _ = codecs.compress(K.INT4, x, None, N, C, update_cache=False)
    qh = N * C // 4
    return pkt[:qh].view(torch.uint8).view(N // 2, C), pkt[qh:qh + C].view(1, C), pkt[qh + C:].view(1, C)


def dequantize_int4(packed_tensor: torch.Tensor, scale: torch.Tensor, min_val: torch.Tensor):
    assert packed_tensor.dtype == torch.uint8 and scale.dtype == torch.half and min_val.dtype == torch.half
    N2, C = packed_tensor.shape
    return codecs.decompress(K.INT4, _cat_packet(packed_tensor, scale, min_val), None, N2 * 2, C)


def sim_int4(input_tensor: torch.Tensor, dim):
    x = _nc(input_tensor)
    if dim == 1:
        return sim_int4(x.t().contiguous(), 0).t().contiguous()
    return dequantize_int4(*quantize_int4(x))
