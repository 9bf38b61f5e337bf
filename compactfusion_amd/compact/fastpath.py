"""Function-level API of the fused 1-bit / 2-bit paths - mirror of `xfuser/compact/fastpath.py`
(binary_quant_fastpath :124-228, binary_dequant_fastpath :371-438, int2_quant_fastpath :584-669,
int2_dequant_fastpath :745-811 and their `sim_*` twins) with identical argument order and return tuples.

Each quantiser is ONE call into libcfx.so that produces the wire packet `[codes | U | V]`; the returned `packed`,
`scale_u`, `scale_v` tensors are VIEWS into that packet (no copies), `new_base` is a fresh (N, C) tensor as in the
reference.  The dequantisers rebuild a packet from the three parts (a copy - the state-machine API in `main.py`
passes whole packets and never pays it)."""
from __future__ import annotations

import torch

from .. import codecs
from ..prof import Profiler

_BIN, _I2 = int(codecs.Codec.BINARY), int(codecs.Codec.INT2)


def _check(x, base):
    assert x.dtype == torch.half and base.dtype == torch.half
    assert x.ndim == 2 and base.ndim == 2 and x.shape == base.shape


def _quant(cid, per_byte, x, base, update_cache):
    x, base = x.contiguous(), base.contiguous()
    N, C = x.shape
    assert C % per_byte == 0, "C_COLS must be divisible by %d for packing output alignment" % per_byte
    pkt, nb = codecs.compress(cid, x, base, N, C, 0, update_cache=update_cache)
    qh = N * (C // per_byte) // 2
    packed = pkt[:qh].view(torch.uint8).view(N, C // per_byte)
    return packed, pkt[qh:qh + N].view(N, 1), pkt[qh + N:].view(C, 1), nb


def _dequant(cid, per_byte, packed, u, v, base):
    assert packed.dtype == torch.uint8 and u.dtype == torch.half and v.dtype == torch.half and base.dtype == torch.half
    N, Cp = packed.shape
    C = Cp * per_byte
    assert u.shape == (N, 1) and v.shape == (C, 1), "scale shapes must be U(N,1), V(C,1) (comp_rank = -1)"
    assert base.shape == (N, C), f"Base shape mismatch: {tuple(base.shape)} vs expected {(N, C)}"
    pkt = torch.cat([packed.contiguous().view(-1).view(torch.half), u.reshape(-1), v.reshape(-1)])
    return codecs.decompress(cid, pkt, base.contiguous(), N, C, 0)


@Profiler.prof_func("compact.binary_quant_fastpath")
def binary_quant_fastpath(x_tensor_nc: torch.Tensor, base_tensor_nc: torch.Tensor, rank: int, update_cache: bool):
    """-> packed (N, C//8) uint8, scale_u (N,K), scale_v (C,K), new_base (N,C) | None.   rank -1 (K = 1: mean scales) or 1..32."""
    assert rank >= 1 or rank == -1, "Rank must be >= 1 or -1"
    _check(x_tensor_nc, base_tensor_nc)
    if rank != -1:
        # scales = rank-K factors of |x - base| (fastpath.py:186-200: subspace_iter on the absolute residual): -> U (N, K), V (C, K)
        from . import lowrank
        x, base = x_tensor_nc.contiguous(), base_tensor_nc.contiguous()
        N, C = x.shape
        pkt = torch.empty(codecs.binary_rank_packet_halves(N, C, rank), dtype=torch.float16, device=x.device)
        nb = torch.empty_like(x) if update_cache else None
        codecs.binary_rank_compress_batch([x], [base], [nb], [pkt], [lowrank._start(C, rank, x.device)], N, C, rank, update_cache=update_cache)
        qh = N * (C // 8) // 2
        return pkt[:qh].view(torch.uint8).view(N, C // 8), pkt[qh:qh + N * rank].view(N, rank), pkt[qh + N * rank:].view(C, rank), nb
    return _quant(_BIN, 8, x_tensor_nc, base_tensor_nc, update_cache)


@Profiler.prof_func("compact.binary_dequant_fastpath")
def binary_dequant_fastpath(packed: torch.Tensor, scale_u_nk: torch.Tensor, scale_v_ck: torch.Tensor, base_nc: torch.Tensor):
    K = scale_u_nk.shape[1]
    if K > 1 or scale_v_ck.shape[1] > 1:              # rank-K scales (the rank is inferred from the scales, fastpath.py:400-404)
        N, Cp = packed.shape
        C = Cp * 8
        assert scale_u_nk.shape == (N, K) and scale_v_ck.shape == (C, K) and base_nc.shape == (N, C)
        pkt = torch.cat([packed.contiguous().view(-1).view(torch.half), scale_u_nk.reshape(-1), scale_v_ck.reshape(-1)])
        out = torch.empty((N, C), dtype=torch.float16, device=base_nc.device)
        codecs.binary_rank_decompress_batch([pkt], [base_nc.contiguous()], [out], N, C, K)
        return out
    return _dequant(_BIN, 8, packed, scale_u_nk, scale_v_ck, base_nc)


@Profiler.prof_func("compact.int2_quant_fastpath")
def int2_quant_fastpath(x_tensor_nc: torch.Tensor, base_tensor_nc: torch.Tensor, update_cache: bool, rank: int = -1):
    """-> packed (N, C//4) uint8, scale_u = token scale (N,1), scale_v = channel scale (C,1), new_base | None."""
    assert rank == -1, "INT2 fastpath only supports channel/token scales (rank=-1 equivalent)"
    _check(x_tensor_nc, base_tensor_nc)
    return _quant(_I2, 4, x_tensor_nc, base_tensor_nc, update_cache)


@Profiler.prof_func("compact.int2_dequant_fastpath")
def int2_dequant_fastpath(packed: torch.Tensor, scale_u_nk: torch.Tensor, scale_v_ck: torch.Tensor, base_nc: torch.Tensor):
    return _dequant(_I2, 4, packed, scale_u_nk, scale_v_ck, base_nc)


# ---- simulation twins: same results through the non-fused (residual-0) codec entry points ------------------------
def sim_binary_quant_fastpath(x_tensor_nc, base_tensor_nc, rank, update_cache):
    from .compress_quantize import dequantize_1bit, quantize_1bit
    delta = x_tensor_nc - base_tensor_nc
    packed, u_nk, v_kc = quantize_1bit(delta, rank=rank)
    nb = base_tensor_nc + dequantize_1bit(packed, u_nk, v_kc) if update_cache else None
    return packed, u_nk, v_kc.transpose(0, 1).contiguous(), nb


def sim_binary_dequant_fastpath(packed_in_nc8, scale_u_nk, scale_v_ck, base_nc):
    from .compress_quantize import dequantize_1bit
    return base_nc + dequantize_1bit(packed_in_nc8, scale_u_nk, scale_v_ck.transpose(0, 1).contiguous())


def sim_int2_quant_fastpath(x_tensor_nc, base_tensor_nc, update_cache, rank=-1):
    from .compress_quantize import dequantize_int2, quantize_int2
    assert rank == -1
    delta = x_tensor_nc - base_tensor_nc
    packed, chan, tok = quantize_int2(delta)
    nb = base_tensor_nc + dequantize_int2(packed, chan, tok) if update_cache else None
    return packed, tok, chan.T.contiguous(), nb


def sim_int2_dequant_fastpath(packed_in_nc4, scale_u_nk, scale_v_ck, base_nc):
    from .compress_quantize import dequantize_int2
    return base_nc + dequantize_int2(packed_in_nc4, scale_v_ck.T.contiguous(), scale_u_nk)
