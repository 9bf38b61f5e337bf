"""Low-rank residual codecs: LOW_RANK (rank-r factors) and LOW_RANK_Q (rank-r factors, int4-quantised).

Host side of `compactfusion_amd/csrc/cfx_lowrank.hip`, which replaces the reference's `subspace_iter`
(`xfuser/compact/compress_lowrank.py:14-61`) and the LOW_RANK / LOW_RANK_Q branches of `slowpath.py`
(:54-75 encode, :120-131 + :151-164 decode) by a fixed chain of gfx950 kernels (randomised subspace iteration with
fp64 Cholesky-QR, factor emission straight into the wire packet, fused decode + residual add):

    LOW_RANK   wire  [ U (N,r) fp16 | V (r,C) fp16 ]                                   decode  U @ V
    LOW_RANK_Q wire  [ q4(U) (N/2,r) | sU r | mU r | q4(V^T) (C/2,r) | sV r | mV r ]   decode  deq(U) @ deq(V^T)^T

Only the random start matrix is made here (torch.randn on the device, as the reference does at compress_lowrank.py:41);
`set_init_q` pins it for reproducible tests.  `subspace_iter` / `svd` below are the reference's function-level API
(library GEMM / QR through PyTorch, any device) kept for callers and tests that want the factors themselves.
Parity is tolerance-based exactly as in the reference's own tests (compress_slowpath_test.py:128-216): the result
depends on the random start and, in the last bits, on the GEMM / QR rounding order.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import codecs
from ..prof import Profiler
from .utils import COMPACT_COMPRESS_TYPE as T

LOW_RANK_ID, LOW_RANK_Q_ID = 101, 102
BINARY_RANK_ID = 103          # 1-bit codec whose scales are rank-K factors of |x - base| (fastpath.py:88-120; deprecated in the reference)
_pinned_q: Optional[torch.Tensor] = None


def native_id(compress_type) -> int:
    return LOW_RANK_ID if compress_type == T.LOW_RANK else LOW_RANK_Q_ID


def set_init_q(q: Optional[torch.Tensor]) -> None:
    """Pin the (C, r) start matrix used by the next compress calls (None: back to torch.randn)."""
    global _pinned_q
    _pinned_q = q


def _start(C: int, rank: int, device, rp: Optional[int] = None) -> torch.Tensor:
    rp = codecs.lr_rank_pad(rank) if rp is None else rp
    q = torch.zeros(C, rp, dtype=torch.float32, device=device)
    if _pinned_q is not None:
        assert tuple(_pinned_q.shape) == (C, rank), f"pinned init_q must be ({C}, {rank})"
        q[:, :rank] = _pinned_q.to(device=device, dtype=torch.float32)
    else:
        q[:, :rank] = torch.randn(C, rank, device=device, dtype=torch.float32)
    return q


def packet_halves(cid: int, rank: int, N: int, C: int) -> int:
    if cid == BINARY_RANK_ID:
        return codecs.binary_rank_packet_halves(N, C, rank)
    return codecs.lr_packet_halves(cid == LOW_RANK_Q_ID, N, C, rank)


@Profiler.prof_func("compact.lowrank.compress")
def compress(cid: int, rank: int, x: torch.Tensor, base: Optional[torch.Tensor], new_base: Optional[torch.Tensor],
             packet: torch.Tensor, update: bool, ef: bool = True) -> None:
    N, C = x.shape
    if cid == BINARY_RANK_ID:
        codecs.binary_rank_compress_batch([x], [base], [new_base if update else None], [packet], [_start(C, rank, x.device)], N, C, rank,
                                          update_cache=update, ef=ef)
        return
    codecs.lr_compress_batch(cid == LOW_RANK_Q_ID, [x], [base], [new_base if update else None], [packet],
                             [_start(C, rank, x.device)], N, C, rank, update_cache=update, ef=ef)


@Profiler.prof_func("compact.lowrank.decompress")
def decompress(cid: int, rank: int, packet: torch.Tensor, base: Optional[torch.Tensor], out: torch.Tensor) -> None:
    N, C = out.shape
    if cid == BINARY_RANK_ID:
        codecs.binary_rank_decompress_batch([packet], [base], [out], N, C, rank)
        return
    codecs.lr_decompress_batch(cid == LOW_RANK_Q_ID, [packet], [base], [out], N, C, rank)


# ---- slowpath.py-level API ----------------------------------------------------------------------------------------------
def slowpath_compress(x: torch.Tensor, compress_type, rank: int) -> torch.Tensor:
    cid = native_id(compress_type)
    N, C = x.shape
    pkt = torch.empty(packet_halves(cid, rank, N, C), dtype=torch.float16, device=x.device)
    compress(cid, rank, x, None, None, pkt, update=False)
    return pkt


def slowpath_decompress(x: torch.Tensor, shape: Tuple[int, int], compress_type, rank: int) -> torch.Tensor:
    cid = native_id(compress_type)
    N, C = shape
    assert x.numel() == packet_halves(cid, rank, N, C)
    out = torch.empty((N, C), dtype=torch.float16, device=x.device)
    if x.data_ptr() % 16:
        x = x.clone()
    decompress(cid, rank, x, None, out)
    return out


# ---- the reference's function-level API (library path, any device) ----------------------------------------------------------
def svd(input_tensor: torch.Tensor, rank: int):
    """Truncated SVD factorisation (compress_lowrank.py:5-12): returns (U S, V^T) in the input dtype."""
    U, S, Vh = torch.linalg.svd(input_tensor.float(), full_matrices=False)
    return (U[:, :rank] * S[:rank]).to(input_tensor.dtype), Vh[:rank, :].to(input_tensor.dtype)


@Profiler.prof_func("compact.subspace_iter")
def subspace_iter(A: torch.Tensor, rank: int, num_iters: int = 10, init_q: Optional[torch.Tensor] = None):
    """A (m, n) -> U (m, r) orthonormal, V (r, n) with A ~ U V, and the right basis Q (n, r); outputs in A.dtype."""
    dtype = A.dtype
    Af = A.float()
    n = Af.shape[1]
    if init_q is None:
        Q, _ = torch.linalg.qr(torch.randn(n, rank, device=A.device, dtype=torch.float))
    else:
        Q = init_q.float()
    for _ in range(num_iters):
        Q, _ = torch.linalg.qr(Af.t() @ (Af @ Q))
    U, _ = torch.linalg.qr(Af @ Q)
    V = U.t() @ Af
    return U.to(dtype), V.to(dtype), Q.to(dtype)
