"""1:m block top-1 sparsifier - mirror of `xfuser/compact/compress_topk.py` (topk_compress :11-41, topk_decompress
:108-125, topk_sparsify :165-193, sim_topk :221-235), same signatures; the work is one gfx950 launch each."""
from __future__ import annotations

import torch

from .. import codecs

SPARSE_LAST_DIM_SIZE = 1024
VALID_COMPRESS_LEVELS = [1, 2, 4, 8, 16]
_TK = int(codecs.Codec.TOPK)


def topk_compress(input_tensor: torch.Tensor, m: int):
    """input (A, 1024) fp16 -> val (A, 1024/m) fp16, idx (A, 512/m) uint8 with (i1 << 4) | i2 per 2m block."""
    A, L = input_tensor.shape
    assert L == SPARSE_LAST_DIM_SIZE and L % (2 * m) == 0, "The number of columns must be 1024 and divisible by 2*m."
    x = input_tensor.contiguous()
    pkt, _ = codecs.compress(_TK, x, None, A, L, m, update_cache=False)
    nv = A * L // m
    return pkt[:nv].view(A, L // m), pkt[nv:].view(torch.uint8).view(A, L // (2 * m))


def topk_decompress(compressed_val_tensor: torch.Tensor, compressed_idx_tensor: torch.Tensor, m: int):
    A, B = compressed_idx_tensor.shape
    L = 2 * m * B
    pkt = torch.cat([compressed_val_tensor.contiguous().view(-1), compressed_idx_tensor.contiguous().view(-1).view(torch.half)])
    return codecs.decompress(_TK, pkt, None, A, L, m)


def topk_sparsify(input_tensor: torch.Tensor, m: int):
    """Keep the largest-|x| element of every m consecutive elements, zero the rest."""
    shp = input_tensor.shape
    x = input_tensor.contiguous().view(-1, SPARSE_LAST_DIM_SIZE)
    return topk_decompress(*topk_compress(x, m), m).view(shp)


def sim_topk(x: torch.Tensor, m: int):
    assert x.shape[-1] % m == 0
    return topk_sparsify(x, m)
