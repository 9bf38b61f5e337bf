"""Low-rank residual codecs (LOW_RANK, LOW_RANK_Q) - SURVEY.md §8(f) rank 2, not built yet in this round."""
from __future__ import annotations


def native_id(compress_type):
    raise NotImplementedError("LOW_RANK / LOW_RANK_Q codecs are the next row of the scope table (SURVEY.md §8f) and are not implemented yet")


def slowpath_compress(x, compress_type, rank):
    native_id(compress_type)


def slowpath_decompress(x, shape, compress_type, rank):
    native_id(compress_type)
