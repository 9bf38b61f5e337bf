"""Patch-gather configuration and the bookkeeping of in-flight asynchronous gathers.

Restates `xfuser/compact/patchpara/df_utils.py` (PatchConfig) and `df_cache.py` (AllGatherCache, DummyHandle);
`df_utils.py` / `df_cache.py` in this package re-export these names for import compatibility."""
from __future__ import annotations

from typing import Dict, List, NamedTuple

import torch


class PatchConfig:
    """use_compact: compress the gathered K/V; async_comm: DistriFusion-style displaced gather (consume the previous
    step's buffers while this step's gather is in flight); the two are mutually exclusive (df_utils.py:13-16)."""

    def __init__(self, use_compact: bool, async_comm: bool, async_warmup: int, displaced_compact: bool = False) -> None:
        # displaced_compact (extension): the compressed gather of step t is consumed at
        # step t+1 (DistriFusion staleness on top of residual compression); it is the only way to combine the two flags
        if use_compact and async_comm and not displaced_compact:
            raise AssertionError("Compact does not support async communication" if use_compact
                                 else "Async communication does not support compact")
        if displaced_compact and not (use_compact and async_comm):
            raise AssertionError("displaced_compact needs use_compact=True and async_comm=True")
        self.use_compact = bool(use_compact)
        self.async_comm = bool(async_comm)
        self.async_warmup = async_warmup
        self.displaced_compact = bool(displaced_compact)


class DummyHandle:
    """Stands in for a collective work handle during warm-up steps, where the gather already completed."""

    def wait(self):
        return None


class _Entry(NamedTuple):
    handle: object
    recv: List[torch.Tensor]
    send: torch.Tensor


HANDLES_IDX, RECV_BUF_IDX, SEND_BUF_IDX, ENTRY_VAL_LEN = 0, 1, 2, 3


class AllGatherCache:
    def __init__(self):
        self.cache: Dict[str, _Entry] = {}

    def clear(self):
        self.cache.clear()

    def put(self, key, handle, recv_buf_list, send_buf):
        if not isinstance(recv_buf_list, list) or not isinstance(send_buf, torch.Tensor):
            raise AssertionError("AllGatherCache.put(key, handle, list_of_recv_buffers, send_tensor)")
        self.cache[key] = _Entry(handle, recv_buf_list, send_buf)

    def get(self, key):
        return self.cache[key]

    def contains(self, key):
        return key in self.cache

    def tensors_size(self) -> int:
        """Bytes held by all send and receive buffers."""
        total = 0
        for ent in self.cache.values():
            for t in [ent.send, *ent.recv]:
                if t is not None:
                    total += t.numel() * t.element_size()
        return total
