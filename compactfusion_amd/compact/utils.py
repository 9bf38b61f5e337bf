"""Configuration, codec enumeration and the per-key state arena of the compressed exchange.

API mirror of the reference's `xfuser/compact/utils.py` (COMPACT_COMPRESS_TYPE :10-28, CompactConfig :31-117,
CompactCache :123-196) - same names, constructor keywords, legal-combination checks and key grammar
("{layer}-{rank}-{k|v}" in ring mode, "{layer}-{k|v}-{rank}" in gather mode) - on top of an MI355X-native state
store: every key owns ONE persistent device buffer (stable pointer, updated in place by the HIP kernels) instead of
the reference's dict of freshly allocated tensors that is re-pointed on every call (main.py:147).
"""
from __future__ import annotations

import os
from enum import Enum
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist

from .patchpara.df_utils import PatchConfig

ALLOW_DEPRECATED = os.environ.get("COMPACT_ALLOW_DEPRECATED", "0") == "1"


class COMPACT_COMPRESS_TYPE(Enum):
    """Same members and values as the reference (utils.py:19-28), plus INT8 which the reference only has as a cache
    quantiser (compress_quantize.py:428-484) and BASELINE.json's config 1 uses as a residual wire codec."""

    WARMUP = "warmup"
    SPARSE = "sparse"
    BINARY = "binary"
    INT2 = "int2"
    INT2_MINMAX = "int2-minmax"
    INT4 = "int4"
    IDENTITY = "identity"
    LOW_RANK = "low-rank"
    LOW_RANK_Q = "low-rank-int4"
    LOW_RANK_AWL = "low-rank-awl"
    INT8 = "int8"          # extension (not in the reference enum)


class CompactConfig:
    """Keyword-for-keyword the reference's CompactConfig (utils.py:33-106), including its consistency rules:
    residual 0 forbids error feedback, residual 2 and the fastpath require it, the fastpath needs residual 1 and no
    simulation, patch-gather needs `enabled` and a PatchConfig, compaction and async gather exclude each other."""

    def __init__(
        self,
        enabled: bool = False,
        override_with_patch_gather_fwd: bool = False,
        patch_gather_fwd_config: Optional[PatchConfig] = None,
        compress_func: Optional[Callable] = None,
        sparse_ratio=None,
        comp_rank=None,
        residual: int = 0,
        ef: bool = False,
        simulate: bool = False,
        log_stats: bool = False,
        check_consist: bool = False,
        fastpath: bool = False,
        quantized_cache: bool = False,
        delta_decay_factor: Optional[float] = None,
    ):
        assert residual in (0, 1, 2)
        # public attribute names are the reference's (utils.py:62-80); keyword -> attribute
        for attr, value in (
            ("enabled", enabled), ("compress_func", compress_func), ("sparse_ratio", sparse_ratio), ("comp_rank", comp_rank),
            ("compress_residual", residual), ("error_feedback", ef), ("simulate_compress", simulate),
            ("log_compress_stats", log_stats), ("check_cache_consistency", check_consist), ("fastpath", fastpath),
            ("quantized_cache", quantized_cache), ("delta_decay_factor", delta_decay_factor),
            ("override_with_patch_gather_fwd", override_with_patch_gather_fwd), ("patch_gather_fwd_config", patch_gather_fwd_config),
        ):
            setattr(self, attr, value)      # compress_func: (layer_idx, step) -> COMPACT_COMPRESS_TYPE

        rules = [
            (residual == 0 and ef, "No residual does not support error feedback."),
            (residual == 2 and not ef, "2nd order compression requires error feedback enabled."),
            (fastpath and not ef, "Fastpath requires error feedback enabled."),
            (fastpath and simulate, "Fastpath does not support simulation."),
            (fastpath and residual != 1, "Fastpath requires 1st order residual."),
        ]
        for broken, why in rules:
            assert not broken, why
        pc = patch_gather_fwd_config
        if override_with_patch_gather_fwd:
            assert enabled, "Compact must be enabled if override_with_patch_gather_fwd is True"
            assert pc is not None, "patch_gather_fwd_config must be set if override_with_patch_gather_fwd is True"
            assert not (pc.use_compact and pc.async_comm) or getattr(pc, "displaced_compact", False), \
                "Compact does not support async communication"
        else:
            assert pc is None, "patch_gather_fwd_config must be None if override_with_patch_gather_fwd is False"

    def get_compress_type(self) -> str:
        """Name used for result files (utils.py:108-117): the codec chosen for layer 0 at step 4."""
        if self.compress_func is None or not self.enabled:
            return "NO_COMPACT"
        t = self.compress_func(0, 4)
        return t.name if isinstance(t, COMPACT_COMPRESS_TYPE) else str(t)


class CompactCache:
    """key -> persistent (N, C) fp16 state buffer (+ optional second-order `delta_base`).

    `put` copies into the key's arena buffer unless it is handed that very buffer (what the in-place kernels do), so
    pointers stay stable for the life of the generation; tensors returned by `get_base` alias the arena and are valid
    until the next update of that key (the reference has the same aliasing: main.py:317-319)."""

    def __init__(self, quantize: bool = False):
        if quantize:
            # deprecated in the reference as well (utils.py:128-129); kept because it cuts the state memory in half
            assert ALLOW_DEPRECATED, "quantized cache is deprecated in the reference (utils.py:128-129): set COMPACT_ALLOW_DEPRECATED=1"
        self.quantize = quantize
        self.base: Dict[str, torch.Tensor] = {}
        self.delta_base: Dict[str, Optional[torch.Tensor]] = {}
        self.passed_count = 0
        self.version = 0        # bumped whenever a state buffer is (re)allocated: cached pointer tables key on it
        self._scratch: Dict[str, torch.Tensor] = {}      # quantize=True: per-key fp16 view handed out by get_base

    # -- arena ----------------------------------------------------------------------------------------------
    def arena(self, key: str, like: torch.Tensor) -> torch.Tensor:
        """The key's persistent buffer, (re)allocated to match `like` (shape (N, C), fp16, same device)."""
        buf = self.base.get(key)
        if buf is None or buf.shape != like.shape or buf.device != like.device or buf.dtype != like.dtype:
            buf = torch.empty(like.shape, dtype=like.dtype, device=like.device)
            self.base[key] = buf
            self.delta_base.setdefault(key, None)
            self.version += 1
        return buf

    def _store(self, slot: Optional[torch.Tensor], value: torch.Tensor) -> torch.Tensor:
        if slot is None or slot.shape != value.shape or slot.device != value.device or slot.dtype != value.dtype:
            slot = torch.empty(value.shape, dtype=value.dtype, device=value.device)
            self.version += 1
        if slot.data_ptr() != value.data_ptr():
            slot.copy_(value)
        return slot

    def touch(self, key) -> None:
        """The state of `key` was updated in place by a kernel: only the reference's `put` side effect (the collector
        hook for K / V keys, utils.py:138-143) is left to do."""
        from .main import compact_get_step
        from ..collector.collector import collect
        if "k" in key:
            collect(self.base[key], "kbase", compact_get_step(), int(key.split("-")[0]))
        elif "v" in key:
            collect(self.base[key], "vbase", compact_get_step(), int(key.split("-")[0]))

    def put(self, key, base, delta_base):
        if self.quantize:
            # int8 storage of the base (utils.py:135-137 -> compress_quantize.py:428-468): the native INT8 codec applied to
            # the tensor itself (no residual); `self.base[key]` holds the packet [q | scale | zero point]
            from .. import codecs
            N, C = base.shape
            pkt = self.base.get(key)
            need = codecs.packet_halves(codecs.Codec.INT8, N, C)
            if pkt is None or pkt.numel() != need or pkt.device != base.device:
                pkt = torch.empty(need, dtype=torch.float16, device=base.device)
                self.base[key] = pkt
                self.version += 1
            codecs.compress_batch(codecs.Codec.INT8, [base.contiguous()], [None], [None], [pkt], N, C, update_cache=False)
            self._scratch[key] = self._store(self._scratch.get(key), base)      # shape / dtype carrier, overwritten by get_base
        else:
            self.base[key] = self._store(self.base.get(key), base)
        self.touch(key)
        if delta_base is None:
            self.delta_base[key] = None
        else:
            self.delta_base[key] = self._store(self.delta_base.get(key), delta_base)

    def get_base(self, key):
        if not self.quantize:
            return self.base.get(key, None)
        pkt = self.base.get(key)
        if pkt is None:
            return None
        from .. import codecs
        out = self._scratch[key]
        N, C = out.shape
        codecs.decompress_batch(codecs.Codec.INT8, [pkt], [None], [out], N, C)      # dequantize_int8 (utils.py:149-153)
        return out

    def get_delta_base(self, key):
        return self.delta_base.get(key, None)

    # -- debugging aid (collective C6 of SURVEY.md §2.3) -----------------------------------------------------------
    def check_consistency(self, group=None):
        """Every rank must hold the same state for every key: all-reduce(SUM)/W must reproduce the local copy
        (atol 1e-2, utils.py:164-196)."""
        if group is None:
            group = dist.group.WORLD
        world = dist.get_world_size(group)
        if world <= 1:
            return
        for key in sorted(self.base.keys()):
            parts = [t.flatten() for t in (self.get_base(key), self.get_delta_base(key)) if t is not None]
            if not parts:
                continue
            mine = torch.cat(parts).float()
            mean = mine.clone()
            dist.all_reduce(mean, op=dist.ReduceOp.SUM, group=group)
            mean /= world
            worst = float((mine - mean).abs().max())
            assert torch.allclose(mine, mean, atol=1e-2), f"Inconsistent cache at key {key}, max diff: {worst:.6f}"
        self.passed_count += 1


def get_emoji():
    return ""
