"""Compression statistics (error, norms, wire volume) - the subset of the reference's `xfuser/compact/stats.py`
(StatsLogger.log :107-328, summary_compression_volume :508-526) that the hot path calls when
`CompactConfig(log_stats=True)`: per key and step it records the reconstruction error, activation / residual
norms and packet vs raw byte volumes.  Plots, eigen-spectra and activation dumps of the reference are out of scope
(SURVEY.md §8f rank 3)."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional

import torch


class StatsLogger:
    def __init__(self):
        self.records: Dict[str, List[dict]] = defaultdict(list)

    def log(self, key, base, delta_base, before_comp_activation, recv_activation, compressed_tensor, compress_residual,
            ref_activation_path: Optional[str] = None):
        from .main import compact_get_step
        x = before_comp_activation.float()
        rec = recv_activation.float()
        row = {
            "step": compact_get_step(),
            "error": float((x - rec).norm()),
            "act_norm": float(x.norm()),
            "raw_bytes": before_comp_activation.numel() * before_comp_activation.element_size(),
            "wire_bytes": compressed_tensor.numel() * compressed_tensor.element_size(),
            "residual": compress_residual,
        }
        if base is not None:
            d = x - base.float()
            row["delta_norm"] = float(d.norm())
            if delta_base is not None:
                row["delta_delta_norm"] = float((d - delta_base.float()).norm())
        self.records[key].append(row)

    def summary_compression_volume(self):
        raw = sum(r["raw_bytes"] for rows in self.records.values() for r in rows)
        wire = sum(r["wire_bytes"] for rows in self.records.values() for r in rows)
        return {"raw_bytes": raw, "wire_bytes": wire, "ratio": (raw / wire) if wire else float("nan")}

    def summary_error(self):
        out = {}
        for key, rows in self.records.items():
            rel = [r["error"] / r["act_norm"] for r in rows if r["act_norm"] > 0]
            out[key] = sum(rel) / len(rel) if rel else 0.0
        return out


_logger = StatsLogger()


def stats_log() -> StatsLogger:
    return _logger


def stats_clear():
    global _logger
    _logger = StatsLogger()


def stats_hello():
    print("compactfusion_amd stats logging enabled")


def stats_verbose():
    vol = _logger.summary_compression_volume()
    print(f"compression volume: raw {vol['raw_bytes']} B, wire {vol['wire_bytes']} B, ratio {vol['ratio']:.2f}x")


def stats_verbose_steps(keys=None):
    for key, rows in _logger.records.items():
        if keys is not None and key not in keys:
            continue
        for r in rows:
            print(f"[{key}] step {r['step']}: err {r['error']:.4f} act {r['act_norm']:.4f}")
