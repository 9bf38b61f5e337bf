"""Optional activation dumps (q/k/v, cached bases, latents) for offline analysis.

Keeps the reference's contract (`xfuser/collector/collector.py`): a module-level singleton installed with `init()`,
`collect(tensor, type, step, layer)` raising when nothing was installed, and the on-disk layout
`<dir>/rank_<r>/step_<s>[/layer_<l>]/<type>.pt`.  `CompactCache.put` calls `collect` for K/V keys (utils.py:138-143).
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, Optional

import torch

COLLECT_TYPE = ("q", "k", "v", "kbase", "vbase", "latents")


class Collector:
    def __init__(self, save_dir: str, target_steps: Optional[Iterable[int]] = None,
                 target_layers: Optional[Iterable[int]] = None, enabled: bool = False, rank: int = 0):
        self.root = Path(save_dir)
        self.save_dir = save_dir
        self.target_steps = None if target_steps is None else set(target_steps)
        self.target_layers = None if target_layers is None else set(target_layers)
        self.enabled = enabled
        self.rank = rank

    def _wanted(self, step, layer) -> bool:
        if self.target_steps is not None and step not in self.target_steps:
            return False
        return self.target_layers is None or layer in self.target_layers

    def collect(self, tensor: torch.Tensor, type: str, step: int, layer: int):
        if not self.enabled:
            return
        if type not in COLLECT_TYPE:
            raise ValueError(f"Invalid collect type: {type}")
        if not self._wanted(step, layer):
            return
        where = self.root / f"rank_{self.rank}" / f"step_{step}"
        if type != "latents":
            where = where / f"layer_{layer}"
        else:
            assert layer is None, "latents are not layer specific"
        where.mkdir(parents=True, exist_ok=True)
        torch.save(tensor.detach().to("cpu"), where / f"{type}.pt")


instance: Optional[Collector] = None


def init(collector: Collector):
    global instance
    instance = collector


def collect(tensor: torch.Tensor, type: str, step: int, layer: int):
    if instance is None:
        raise ValueError("Collector not initialized")
    instance.collect(tensor, type, step, layer)
