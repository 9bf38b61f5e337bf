"""Optional activation dumps (q / k / v, cached bases, latents) for offline analysis.

Contract kept from the reference (`xfuser/collector/collector.py`): a module-level singleton installed with `init()`;
`collect(tensor, type, step, layer)` raises `ValueError("Collector not initialized")` when nothing was installed (the hot
path calls it from `CompactCache.put` for K / V keys, utils.py:138-143); files land in
`<dir>/rank_<r>/step_<s>[/layer_<l>]/<type>.pt`.
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, Optional

import torch

COLLECT_TYPE = ("q", "k", "v", "kbase", "vbase", "latents")


class Collector:
    """enabled=False makes every call a no-op; target_steps / target_layers (None = all) filter what is written."""

    def __init__(self, save_dir: str, target_steps: Optional[Iterable[int]] = None,
                 target_layers: Optional[Iterable[int]] = None, enabled: bool = False, rank: int = 0):
        self.save_dir = save_dir
        self.enabled, self.rank = enabled, rank
        self.target_steps = None if target_steps is None else frozenset(target_steps)
        self.target_layers = None if target_layers is None else frozenset(target_layers)

    def _destination(self, kind: str, step, layer) -> Path:
        parts = [f"rank_{self.rank}", f"step_{step}"]
        if kind == "latents":
            if layer is not None:
                raise AssertionError("latents are not layer specific")
        else:
            parts.append(f"layer_{layer}")
        return Path(self.save_dir).joinpath(*parts)

    def collect(self, tensor: torch.Tensor, type: str, step: int, layer: int):
        if not self.enabled:
            return
        if type not in COLLECT_TYPE:
            raise ValueError(f"Invalid collect type: {type}")
        skip_step = self.target_steps is not None and step not in self.target_steps
        skip_layer = self.target_layers is not None and layer not in self.target_layers
        if skip_step or skip_layer:
            return
        folder = self._destination(type, step, layer)
        folder.mkdir(parents=True, exist_ok=True)
        torch.save(tensor.detach().to("cpu"), folder / f"{type}.pt")


instance: Optional[Collector] = None


def init(collector: Collector):
    """Install the process-wide collector."""
    global instance
    instance = collector


def collect(tensor: torch.Tensor, type: str, step: int, layer: int):
    if instance is None:
        raise ValueError("Collector not initialized")
    instance.collect(tensor, type, step, layer)
