"""Sequence-parallel process groups for the exchange path.

The reference builds these in `xfuser/core/distributed/parallel_state.py:419-437` by calling yunchang's
`set_seq_parallel_pg(ulysses_degree, ring_degree, rank, world_size)` and wraps them in a
`SequenceParallelGroupCoordinator` (`group_coordinator.py:1046-1080`) whose `.ring_group`, `.ulysses_group`,
`.ring_world_size`, `.ring_rank`, `.ulysses_world_size`, `.ulysses_rank` are what the hot path reads.  This module
provides the same object for a sequence-parallel group spanning the whole world (the other xDiT parallelisms - data,
CFG, PipeFusion, tensor, VAE - are out of scope).

Rank layout = yunchang's default (`use_ulysses_low=True`): Ulysses groups are runs of consecutive ranks, ring groups
stride across them, e.g. world 8, ulysses 2, ring 4: ulysses {0,1} {2,3} {4,5} {6,7}; ring {0,2,4,6} {1,3,5,7}.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch.distributed as dist


@dataclass
class SequenceParallelGroup:
    ulysses_group: object
    ring_group: object
    ulysses_world_size: int
    ulysses_rank: int
    ring_world_size: int
    ring_rank: int
    world_size: int
    rank: int


_SP: Optional[SequenceParallelGroup] = None


def init_sequence_parallel(ulysses_degree: int = 1, ring_degree: Optional[int] = None, backend: Optional[str] = None) -> SequenceParallelGroup:
    """Create the Ulysses and ring sub-groups of the default (world) group; every rank must call it."""
    global _SP
    assert dist.is_initialized(), "torch.distributed must be initialised first (one process per GPU)"
    world, rank = dist.get_world_size(), dist.get_rank()
    if ring_degree is None:
        ring_degree = world // ulysses_degree
    assert ulysses_degree * ring_degree == world, f"ulysses_degree ({ulysses_degree}) x ring_degree ({ring_degree}) must equal the world size ({world})"
    u_group = r_group = None
    for i in range(ring_degree):                       # consecutive ranks share a Ulysses group
        ranks = list(range(i * ulysses_degree, (i + 1) * ulysses_degree))
        g = dist.new_group(ranks, backend=backend)
        if rank in ranks:
            u_group = g
    for j in range(ulysses_degree):                    # ring groups stride over the Ulysses groups
        ranks = list(range(j, world, ulysses_degree))
        g = dist.new_group(ranks, backend=backend)
        if rank in ranks:
            r_group = g
    _SP = SequenceParallelGroup(u_group, r_group, ulysses_degree, dist.get_rank(u_group), ring_degree, dist.get_rank(r_group), world, rank)
    return _SP


def get_sp_group() -> SequenceParallelGroup:
    assert _SP is not None, "sequence parallel group is not initialized (call init_sequence_parallel)"
    return _SP


def destroy_sequence_parallel() -> None:
    global _SP
    _SP = None
