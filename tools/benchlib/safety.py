"""bench.py, part (c): the N > 1 safety net - states_consistent, validate, the fall-back ladder.

At N > 1 the packets are read in place from the peers' memory (peer-to-peer) or delivered by a collective kernel that has to find CUs beside
the waiting reconstruction workgroups; neither has ever run on more than one GPU.  So a multi-rank run is VALIDATED after the warm-up steps
and AGAIN after the timed region (a stale cache line only shows from the second use of an address on): gate time-outs, and every rank's
reconstruction of a shard against its owner's state.  A failed check never ends the run: every rank - the verdict is all-reduced, all
ranks decide alike - drops to the next schedule of the LADDER

    p2p (packets read in place)  ->  native (compress ; ncclAllGather ; reconstruct, two launches per layer in stream order, libcfx's own
    communicator)  ->  torch (compress ; torch.distributed.all_gather_into_tensor ; reconstruct, issued per layer from Python)

resets its states, and warm-up + timed region run again; `Ladder.text` records which check tripped and what the run continued as.

Nothing in this module touches libcfx or needs a GPU: the tensors may live on any device, `dist` is torch.distributed (any backend) and
the gate-error count comes in as a number.  tests/test_bench_safety.py drives it with two gloo processes on CPU tensors - a consistent
state, a poisoned reconstruction on one rank, a gate time-out on one rank - and tests/test_gpu_bench.py with two processes on one GPU.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

SAMPLE_HALVES = 8192          # halves of a tensor that travel in a consistency check (as int32: a dtype every backend moves)


def sample_keys(L: int, G: int):
    """(layer, k|v) pairs a consistency check looks at: layers that sit at different positions of an all-gather group, K and V."""
    return sorted({(l, kv) for l in (0, 1, min(L - 1, G - 1), L // 2, L - 1) for kv in (0, 1) if 0 <= l < L})


def states_consistent(torch, dist, own_base, peer_base, rank: int, live: int, real_live: int, G: int, n_logical_peers: int) -> Tuple[bool, str]:
    """What a rank holds for its own shard must be, bit for bit, what every peer reconstructed for that shard.

    own_base   [L, 2, N, C] fp16: this rank's sender (error-feedback) states
    peer_base  [L, W - 1, 2, N, C] fp16: logical peer p = rank (rank + 1 + p) mod live for p < live - 1, looped-back copies of our own beyond
    One rank (real_live == 1): every looped-back peer state equals the sender's.  More: the first SAMPLE_HALVES halves of every sampled
    tensor are all-gathered and compared with what this rank reconstructed; the verdict is the MIN over the ranks - the same on all."""
    L = own_base.shape[0]
    samples = sample_keys(L, G)
    i16 = torch.int16
    if real_live == 1:
        same = all(torch.equal(own_base[l, kv].view(i16), peer_base[l, p, kv].view(i16)) for l, kv in samples for p in range(n_logical_peers))
        return same, "EF state of a looped-back peer diverged from the sender's"
    dev = own_base.device
    good = torch.ones(1, dtype=torch.int32, device=dev)
    n32 = SAMPLE_HALVES // 2
    for l, kv in samples:
        mine = own_base[l, kv].reshape(-1)[:SAMPLE_HALVES].view(torch.int32).contiguous()
        allm = torch.empty(live * n32, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allm, mine)
        for p in range(live - 1):
            src = (rank + 1 + p) % live
            got = peer_base[l, p, kv].reshape(-1)[:SAMPLE_HALVES].view(torch.int32)
            if not torch.equal(got, allm[src * n32:(src + 1) * n32]):
                good.zero_()
    dist.all_reduce(good, op=dist.ReduceOp.MIN)
    return bool(good.item()), "on at least one rank a peer's reconstructed state diverged from its owner's"


def validate(torch, dist, label: str, use_dist: bool, world: int, gate_errors: int, consistent: Callable[[], Tuple[bool, str]], dev) -> Optional[str]:
    """None when this rank AND every other rank is fine, else what tripped - the same answer on every rank (MAX over the ranks of:
    2 a gate / flag wait timed out, 1 a reconstructed state differs, 0 fine).  `gate_errors`: this rank's count (cfx_gate_errors, read
    after a device synchronisation); `consistent`: this rank's states_consistent, called on EVERY rank (it is collective)."""
    if not use_dist:
        return None
    ok, why = consistent()
    bad = torch.tensor([2 if gate_errors else (0 if ok else 1)], device=dev, dtype=torch.int32)
    if world > 1:
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    code = int(bad.item())
    if code == 0:
        return None
    return label + ": " + ("a gate / flag wait timed out (the packets did not arrive in time)" if code == 2 else
                           "a reconstructed state differs from its owner's (" + why + ")")


class Ladder:
    """The schedules a multi-rank run can fall back through, in order.  `rung` is what runs now:

        "p2p"     the peer-to-peer exchange layer (packets read in place through IPC mappings), one launch per layer
        "xgate"   the exchange-layer launch around ncclAllGather (collective in the path), one launch per layer
        "native"  compress ; ncclAllGather ; reconstruct - two launches per layer in stream order, libcfx's own RCCL communicator
        "torch"   compress ; torch.distributed.all_gather_into_tensor ; reconstruct, issued per layer from Python

    `down(reason, have_native_comm, world)` moves to the next rung that exists here and returns it; when none is left it raises SystemExit.
    Every rank calls it with the same (all-reduced) reason, so every rank lands on the same rung.  `text` accumulates the story for the
    bench line's `schedule_fallback`."""

    NAMES = {"p2p": "the peer-to-peer exchange layer (packets read in place through IPC mappings)",
             "xgate": "the exchange-layer launch around ncclAllGather",
             "native": "two launches per layer around ncclAllGather",
             "torch": "torch.distributed per layer"}
    NOW = {"native": "compress ; ncclAllGather ; reconstruct, two launches per layer in stream order (libcfx's own RCCL communicator)",
           "torch": "compress ; torch.distributed.all_gather_into_tensor ; reconstruct, issued per layer from Python"}

    def __init__(self, rung: str, text: Optional[str] = None):
        assert rung in self.NAMES
        self.rung, self.text = rung, text

    def down(self, reason: str, have_native_comm: bool, world: int, stream_mode: int = 0) -> str:
        was = self.NAMES[self.rung]
        if (self.rung in ("p2p", "xgate") or stream_mode != 0) and have_native_comm:
            new = "native"
        elif self.rung != "torch" and world > 1:
            new = "torch"
        else:
            raise SystemExit(f"[bench] {was} failed validation ({reason}) and no schedule is left to fall back to")
        self.text = ((self.text + " ; then " if self.text else "") + was + " failed validation - " + reason + " - and the run continued as: " + self.NOW[new])
        self.rung = new
        return new
