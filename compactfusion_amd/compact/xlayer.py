"""One layer's residual-compressed K,V exchange as ONE native op - the product path of the gather schedules.

What the reference does per layer and denoise step in Python (xfuser/compact/main.py:390-420 `compact_all_gather`: compress, list
all-gather, W decompress calls; patchpara/fwd.py:88-102; ring.py:188-206 + 265-269: compress K and V, W-1 relay hops, a
decompress per hop) is here ONE host call into libcfx per layer: `cfx_plan_add_exchange_layer[_p2p]` replayed by `cfx_plan_run_x`
(include/cfx.h).  For every streaming codec (1-bit, 2-bit, int4, int8, top-k) that is ONE codec launch whose reconstruction workgroups wait, state
tiles already in registers, for the packets' arrival - and in the peer-to-peer transport the exchange itself (publish this rank's word, await
the peers') runs inside that launch; top-k and shapes without the one-launch form run the same work in stream order
(compress ; exchange ; reconstruct) - same results, still one host call.

Transports, tried in this order (`CFX_EXCHANGE` = auto | p2p | rccl | torch):
  p2p    ranks of ONE node: every rank's packets stay in uncached IPC device memory of its own GPU (`P2PArena`), the peers'
         reconstruction workgroups read them in place over xGMI; what is exchanged is one word per rank and layer, by the launch itself
         (no exchange stream, any run stream).  The first VALIDATE_FIRST
         executions of every layer - through the first REUSE of both packet parities, where a stale line in a reader's cache would first
         show - are VALIDATED (gate time-outs, and every rank's reconstruction of a shard against what its owner says the peers must
         hold, by checksum over the process group); arenas whose memory did not come out uncached keep validating every
         REVALIDATE_EVERY-th execution.  On any failure every rank restores the layer's states, the peers' copies are re-synchronised
         from the owners' states, the group's arena is marked bad and all its layers continue on the next transport.  A gate time-out in
         the steady path (a peer more than the gate timeout late) stores nothing (include/cfx.h: cfx_gate_recover); the ranks agree on it
         at the next step boundary (`GroupHealth`) and validate every layer again.
  rccl   ncclAllGather issued by libcfx's own communicator between a flag-wait and a flag-set kernel on the exchange stream.
  torch  compress ; torch.distributed.all_gather_into_tensor ; reconstruct (three host calls) - the last resort.
"""
from __future__ import annotations

import ctypes
import os
import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .. import _lib, codecs
from .. import config as _settings

MAX_ITEMS = codecs.CFX_MAX_BATCH
VALIDATE_FIRST = 4           # validated executions of every p2p layer: parity 0, parity 1, and the first reuse of each
REVALIDATE_EVERY = 64        # ... and every so many executions afterwards where the arena's memory is not uncached (cfx_ipc_memory_kind != 2)


def transport_pref() -> str:
    """compactfusion_amd.configure(exchange=..., ring_exchange=..., ring_p2p=...) (compactfusion_amd/config.py)"""
    pref = _settings.get("exchange")
    if _settings.get("ring_exchange") == "torch":
        pref = "torch"
    if _settings.get("ring_p2p") in ("0", "off"):
        pref = "rccl" if pref in ("auto", "p2p") else pref
    return pref


# ---------------------------------------------------------------------------------------------------------------------------
# IPC-shared packet memory
# ---------------------------------------------------------------------------------------------------------------------------
class _Region:
    """A layer's slice of the arena: [parity 0: K slot | V slot][parity 1: K slot | V slot][flag word of parity 0, 64 B][parity 1]."""

    __slots__ = ("own", "peer", "slot", "n_exec", "validated", "recheck")

    def __init__(self, own: int, peer: Dict[int, int], slot: int):
        self.own, self.peer, self.slot = own, peer, slot
        self.n_exec = 0          # executions so far: the NEXT one writes parity n_exec & 1 (kept across plan rebuilds, like the device-side epochs)
        self.validated = 0       # validated executions so far (p2p with real peers: the first VALIDATE_FIRST are checked)
        self.recheck = 0         # executions still to validate after the group agreed that somebody's wait had timed out

    def packet(self, r: Optional[int], parity: int, kv: int) -> int:
        base = self.own if r is None else self.peer[r]
        return base + (2 * parity + kv) * self.slot

    def flag(self, r: Optional[int], parity: int) -> int:
        base = self.own if r is None else self.peer[r]
        return base + 4 * self.slot + 64 * parity


class GroupHealth:
    """Do the ranks of a group agree that nobody's in-launch wait has timed out?  A time-out is local knowledge (the late rank itself
    notices nothing), but what follows - validating the layers again - is collective.  Once per denoise step (the first layer op that
    sees a new step number) every rank contributes its flag to a MAX all-reduce whose result lands in pinned host memory and is read
    at the NEXT boundary: no host synchronisation in the steady path, the agreement is one step late and the same on every rank."""

    def __init__(self, group, device: int):
        self.group, self.device = group, device
        self.local_error = False
        self.step = None
        self.first_key = None        # the layer key that opened the current boundary interval (fallback boundary: seen again = one pass over the layers)
        self.pending = None          # (event, pinned result) of the all-reduce issued at the previous boundary
        self._flag, self._host, self._turn = None, None, 0

    def note_error(self) -> None:
        self.local_error = True

    def _new_interval(self, step, key) -> bool:
        """Is this call the first of a new boundary interval?  A new step number (compact_set_step) - or, for callers that never set one
        or sit on one step (such a run never reached a boundary, a timed-out launch produced a warning and nothing else),
        the layer that opened the interval coming round AGAIN: one pass over the model's layers.  Pure bookkeeping (no device, no
        collective): every rank runs the same layers in the same order, so every rank sees the same boundaries."""
        if step is not None and step != self.step:
            self.step = step
        elif key is not None and key == self.first_key:
            pass                                             # (a full pass since the last boundary, the step number did not move)
        else:
            if self.first_key is None:
                self.first_key = key
            return False
        self.first_key = key
        return True

    def boundary(self, step, key=None) -> bool:
        """Called by every layer op before it runs; True once per agreement that some rank had a time-out."""
        if not self._new_interval(step, key):
            return False
        agreed = False
        if self.pending is not None:
            ev, host = self.pending
            ev.synchronize()         # (issued a whole step ago)
            agreed = bool(int(host[0]))
        if self._flag is None:
            self._flag = torch.zeros(1, dtype=torch.int32, device=f"cuda:{self.device}")
            self._host = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(2)]
        flag = self._flag
        flag.fill_(1 if self.local_error else 0)          # (a kernel on the current stream: no host-to-device copy to wait for)
        self.local_error = False
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        self._turn ^= 1
        host = self._host[self._turn]                       # (the other one is what the previous boundary's copy landed in)
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.pending = (ev, host)
        return agreed


class P2PArena:
    """Uncached IPC device memory of one (process group, device), mapped by every peer of the group (cfx_ipc_alloc / cfx_ipc_open): the
    packets and flag words of the group's p2p exchange layers.  Chunks are allocated collectively (every rank asks for the same
    regions in the same order: all ranks run the same layers); a region is keyed and handed out again for the same key, so
    rebuilding the layers (compact_reset, a new state arena) allocates nothing."""

    CHUNK = 64 << 20

    def __init__(self, group, rank: int, world: int, device: int, loopback: bool = False):
        self.group, self.rank, self.world, self.device, self.loopback = group, rank, world, device, loopback
        self.lib = _lib.load()
        self.ctx = codecs.context(device)
        self.chunks: List[Tuple[int, int, Dict[int, int]]] = []     # (own pointer, bytes, {peer rank: its mapping here})
        self.off = 0
        self.regions: Dict[Tuple, _Region] = {}
        self.health = None if loopback else GroupHealth(group, device)
        self.ok = True               # False: IPC unavailable, or a validation failed - the group's layers use another transport
        self.why = None
        self.kind = None             # cfx_ipc_memory_kind of the chunks: 2 uncached, 1 fine-grained, 0 ordinary device memory

    def _new_chunk(self, need: int) -> bool:
        size = max(self.CHUNK, (need + 4095) // 4096 * 4096)
        ptr, handle = ctypes.c_void_p(), ctypes.create_string_buffer(64)
        mine = bytes(handle.raw) if self.lib.cfx_ipc_alloc(self.ctx, size, ctypes.byref(ptr), handle) == 0 else None
        opened: Dict[int, int] = {}
        if self.loopback:
            good = mine is not None
            if good:
                opened = {r: ptr.value for r in range(self.world) if r != self.rank}      # every logical peer is this rank
        else:
            handles = [None] * self.world
            dist.all_gather_object(handles, mine, group=self.group)
            good = all(h is not None for h in handles)
            if good:
                for r in range(self.world):
                    if r == self.rank:
                        continue
                    pq = ctypes.c_void_p()
                    if self.lib.cfx_ipc_open(self.ctx, handles[r], ctypes.byref(pq)) != 0:
                        good = False
                        break
                    opened[r] = pq.value
            votes = [None] * self.world
            dist.all_gather_object(votes, bool(good), group=self.group)
            good = all(votes)
        if not good:
            if not self.loopback:
                for pq in opened.values():
                    self.lib.cfx_ipc_close(self.ctx, ctypes.c_void_p(pq))
            if mine is not None:
                self.lib.cfx_ipc_free(self.ctx, ptr)
            self.ok, self.why = False, "IPC-shared device memory is not available between the ranks of this group"
            return False
        kind = int(self.lib.cfx_ipc_memory_kind(self.ctx))
        if not self.loopback:
            kinds = [None] * self.world
            dist.all_gather_object(kinds, kind, group=self.group)
            kind = min(kinds)                                       # every rank takes the same decisions: the weakest kind of the group
        self.kind = kind if self.kind is None else min(self.kind, kind)
        self.chunks.append((ptr.value, size, opened))
        self.off = 0
        return True

    def region(self, key, slot_bytes: int) -> Optional[_Region]:
        """The region of `key` (allocated on first use; collective when a new chunk is needed), or None when the arena is unusable."""
        if not self.ok:
            return None
        reg = self.regions.get((key, slot_bytes))
        if reg is not None:
            return reg
        need = (4 * slot_bytes + 128 + 255) // 256 * 256
        if not self.chunks or self.off + need > self.chunks[-1][1]:
            if not self._new_chunk(need):
                return None
        own, _, opened = self.chunks[-1]
        reg = _Region(own + self.off, {r: p + self.off for r, p in opened.items()}, slot_bytes)
        self.off += need
        self.regions[(key, slot_bytes)] = reg
        return reg

    def mark_bad(self, why: str) -> None:
        self.ok, self.why = False, why

    def close(self, collective: bool = True) -> None:
        """Unmap the peers' chunks and free our own.  Nobody may still be reading: with `collective` the ranks meet before the
        unmapping and again before the freeing."""
        if not self.chunks:
            return
        torch.cuda.synchronize(self.device)
        meet = collective and not self.loopback and dist.is_available() and dist.is_initialized() and self.world > 1
        if meet:
            dist.barrier(group=self.group)
        if not self.loopback:
            for _, _, opened in self.chunks:
                for pq in opened.values():
                    self.lib.cfx_ipc_close(self.ctx, ctypes.c_void_p(pq))
        if meet:
            dist.barrier(group=self.group)
        for own, _, _ in self.chunks:
            self.lib.cfx_ipc_free(self.ctx, ctypes.c_void_p(own))
        self.chunks, self.regions, self.off = [], {}, 0


def _current_step():
    global _get_step
    if _get_step is None:
        from .main import compact_get_step
        _get_step = compact_get_step
    return _get_step()


_get_step = None
_arenas: Dict[Tuple, P2PArena] = {}
_loopback = False            # tests / one-GPU measurements: the W logical ranks of a group are all this process


def set_p2p_loopback(on: bool) -> None:
    """Stand-in for real peers on one GPU (tests, bench.py's plugin_path leg at N = 1): every logical peer's packets are this rank's
    own (the arena maps no peer), nothing is validated across ranks.  Drops the arenas."""
    global _loopback
    release()
    _loopback = bool(on)


def arena_for(group, rank: int, world: int, device: int) -> P2PArena:
    key = (id(group) if group is not None else None, device, world)
    a = _arenas.get(key)
    if a is None:
        a = _arenas[key] = P2PArena(group, rank, world, device, loopback=_loopback)
    return a


def release(collective: bool = True) -> None:
    """Free every arena (IPC mappings and allocations).  Layer ops built on them must not be run afterwards."""
    for a in _arenas.values():
        try:
            a.close(collective)
        except Exception:  # noqa: BLE001  (interpreter shutdown / a torn-down process group)
            pass
    _arenas.clear()
    _start_pools.clear()


# ---------------------------------------------------------------------------------------------------------------------------
# start matrices of the low-rank layer ops
# ---------------------------------------------------------------------------------------------------------------------------
class StartPool:
    """Start matrices (2, C, rp) fp32 for the subspace iterations of the low-rank layer ops of one shape, drawn a CHUNK at a time.

    The reference draws torch.randn per compress call (compress_lowrank.py:41): through the layer op that is one tiny launch in front
    of every layer's chain - 57 a FLUX step.  Every layer op owns one slot of a chunk of `CHUNK` slots for its lifetime (the plan ops
    hold the slot's address); an execution consumes its slot's draw, and the first execution that finds its slot consumed redraws the
    whole chunk with one normal_() - one launch per CHUNK layers and step, every execution still starts from its own fresh i.i.d.
    Gaussian matrix.  A draw is ordered with its consumers by the stream: a chunk is redrawn as a whole only while all its slots were
    last used on the drawing stream; a slot used from another stream draws for itself (what every slot did before round 6).
    No GPU needed: tests/test_host_logic.py drives the bookkeeping with CPU tensors."""
    CHUNK = 32

    def __init__(self, C: int, rp: int, rank: int, device):
        self.C, self.rp, self.r, self.device = C, rp, rank, device
        self.chunks: List[dict] = []
        self.free: List[Tuple[int, int]] = []
        self.draws = 0                          # normal_() launches so far (tests, tools)

    def acquire(self) -> Tuple[int, int, "torch.Tensor"]:
        if not self.free:
            t = torch.zeros(self.CHUNK, 2, self.C, self.rp, dtype=torch.float32, device=self.device)
            self.chunks.append({"t": t, "fresh": [False] * self.CHUNK, "last": [None] * self.CHUNK, "stream": None, "gen": 0})
            self.free = [(len(self.chunks) - 1, i) for i in reversed(range(self.CHUNK))]
        c, i = self.free.pop()
        ch = self.chunks[c]
        ch["last"][i] = None                    # (an unconsumed draw in the slot stays valid: nobody has read it; the padding columns are never written)
        return c, i, ch["t"][i]

    def give_back(self, c: int, i: int) -> None:
        self.chunks[c]["last"][i] = None
        self.free.append((c, i))

    def take(self, c: int, i: int, stream) -> None:
        """Slot (c, i) is about to be read by a chain enqueued on `stream` (any hashable; None on CPU): make sure it holds a draw
        nobody has consumed, enqueued on that stream."""
        ch = self.chunks[c]
        ch["last"][i] = stream
        if ch["fresh"][i] and ch["stream"] == stream:
            ch["fresh"][i] = False
            return
        if all(s is None or s == stream for s in ch["last"]):
            ch["t"][..., :self.r].normal_()
            ch["stream"], ch["fresh"] = stream, [True] * self.CHUNK
            ch["gen"] += 1                      # (a pinned matrix in another slot is gone: its owner pins again, see LayerOp._draw_start)
        else:
            ch["t"][i][..., :self.r].normal_()
        ch["fresh"][i] = False
        self.draws += 1

    def pin(self, c: int, i: int, q: "torch.Tensor") -> int:
        t = self.chunks[c]["t"][i]
        t.zero_()
        t[:, :, :self.r] = q.to(device=t.device, dtype=torch.float32)
        self.chunks[c]["fresh"][i] = False
        return self.chunks[c]["gen"]


_start_pools: Dict[Tuple, StartPool] = {}


def start_pool(device, C: int, rp: int, rank: int) -> StartPool:
    key = (str(device), C, rp, rank)
    sp = _start_pools.get(key)
    if sp is None:
        sp = _start_pools[key] = StartPool(C, rp, rank, device)
    return sp


# ---------------------------------------------------------------------------------------------------------------------------
# the layer op
# ---------------------------------------------------------------------------------------------------------------------------
_xstreams: Dict[Tuple[int, bool], "torch.cuda.Stream"] = {}


def _exchange_stream(device: int, beside_null_stream: bool):
    """ONE exchange stream per device for all layer ops (every stream is a hardware queue): it runs the flag kernels, which poll, so it
    must own its queue.  A CU-masked stream (full mask) is never pooled with other streams - but it is a BLOCKING stream, and the
    legacy NULL stream serialises with blocking streams: beside a model that runs on the default stream the exchange stream is a
    high-priority non-blocking one instead (high-priority streams do not share a queue with normal-priority ones)."""
    key = (device, beside_null_stream)
    s = _xstreams.get(key)
    if s is None:
        if beside_null_stream:
            s = torch.cuda.Stream(device, priority=-1)
        else:
            from .. import lanes
            s = lanes.dedicated_stream(device)
        _xstreams[key] = s
    return s


class LayerOp:
    """compress own K,V ; exchange the packets ; reconstruct every peer's K,V onto its state - one op of a native plan.

    own_states   [K, V] state buffers of this rank's shard (N, C) fp16 - the compress reads them (residual) and, per `own_update`,
                 the op updates them in place
    peer_states  [(rank, K state, V state), ...] in the order the consumer visits the peers
    own_update   "ef"     error feedback: own state <- own state + decode(own packet)   (ring.py: compact_compress(update_cache=True);
                          gather mode: the own shard is replaced by its reconstruction, main.py:406-419)
                 "x"      error feedback off in ring mode: own state <- the activation  (main.py:240-243)
    """

    def __init__(self, key, cid: int, param: int, N: int, C: int, rank: int, world: int, group, device: torch.device,
                 own_states: Sequence[torch.Tensor], peer_states: Sequence[Tuple[int, torch.Tensor, torch.Tensor]],
                 own_update: str = "ef"):
        assert own_update in ("ef", "x")
        assert 2 * len(peer_states) + 2 <= MAX_ITEMS, "a layer op carries at most CFX_MAX_BATCH reconstruction items"
        self.key, self.cid, self.param, self.N, self.C = key, int(cid), int(param), N, C
        self.rank, self.world, self.group, self.device = rank, world, group, device
        self.dev = device.index if device.index is not None else torch.cuda.current_device()
        self.own, self.peers, self.own_update = list(own_states), list(peer_states), own_update
        self.lib = _lib.load()
        self.ctx = codecs.context(self.dev)
        # low-rank family (LOW_RANK / LOW_RANK_Q, compact/lowrank.py ids 101 / 102; param = rank): the same op as a CHAIN of native plan ops -
        # factor chain of K,V (cfx_plan_add_lr_compress) ; the exchange (publish-and-wait kernel / collective) ; ONE batched reconstruction
        # of every peer tensor (cfx_plan_add_lr_decompress) - replayed by one host call.  Round 6: the generic path issued one compress
        # per tensor and one reconstruction per peer tensor from Python (16 launches and 0.44 ms of host time per FLUX layer).
        self.lowrank = self.cid >= 100
        self.quantized = self.cid == 102
        if self.lowrank:
            assert self.cid in (101, 102) and own_update == "ef", "the low-rank layer op exists with error feedback only"
            self.pkt_bytes = 2 * codecs.lr_packet_halves(self.quantized, N, C, self.param)
            rp = codecs.lr_rank_pad(self.param)
            self._pool = start_pool(device, C, rp, self.param)                         # start matrices of K and V: a fresh draw per execution
            self._slot = self._pool.acquire()
            self._q0 = self._slot[2]
            self._q0p = (ctypes.c_void_p * 2)(self._q0[0].data_ptr(), self._q0[1].data_ptr())
            self._pinned, self._pin_gen = None, -1
        else:
            self.pkt_bytes = codecs.packet_bytes(self.cid, N, C, self.param)
        self.slot = (self.pkt_bytes + 255) // 256 * 256
        self.flags = _lib.FLAG_UPDATE_CACHE | (0 if own_update == "ef" else _lib.FLAG_NO_EF)
        self.nops = 1                          # plan ops of one execution on the p2p / collective transports (low-rank chain: 3)
        self._xs = (ctypes.c_void_p * 2)()
        self._plans: Dict[int, Tuple] = {}       # run-stream handle -> (plan, keep-alive)
        self._run_x = self.lib.cfx_plan_run_x
        self._run = self.lib.cfx_plan_run
        self.transport = None
        self.fallback_reason = None
        self.region: Optional[_Region] = None
        self._comm = None
        self._recv = self._send = None
        self._choose_transport()

    # ---- transport -------------------------------------------------------------------------------------------------
    def _choose_transport(self, exclude: Sequence[str] = ()) -> None:
        pref = transport_pref()
        order = {"auto": ("p2p", "rccl", "torch"), "p2p": ("p2p", "rccl", "torch"), "rccl": ("rccl", "torch"), "torch": ("torch",)}[pref]
        self._drop_plans()
        self.region, self._comm, self._recv, self._send = None, None, None, None
        if self.world == 1 or not self.peers:
            self.transport = "none"
            return
        for t in order:
            if t in exclude:
                continue
            if t == "p2p":
                if self.world - 1 > 15:
                    continue
                arena = arena_for(self.group, self.rank, self.world, self.dev)
                reg = arena.region(self.key, self.slot)
                if reg is None:
                    continue
                self.region, self.arena = reg, arena
            elif t == "rccl":
                from ..exchange import native_comm_for
                comm = native_comm_for(self.group, self.dev)
                if comm is None:
                    continue
                self._comm = comm
                self._recv = torch.empty(self.world * 2 * self.slot, dtype=torch.uint8, device=self.device)
            else:
                self._send = torch.empty(2 * self.slot, dtype=torch.uint8, device=self.device)
                self._recv = torch.empty(self.world * 2 * self.slot, dtype=torch.uint8, device=self.device)
            self.transport = t
            return
        raise _lib.CfxError("no transport for the layer exchange")

    # ---- plans -----------------------------------------------------------------------------------------------------
    def _drop_plans(self) -> None:
        ln = getattr(self, "_lane", None)
        if ln is not None:
            self.lib.cfx_plan_destroy(ln["plan"])
            self._lane = None
        for plan, _ in getattr(self, "_plans", {}).values():
            self.lib.cfx_plan_destroy(plan)
        self._plans = {}

    def _drop_slot(self) -> None:
        if getattr(self, "_slot", None) is not None:
            self._pool.give_back(self._slot[0], self._slot[1])
            self._slot = None

    def close(self) -> None:
        self._drop_plans()
        self._drop_slot()

    def __del__(self):
        try:
            self._drop_plans()
            self._drop_slot()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def _check(self, ok, what: str) -> None:
        if not ok:
            raise _lib.CfxError(f"building the layer exchange op failed ({what}): " + (self.lib.cfx_last_error_string(self.ctx) or b"").decode())

    def _comp_items(self, pk_k: int, pk_v: int):
        o = self.own
        nb = [o[0].data_ptr(), o[1].data_ptr()]
        return (_lib.CompItem * 2)(_lib.CompItem(None, o[0].data_ptr(), nb[0], pk_k), _lib.CompItem(None, o[1].data_ptr(), nb[1], pk_v))

    def _build(self, sh: int):
        lib, ctx = self.lib, self.ctx
        cid, param, N, C = self.cid, self.param, self.N, self.C
        plan = lib.cfx_plan_create(ctx)
        self._check(plan, "cfx_plan_create")
        if self.lowrank:
            return self._build_lowrank(plan, sh)
        ws = codecs.workspace(cid, N, C, param, 2, self.dev, stream_handle=sh)
        wsp, wsn = (None, 0) if ws is None else (ws.data_ptr(), ws.numel())
        keep = [ws]
        t = self.transport
        if t == "rccl":          # (the peer-to-peer form runs its exchange inside the layer launch: no second stream)
            xs = _exchange_stream(self.dev, beside_null_stream=(sh == 0))
            keep.append(xs)
            self._check(lib.cfx_plan_use_exchange_stream(plan, xs.cuda_stream) == 0, "exchange stream")
        n_rec = 2 * len(self.peers)
        if t == "none":
            c = (_lib.CompItem * 2)(*[_lib.CompItem(None, s.data_ptr(), s.data_ptr(), p.data_ptr()) for s, p in zip(self.own, self._solo_packets())])
            self._check(lib.cfx_plan_add_compress(plan, cid, N, C, param, self.flags, 2, c, wsp, wsn) == 0, "compress")
        elif t == "p2p":
            reg = self.region
            for parity in (0, 1):
                c = self._comp_items(reg.packet(None, parity, 0), reg.packet(None, parity, 1))
                items = []
                for r, ks, vs in self.peers:
                    items.append(_lib.DecompItem(reg.packet(r, parity, 0), ks.data_ptr(), ks.data_ptr()))
                    items.append(_lib.DecompItem(reg.packet(r, parity, 1), vs.data_ptr(), vs.data_ptr()))
                live = sorted({r for r, _, _ in self.peers}) if not self.arena.loopback else []
                pf = (ctypes.c_void_p * max(1, len(live)))(*[reg.flag(r, parity) for r in live])
                op = lib.cfx_plan_add_exchange_layer_p2p(plan, cid, N, C, param, self.flags, 2, c, n_rec, (_lib.DecompItem * n_rec)(*items),
                                                         reg.flag(None, parity), len(live), pf, wsp, wsn)
                self._check(op == parity, "exchange layer (p2p)")
        elif t == "rccl":
            recv, slot = self._recv.data_ptr(), self.slot
            c = self._comp_items(recv + (2 * self.rank) * slot, recv + (2 * self.rank + 1) * slot)
            items = []
            for r, ks, vs in self.peers:
                items.append(_lib.DecompItem(recv + (2 * r) * slot, ks.data_ptr(), ks.data_ptr()))
                items.append(_lib.DecompItem(recv + (2 * r + 1) * slot, vs.data_ptr(), vs.data_ptr()))
            op = lib.cfx_plan_add_exchange_layer(plan, cid, N, C, param, self.flags, 2, c, n_rec, (_lib.DecompItem * n_rec)(*items),
                                                 self._comm.handle, recv + 2 * self.rank * slot, recv, 2 * slot, wsp, wsn)
            self._check(op == 0, "exchange layer (collective)")
        else:
            send, recv, slot = self._send.data_ptr(), self._recv.data_ptr(), self.slot
            c = self._comp_items(send, send + slot)
            self._check(lib.cfx_plan_add_compress(plan, cid, N, C, param, self.flags, 2, c, wsp, wsn) == 0, "compress")
            items = []
            for r, ks, vs in self.peers:
                items.append(_lib.DecompItem(recv + (2 * r) * slot, ks.data_ptr(), ks.data_ptr()))
                items.append(_lib.DecompItem(recv + (2 * r + 1) * slot, vs.data_ptr(), vs.data_ptr()))
            self._check(lib.cfx_plan_add_decompress(plan, cid, N, C, param, n_rec, (_lib.DecompItem * n_rec)(*items)) == 1, "reconstruct")
        self._check(lib.cfx_plan_finalize(plan) == 0, "finalize")
        ent = self._plans[sh] = (plan, keep)
        return ent

    def _build_lowrank(self, plan, sh: int):
        """The low-rank layer as plan ops: [factor chain of K,V -> packets, own state updated] ; [exchange] ; [batched reconstruction of
        the peers' K,V].  p2p: the packets sit in the arena region, the exchange is the publish-and-wait kernel (cfx_plan_add_p2p_sync),
        the reconstruction reads the peers' packets in place - three ops per parity, one host call per execution."""
        lib, q, N, C, rank = self.lib, int(self.quantized), self.N, self.C, self.param
        n_rec = 2 * len(self.peers)
        wsn = int(lib.cfx_lr_workspace_bytes(q, N, C, rank, max(2, n_rec)))
        self._check(wsn > 0, "low-rank workspace")
        ws = torch.empty(wsn, dtype=torch.uint8, device=self.device)
        keep = [ws, self._q0]
        wsp = ws.data_ptr()
        t = self.transport

        def comp(pk_k, pk_v):
            c = self._comp_items(pk_k, pk_v)
            keep.append(c)
            return lib.cfx_plan_add_lr_compress(plan, q, N, C, rank, self.flags, 2, c, self._q0p, wsp, wsn)

        def recon(packet_of):
            items = []
            for r, ks, vs in self.peers:
                items.append(_lib.DecompItem(packet_of(r, 0), ks.data_ptr(), ks.data_ptr()))
                items.append(_lib.DecompItem(packet_of(r, 1), vs.data_ptr(), vs.data_ptr()))
            arr = (_lib.DecompItem * n_rec)(*items)
            keep.append(arr)
            return lib.cfx_plan_add_lr_decompress(plan, q, N, C, rank, n_rec, arr, wsp, wsn)
        if t == "none":
            solo = self._solo_packets()
            self._check(comp(solo[0].data_ptr(), solo[1].data_ptr()) == 0, "low-rank compress")
            self.nops = 1
        elif t == "p2p":
            reg = self.region
            live = sorted({r for r, _, _ in self.peers}) if not self.arena.loopback else []
            for parity in (0, 1):
                self._check(comp(reg.packet(None, parity, 0), reg.packet(None, parity, 1)) == 3 * parity, "low-rank compress")
                pf = (ctypes.c_void_p * max(1, len(live)))(*[reg.flag(r, parity) for r in live])
                keep.append(pf)
                self._check(lib.cfx_plan_add_p2p_sync(plan, reg.flag(None, parity), len(live), pf) == 3 * parity + 1, "p2p sync")
                self._check(recon(lambda r, kv, parity=parity: reg.packet(r, parity, kv)) == 3 * parity + 2, "low-rank reconstruct")
            self.nops = 3
        elif t == "rccl":
            recv, slot = self._recv.data_ptr(), self.slot
            self._check(comp(recv + (2 * self.rank) * slot, recv + (2 * self.rank + 1) * slot) == 0, "low-rank compress")
            self._check(lib.cfx_plan_add_all_gather(plan, self._comm.handle, recv + 2 * self.rank * slot, recv, 2 * slot) == 1, "all-gather")
            self._check(recon(lambda r, kv: recv + (2 * r + kv) * slot) == 2, "low-rank reconstruct")
            self.nops = 3
        else:
            send, recv, slot = self._send.data_ptr(), self._recv.data_ptr(), self.slot
            self._check(comp(send, send + slot) == 0, "low-rank compress")
            self._check(recon(lambda r, kv: recv + (2 * r + kv) * slot) == 1, "low-rank reconstruct")
            self.nops = 1                      # (the collective is issued from Python between the two ops)
        self._check(lib.cfx_plan_finalize(plan) == 0, "finalize")
        ent = self._plans[sh] = (plan, keep)
        return ent

    def _draw_start(self, sh: int) -> None:
        """The start matrices of this execution's subspace iterations: a fresh Gaussian draw per call as the reference makes
        (compress_lowrank.py:41) - from the shape's StartPool, one launch per 32 layer executions - or the matrix a test pinned
        (lowrank.set_init_q)."""
        from . import lowrank
        pinned = lowrank._pinned_q
        if pinned is None:
            self._pool.take(self._slot[0], self._slot[1], sh)
            self._pinned = None
        elif pinned is not self._pinned or self._pool.chunks[self._slot[0]]["gen"] != self._pin_gen:
            assert tuple(pinned.shape) == (self.C, self.param), f"pinned init_q must be ({self.C}, {self.param})"
            self._pin_gen = self._pool.pin(self._slot[0], self._slot[1], pinned)
            self._pinned = pinned

    def _solo_packets(self):
        if getattr(self, "_solo", None) is None:
            self._solo = [torch.empty(self.slot, dtype=torch.uint8, device=self.device) for _ in range(2)]
        return self._solo

    # ---- run -------------------------------------------------------------------------------------------------------
    def run(self, k: torch.Tensor, v: torch.Tensor, sh: int, lane: bool = False) -> Optional[int]:
        """The layer's exchange on the stream with handle `sh` (the caller's current stream): ONE native call.
        lane = True (low-rank family on the peer-to-peer transport, the caller on the exchange lane's compute stream): only the factor chain
        runs on `sh`; the publish-and-wait and the reconstructions of the peers are left to `lane_chain()`, which puts them on the exchange
        lane behind a flag published here - the epoch the caller's merge launches wait for is returned.  None: everything ran in stream
        order on `sh` (a validated execution, another transport), nothing is left to wait for."""
        ent = self._plans.get(sh)
        if ent is None:
            ent = self._build(sh)
        t = self.transport
        if t == "p2p":
            if not self.arena.ok:              # another layer of the group failed its validation: every layer leaves p2p (all ranks alike)
                # ... and takes its peers' copies from their owners on the way out, whether or not it owed the group a proof: the failed
                # validation may have been the FIRST native call after this layer's own steady launch gave up waiting (the context's error
                # word makes that call return CFX_ERR_GATE without launching and is cleared by it) - this layer then stored nothing, nobody
                # set its `recheck`, and under error feedback its copies would stay a delta behind for good.  One all-gather of the layer's
                # K,V per layer, once, on a path that has already failed.
                if not self.arena.loopback:
                    self._resync(self.own)
                self.region.recheck = 0
                self.fallback_reason = self.arena.why
                self._choose_transport(exclude=("p2p",))
                return self.run(k, v, sh, lane)
            reg = self.region
            if not self.arena.loopback:
                if self.arena.health.boundary(_current_step(), self.key):
                    # some rank's wait timed out a step or two ago (every rank reads the same agreement at the same boundary): the launch
                    # it belonged to stored nothing, so its states are a delta behind their owners' - every layer proves itself again
                    self.lib.cfx_gate_recover(self.ctx)
                    for r_ in self.arena.regions.values():
                        r_.recheck = 1
                if reg.validated < VALIDATE_FIRST or reg.recheck or (self.arena.kind != 2 and reg.n_exec % REVALIDATE_EVERY == 0):
                    return self._run_validated(k, v, sh)
            op = (reg.n_exec & 1) * self.nops
            reg.n_exec += 1
        else:
            op = 0
        if self.lowrank:
            self._draw_start(sh)
        xs = self._xs
        xs[0], xs[1] = k.data_ptr(), v.data_ptr()
        split = lane and self.lowrank and t == "p2p" and self.nops == 3
        nops = 1 if split else self.nops          # (lane: the factor chain only - the publish-and-wait opens the lane's chain instead)
        rc = self._run_x(ent[0], op, nops, xs, 2, sh)
        if rc == _lib.CFX_ERR_GATE and t == "p2p" and not self.arena.loopback:
            # an EARLIER launch's wait gave up (a peer later than the gate timeout): it stored nothing.  Clear the word, tell the group at
            # the next step boundary, and issue this layer's launch - the flag epochs have to keep step with the peers'
            if self.lib.cfx_gate_errors(self.ctx) > 0:
                self.arena.health.note_error()
                self.lib.cfx_gate_recover(self.ctx)      # local counters only: without it every launch of this ring slot would sit out the whole
                #                                          timeout on the short arrival count until the group's boundary a step or two later
                warnings.warn("compactfusion_amd: an in-launch wait of the peer-to-peer exchange timed out (a peer was later than the gate "
                              "timeout); nothing was stored, the group validates its layers again at the next step boundary")
            rc = self._run_x(ent[0], op, nops, xs, 2, sh)
        if rc == 0 and t == "torch":
            dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
            rc = self._run(ent[0], 1, 1, sh)
        if rc != 0:
            raise _lib.CfxError("the layer exchange op failed: " + (self.lib.cfx_last_error_string(self.ctx) or b"").decode())
        return self._lane_begin(op // self.nops, sh) if split else None

    # ---- the low-rank layer beside the attention blocks (protocol 2) ---------------------------------------------------------------
    # The factor chain is one persistent launch that wants the chip: it stays on the caller's (compute-lane) stream, exposed.  The
    # reconstruction of the 7 peers' K,V - 14 tensors read and written, as long as the chain itself at rank 8 - needs nothing but the
    # packets: it runs on the exchange lane, peer by peer in the order the attention blocks visit them, each peer's flag waited for
    # inside the merge launch of the block in front of it (the streaming codecs' lane, compact/ring.py, with the compress left where it is).
    def lane_capable(self) -> bool:
        return self.lowrank and self.transport == "p2p" and self.nops == 3 and _settings.get("lowrank_lane") == "on"

    def _build_lane(self):
        from .. import lanes
        lib, q, N, C, rank = self.lib, int(self.quantized), self.N, self.C, self.param
        plan = lib.cfx_plan_create(self.ctx)
        self._check(bool(plan), "lane plan")
        xstream = lanes.exchange_stream(self.dev)
        self._check(lib.cfx_plan_use_exchange_stream(plan, xstream.cuda_stream) == 0, "lane stream")
        n_peers = len(self.peers)
        flags = lib.cfx_plan_flags(plan, n_peers + 1)            # 0: own packets complete (compute lane) ; s = 1 .. W-1: peer s reconstructed
        self._check(bool(flags), "lane flags")
        wsn = int(lib.cfx_lr_workspace_bytes(q, N, C, rank, 2))
        ws = torch.empty(max(wsn, 16), dtype=torch.uint8, device=self.device)
        keep = [ws, xstream]
        reg = self.region
        per = 2 + 2 * n_peers
        live = sorted({r for r, _, _ in self.peers}) if not self.arena.loopback else []
        for parity in (0, 1):
            self._check(lib.cfx_plan_add_flag_wait(plan, 0) == parity * per, "lane wait")
            pf = (ctypes.c_void_p * max(1, len(live)))(*[reg.flag(r, parity) for r in live])
            keep.append(pf)
            self._check(lib.cfx_plan_add_p2p_sync(plan, reg.flag(None, parity), len(live), pf) >= 0, "lane p2p sync")
            for s_, (r, ks, vs) in enumerate(self.peers, start=1):
                arr = (_lib.DecompItem * 2)(_lib.DecompItem(reg.packet(r, parity, 0), ks.data_ptr(), ks.data_ptr()),
                                            _lib.DecompItem(reg.packet(r, parity, 1), vs.data_ptr(), vs.data_ptr()))
                keep.append(arr)
                self._check(lib.cfx_plan_add_lr_decompress(plan, q, N, C, rank, 2, arr, ws.data_ptr(), wsn) >= 0, "lane reconstruct")
                self._check(lib.cfx_plan_add_flag_set(plan, s_) >= 0, "lane flag")
        self._check(lib.cfx_plan_finalize(plan) == 0, "lane finalize")
        self._lane = {"plan": plan, "keep": keep, "flags": int(flags), "per": per, "epoch": ctypes.c_uint(0), "first": 0}
        return self._lane

    def _lane_begin(self, parity: int, sh: int) -> int:
        ln = getattr(self, "_lane", None) or self._build_lane()
        ln["first"] = parity * ln["per"]
        if self.lib.cfx_plan_lane_begin(ln["plan"], 0, sh, ln["epoch"]) != 0:
            raise _lib.CfxError("the low-rank layer's lane hand-over failed: " + (self.lib.cfx_last_error_string(self.ctx) or b"").decode())
        return ln["epoch"].value

    def lane_chain(self, first_peer: int = 1, n_peers: Optional[int] = None) -> None:
        """After `run(..., lane=True)` returned an epoch: the reconstructions of peers first_peer .. first_peer + n_peers - 1 (in the order
        the attention blocks visit them, 1-based; default: all) onto the exchange lane.  The caller issues them a few at a time between
        its attention blocks: the chain is 15 launches at W = 8, and a compute queue that has run dry behind the local block while the
        host is still issuing them is exposed time (measured: 120 us per layer with the whole chain in one go)."""
        ln = self._lane
        total = len(self.peers)
        if n_peers is None:
            n_peers = total - first_peer + 1
        n_peers = min(n_peers, total - first_peer + 1)
        if n_peers <= 0:
            return
        first = ln["first"] + (0 if first_peer == 1 else 2 + 2 * (first_peer - 1))
        count = 2 * n_peers + (2 if first_peer == 1 else 0)
        if self.lib.cfx_plan_run_lane(ln["plan"], first, count, None, 0, 0, None, ln["epoch"]) != 0:
            raise _lib.CfxError("the low-rank layer's lane chain failed: " + (self.lib.cfx_last_error_string(self.ctx) or b"").decode())

    def lane_flag(self, i: int) -> int:
        return self._lane["flags"] + 64 * i

    def _checksums(self, tensors: Sequence[torch.Tensor]) -> torch.Tensor:
        return torch.stack([t.view(torch.int32).sum(dtype=torch.int64) for t in tensors])

    def _packet_view(self, r: Optional[int], parity: int, kv: int, n_half: int) -> torch.Tensor:
        """The packet of rank r (None: this rank's own) as a tensor over the arena / the IPC mapping (an even number of halves: int32 sums)."""
        from .ring import _raw_halves
        return _raw_halves(self.region.packet(r, parity, kv), n_half & ~1, self.device)

    def _resync(self, snap: Sequence[torch.Tensor]) -> None:
        """After a failed validation: every rank's copy of a shard's state <- its owner's state (as it was before the failed execution).
        A launch whose wait timed out stored nothing, so a rank may be a delta behind; the owner's state is what the next residual is
        taken against and therefore the truth.  (Without error feedback the owner keeps the activation instead of what its peers
        hold - there is no authoritative copy to fetch, the restored states are all there is.)"""
        if self.own_update != "ef":
            return
        n = snap[0].numel() // 2                                                              # fp16 pairs as int32: a dtype every backend moves
        own = torch.cat([snap[0].reshape(-1).view(torch.int32), snap[1].reshape(-1).view(torch.int32)])
        every = torch.empty(self.world * 2 * n, dtype=torch.int32, device=own.device)
        dist.all_gather_into_tensor(every, own, group=self.group)
        every = every.view(self.world, 2, n)
        for r, ks, vs in self.peers:
            ks.view(torch.int32).reshape(-1).copy_(every[r, 0])
            vs.view(torch.int32).reshape(-1).copy_(every[r, 1])

    def _run_validated(self, k, v, sh) -> None:
        """A validated p2p execution of this layer: run it, then check that no gate timed out and that what this rank reconstructed for
        every peer's shard IS what that shard's owner says its peers must hold (checksums over the group): the owner's own state with
        error feedback, the owner's previous state + its decoded packet without (own_update "x": the owner keeps the activation,
        main.py:233).  All ranks decide together; on a failure the layer's states are restored, the peers' copies are re-synchronised
        from their owners, the group's arena is marked bad and the execution is repeated on the next transport."""
        reg, lib, ctx = self.region, self.lib, self.ctx
        states = list(self.own) + [t for _, ks, vs in self.peers for t in (ks, vs)]
        snap = [t.clone() for t in states]
        ent = self._plans[sh]
        op = reg.n_exec & 1
        reg.n_exec += 1
        if self.lowrank:
            self._draw_start(sh)
        self._xs[0], self._xs[1] = k.data_ptr(), v.data_ptr()
        rc = self._run_x(ent[0], op * self.nops, self.nops, self._xs, 2, sh)
        stream = torch.cuda.current_stream(self.device)
        stream.synchronize()
        bad = 1 if rc != 0 else 0
        if lib.cfx_gate_errors(ctx) != 0:      # (also the word an EARLIER layer's steady launch left: rc is CFX_ERR_GATE then and nothing was launched)
            bad = 1
            self.arena.health.note_error()
            lib.cfx_gate_recover(ctx)          # local counters only: the launches that follow must not sit out the timeout on a short count
        if self.own_update == "ef":
            # error feedback: what every peer reconstructed of a shard IS its owner's state
            mine = self._checksums(self.own)
            allc = torch.empty(self.world * 2, dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(allc, mine, group=self.group)
            for r, ks, vs in self.peers:
                if not torch.equal(self._checksums([ks, vs]), allc[2 * r:2 * r + 2]):
                    bad = 1
        else:
            # without error feedback the owner keeps the activation and only its peers hold previous copy + decoded packet (main.py:233):
            # there is no owner's state to compare with.  Checked instead: (1) the packet AS THIS RANK READS IT through its mapping is the
            # packet its owner wrote (checksums over the group), and (2) what the launch made of it is what the receiver's kernel makes of
            # that packet and this rank's previous copy when run again now, in stream order
            n_half = self.pkt_bytes // 2
            mine = self._checksums([self._packet_view(None, op, kv, n_half) for kv in (0, 1)])
            allc = torch.empty(self.world * 2, dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(allc, mine, group=self.group)
            for i, (r, ks, vs) in enumerate(self.peers):
                seen = self._checksums([self._packet_view(r, op, kv, n_half) for kv in (0, 1)])
                if not torch.equal(seen, allc[2 * r:2 * r + 2]):
                    bad = 1
                for kv, got in ((0, ks), (1, vs)):
                    again = torch.empty_like(got)
                    if lib.cfx_decompress(ctx, self.cid, reg.packet(r, op, kv), snap[2 + 2 * i + kv].data_ptr(), again.data_ptr(), self.N, self.C,
                                          self.param, sh) != 0:
                        bad = 1
                    stream.synchronize()
                    if not torch.equal(again.view(torch.int16), got.view(torch.int16)):
                        bad = 1
        verdict = torch.tensor([bad], dtype=torch.int32, device=self.device)
        dist.all_reduce(verdict, op=dist.ReduceOp.MAX, group=self.group)
        if int(verdict.item()) == 0:
            reg.validated += 1
            reg.recheck = 0
            return
        why = ("the peer-to-peer exchange failed its validation (execution %d of layer %r: gate time-out or a reconstruction that differs "
               "from its owner's state); this group's layers continue on the next transport" % (reg.validated + 1, self.key))
        if self.rank == 0:
            warnings.warn("compactfusion_amd: " + why)
        for t, s in zip(states, snap):
            t.copy_(s)
        self._resync(snap)
        reg.recheck = 0
        self.arena.mark_bad(why)
        self.fallback_reason = why
        self._choose_transport(exclude=("p2p",))
        self.run(k, v, sh)


def usable(cid: int, world: int, is_cuda: bool, ef: bool = True) -> bool:
    """Can the layer's exchange run as a LayerOp: a native streaming codec - or the low-rank family with error feedback -, W ranks whose
    2 (W - 1) peer tensors + own K,V fit one batch."""
    return is_cuda and (1 <= cid <= 5 or (cid in (101, 102) and ef)) and 2 * (world - 1) + 2 <= MAX_ITEMS
