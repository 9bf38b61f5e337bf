"""Exchange lane: two CU-masked HIP streams per device with disjoint CU sets.

The reference overlaps the K,V exchange with the local attention block on NCCL's own stream and decompresses on the compute
stream (xfuser/compact/ring.py:191-269).  Here the layer's whole chain - compress, collective, per-peer reconstruction - runs
on an EXCHANGE stream restricted to a few CUs of every XCD, and the model (attention, projections) on a COMPUTE stream
restricted to the rest, so that the chain's workgroups never share a CU with an attention workgroup (measured on MI355X,
tools/sdpa_mask_probe.py: one SDPA block at the FLUX ring shape 37 us alone, 38 us beside a saturating copy on 32 other CUs,
93 us when the two share CUs; 224 CUs are as fast as 256 for it, 192 are not).  The two streams are ordered only through flag
words in device memory (`cfx_plan_run_lane`, `cfx_attn_merge_wait`; include/cfx.h).

Use:
    from compactfusion_amd import lanes
    with torch.cuda.stream(lanes.compute_stream()):        # the model runs on the lane's compute stream
        ... pipeline(...) ...
`compact_fwd` takes the masked exchange stream whenever the current stream is the lane's compute stream; on any other stream
it keeps an unmasked exchange stream (same flags, no CU partition).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from . import _lib
from . import config as _settings
from .codecs import context


class Lane:
    def __init__(self, device: int, exchange_cus: int):
        lib = _lib.load()
        self.device = device
        self.ctx = context(device)
        total = torch.cuda.get_device_properties(device).multi_processor_count
        if not (8 <= exchange_cus <= total - 8):
            raise ValueError(f"exchange CUs must be between 8 and {total - 8}")
        self.exchange_cus, self.compute_cus = exchange_cus, total - exchange_cus
        ex, co = ctypes.c_void_p(), ctypes.c_void_p()
        for (first, n, out) in ((0, exchange_cus, ex), (exchange_cus, total - exchange_cus, co)):
            rc = lib.cfx_stream_create_masked(self.ctx, first, n, ctypes.byref(out))
            if rc != 0:
                raise _lib.CfxError("cannot create a CU-masked stream: " + (lib.cfx_last_error_string(self.ctx) or b"").decode())
        self.exchange = torch.cuda.ExternalStream(ex.value, device=torch.device("cuda", device))
        self.compute = torch.cuda.ExternalStream(co.value, device=torch.device("cuda", device))


_lanes: Dict[int, Lane] = {}
_dedicated: Dict[int, "torch.cuda.Stream"] = {}


def dedicated_stream(device: Optional[int] = None) -> "torch.cuda.Stream":
    """An exchange stream over ALL CUs that owns its hardware queue.  Streams ordered by flags (one polls what the other sets) must not
    share a hardware queue - the polling kernel would block the kernel it waits for until the wait times out - and HIP multiplexes
    ordinary streams (hipStreamCreate, torch.cuda.Stream) over a small pool of queues.  A stream created with a CU mask is never
    pooled, so this one is made by cfx_stream_create_masked with the full mask."""
    if device is None:
        device = torch.cuda.current_device()
    s = _dedicated.get(device)
    if s is None:
        lib = _lib.load()
        total = torch.cuda.get_device_properties(device).multi_processor_count
        h = ctypes.c_void_p()
        ctx = context(device)
        if lib.cfx_stream_create_masked(ctx, 0, total, ctypes.byref(h)) != 0:
            raise _lib.CfxError("cannot create the exchange stream: " + (lib.cfx_last_error_string(ctx) or b"").decode())
        s = _dedicated[device] = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", device))
    return s


def lane(device: Optional[int] = None) -> Lane:
    if device is None:
        device = torch.cuda.current_device()
    ln = _lanes.get(device)
    if ln is None:
        ln = _lanes[device] = Lane(device, int(_settings.get("lane_exchange_cus")))
    return ln


def compute_stream(device: Optional[int] = None) -> "torch.cuda.Stream":
    """The stream to run the model on (CU-masked to everything the exchange lane does not use)."""
    return lane(device).compute


def exchange_stream(device: Optional[int] = None) -> "torch.cuda.Stream":
    return lane(device).exchange


def on_compute_stream(device: int) -> bool:
    """True when the CURRENT stream of `device` is the lane's compute stream (a lane exists and the caller opted in)."""
    ln = _lanes.get(device)
    return ln is not None and torch.cuda.current_stream(device).cuda_stream == ln.compute.cuda_stream


# ---------------------------------------------------------------------------------------------------------------------------
# getting onto the lane without the caller's help (compactfusion_amd.configure(lane="auto" | "sticky" | "off"))
# ---------------------------------------------------------------------------------------------------------------------------
_handover: Dict[int, dict] = {}


def _hand(device: int) -> dict:
    h = _handover.get(device)
    if h is None:
        buf = torch.zeros(64, dtype=torch.int32, device=torch.device("cuda", device))      # two flag words, 128 bytes apart
        torch.cuda.synchronize(device)                                                      # zeroed before a flag kernel of another stream reads them
        h = _handover[device] = {"buf": buf, "fork": buf.data_ptr(), "join": buf.data_ptr() + 128, "epoch": 0,
                                 "lib": _lib.load(), "ctx": context(device)}
    return h


def usable(device: int) -> bool:
    """Can this process order streams by flag words at all (hardware queues: include/cfx.h cfx_hw_queues_ok)?"""
    return bool(_lib.load().cfx_hw_queues_ok())


def fork_to_compute(device: int, begin=None, flag=None):
    """The caller's current stream hands over to the lane's compute stream: `flag set` behind everything the caller has enqueued, `flag
    wait` in front of everything that follows on the compute stream - two one-wave kernels, ~3 us of hop (an event pair costs ~14) -
    and the compute stream becomes the thread's current stream.  Returns the token `join_from_compute` takes, or None when the caller is
    on the compute stream already.  `begin(stream_handle) -> epoch` / `flag(epoch) -> address`: the set kernel is somebody else's
    publication (a layer plan's "K,V exist": cfx_plan_lane_begin launched on the CALLER's stream) and the compute stream waits for that
    word - one kernel fewer in front of the local attention block; the token then carries the epoch as a third element.
    Tensor lifetimes need no record_stream: every use on the compute stream lies between a fork and a join (or, sticky, stays there), so
    whatever the caching allocator hands out again on either stream is ordered behind its last use."""
    ln = lane(device)
    cur = torch.cuda.current_stream(device)
    if cur.cuda_stream == ln.compute.cuda_stream:
        return None
    h = _hand(device)
    lib, ctx = h["lib"], h["ctx"]
    h["epoch"] += 1
    je = h["epoch"]                              # the JOIN's epoch is always the hand-over's own
    if begin is not None:
        e = begin(cur.cuda_stream)
        rc = lib.cfx_flag_wait(ctx, flag(e), e, ln.compute.cuda_stream)
    else:
        e = None
        rc = lib.cfx_flag_set(ctx, h["fork"], je, cur.cuda_stream) or lib.cfx_flag_wait(ctx, h["fork"], je, ln.compute.cuda_stream)
    if rc != 0:
        raise _lib.CfxError("exchange lane hand-over failed: " + (lib.cfx_last_error_string(ctx) or b"").decode())
    torch.cuda.set_stream(ln.compute)
    return (cur, je, e)


def join_from_compute(device: int, token) -> None:
    """The reverse hand-over: the caller's stream continues behind everything enqueued on the compute stream since the fork."""
    if token is None:
        return
    cur, e = token[0], token[1]
    ln = lane(device)
    h = _hand(device)
    lib, ctx = h["lib"], h["ctx"]
    if lib.cfx_flag_set(ctx, h["join"], e, ln.compute.cuda_stream) != 0 or lib.cfx_flag_wait(ctx, h["join"], e, cur.cuda_stream) != 0:
        raise _lib.CfxError("exchange lane hand-over failed: " + (lib.cfx_last_error_string(ctx) or b"").decode())
    torch.cuda.set_stream(cur)
