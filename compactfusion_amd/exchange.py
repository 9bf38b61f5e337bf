"""RCCL communicator owned by libcfx.so, bootstrapped through torch.distributed.

The per-layer exchange of the compressed path is an all-gather of ~0.2-0.4 MB per rank; driving it through
`torch.distributed` costs ~50 us of host time per call (measured on MI355X: 3.3 ms vs 1.8 ms per 57-layer step), more
than the GPU work it overlaps.  `NativeComm` lets a `cfx_plan` issue the collective itself (ncclAllGather on the plan's
side HIP stream, ordered with events), so a whole pipelined step is ONE native call.  The RCCL library instance is the
one PyTorch-ROCm already loaded (found in /proc/self/maps), the 128-byte unique id travels over the existing process
group."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib
from .codecs import context


def loaded_rccl_path() -> Optional[str]:
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    return line.split()[-1]
    except OSError:
        pass
    return None


class NativeComm:
    def __init__(self, device: int, group=None, library: Optional[str] = None, solo_ranks: int = 0):
        """library: path of the collective library to load instead of the RCCL the process already uses (tests / bench debug:
        tests/fake_rccl).  solo_ranks > 0: a communicator that needs no torch.distributed - this process is rank 0 of
        `solo_ranks` (1 with real RCCL: a one-rank communicator; > 1 only with a loop-back stand-in library)."""
        self.lib = _lib.load()
        self.ctx = context(device)
        solo = solo_ranks > 0
        self.rank = 0 if solo else dist.get_rank(group)
        self.world = solo_ranks if solo else dist.get_world_size(group)
        path = library or loaded_rccl_path()
        # Everything that can fail on ONE rank alone (loading the library, making the unique id) happens before the first collective of
        # this bootstrap, and the ranks vote on it: a rank that failed here used to leave its peers inside broadcast_object_list /
        # ncclCommInitRank while it was already in the caller's own all-or-none vote - mismatched collectives, a hang instead of a fall-back.
        err = None
        if self.lib.cfx_rccl_load(path.encode() if path else None) != 0:
            err = "cannot load RCCL (librccl.so) for the native exchange"
        uid = ctypes.create_string_buffer(128)
        if err is None and self.rank == 0 and self.lib.cfx_comm_unique_id(self.ctx, uid) != 0:
            err = "ncclGetUniqueId failed"
        if not solo and self.world > 1:
            votes = [None] * self.world
            dist.all_gather_object(votes, err is None, group=group)
            if err is None and not all(votes):
                err = "a peer could not load RCCL / make the unique id"
        if err is not None:
            raise _lib.CfxError(err)
        box = [bytes(uid.raw)]
        if not solo:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._uid = ctypes.create_string_buffer(box[0], 128)
        torch.cuda.synchronize(device)
        self.handle = self.lib.cfx_comm_create(self.ctx, self._uid, self.world, self.rank)
        if not self.handle:
            msg = self.lib.cfx_last_error_string(self.ctx)
            raise _lib.CfxError("ncclCommInitRank failed: %s" % (msg.decode() if msg else ""))
        self.device = device

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> None:
        nbytes = send.numel() * send.element_size()
        assert recv.numel() * recv.element_size() == nbytes * self.world
        s = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        rc = self.lib.cfx_comm_all_gather(self.handle, send.data_ptr(), recv.data_ptr(), nbytes, s)
        if rc != 0:
            raise _lib.CfxError("native all-gather failed")

    def self_test(self) -> None:
        """Gather a rank-stamped pattern and check every slot (run once before trusting the communicator)."""
        dev = torch.device("cuda", self.device)
        send = torch.full((4096,), self.rank + 1, dtype=torch.uint8, device=dev)
        recv = torch.zeros(4096 * self.world, dtype=torch.uint8, device=dev)
        self.all_gather(send, recv)
        torch.cuda.synchronize(self.device)
        want = torch.arange(1, self.world + 1, dtype=torch.uint8, device=dev).repeat_interleave(4096)
        if not torch.equal(recv, want):
            raise _lib.CfxError("native all-gather self-test failed")

    def close(self):
        if self.handle:
            self.lib.cfx_comm_destroy(self.handle)
            self.handle = None


# ---- one communicator per (process group, device), created on first use by every rank of the group together -----------------
_comms = {}
_factory = None          # tests / measurement tools: callable(group, device) -> object with .handle (e.g. a fake-RCCL communicator)
_failed = set()


def set_comm_factory(fn) -> None:
    """Replace how `native_comm_for` obtains a communicator (None restores the RCCL bootstrap).  Drops cached communicators."""
    global _factory
    _factory = fn
    _comms.clear()
    _failed.clear()


def native_comm_for(group, device: int):
    """The library-owned communicator of `group` on `device`, or None when it cannot be created (the caller then keeps the
    collective in torch.distributed - and says so once)."""
    key = (id(group) if group is not None else None, device)
    c = _comms.get(key)
    if c is not None or key in _failed:
        return c
    try:
        c = _factory(group, device) if _factory is not None else NativeComm(device, group)
        if _factory is None:
            c.self_test()
        _comms[key] = c
    except Exception as e:  # noqa: BLE001
        import warnings
        warnings.warn(f"compactfusion_amd: native exchange unavailable for this group ({e}); the per-layer collective stays in "
                      "torch.distributed (about 50 us of host time per call)")
        _failed.add(key)
        c = None
    # every rank of the group takes the same path: a rank that failed would issue torch.distributed's all-gather while its peers sit in
    # the library's - all or none (the stand-in factories of the tests are per-process by design and skip the vote)
    if _factory is None and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        votes = [None] * dist.get_world_size(group)
        dist.all_gather_object(votes, c is not None, group=group)
        if not all(votes) and c is not None:
            import warnings
            warnings.warn("compactfusion_amd: a peer could not create the native communicator; this group keeps its collective in torch.distributed")
            try:
                c.close()
            except Exception:  # noqa: BLE001
                pass
            _comms.pop(key, None)
            _failed.add(key)
            c = None
    return c
