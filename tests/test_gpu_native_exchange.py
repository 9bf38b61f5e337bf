"""The library-owned exchange (cfx_comm_* / cfx_plan_add_all_gather / cfx_plan_add_ring_hop, every exchange-stream mode, the
in-order and the software-pipelined replay) with MORE THAN ONE RANK on one GPU: the ranks are host threads of this process and
the collective library is tests/fake_rccl (a stand-in for the RCCL entry points libcfx.so resolves at run time: same
stream-ordered semantics, device-to-device copies instead of xGMI).  Every rank's state is checked against the oracle:
what a rank holds for its own shard, and what every peer reconstructed for that shard, must be the oracle's error-feedback
state bit for bit (reference flow: xfuser/compact/ring.py:188-269, main.py:406-419)."""
import os
import sys
import threading

import numpy as np
import pytest
import torch

from oracle import ref_np as R

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _fake_path():
    sys.path.insert(0, os.path.join(HERE, "fake_rccl"))
    try:
        import build as fake_build
        return fake_build.build()
    finally:
        sys.path.pop(0)
        sys.modules.pop("build", None)


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def dev16(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.float16).cuda()


class Ranks:
    """W ranks as threads: every rank gets its own cfx_ctx, stream and communicator of one fake-RCCL group."""

    def __init__(self, W):
        from compactfusion_amd import _lib
        self.lib = _lib.load()
        self.W = W
        os.environ["CFX_FAKE_RCCL_MODE"] = "threads"
        assert self.lib.cfx_rccl_load(_fake_path().encode()) == 0
        self.ctx = [self.lib.cfx_create(0) for _ in range(W)]
        for c in self.ctx:
            assert self.lib.cfx_prepare(c) == 0
        import ctypes
        uid = ctypes.create_string_buffer(128)
        assert self.lib.cfx_comm_unique_id(self.ctx[0], uid) == 0
        self.comm = [None] * W
        self.streams = [torch.cuda.Stream() for _ in range(W)]
        self.run(lambda r: self._mk(r, uid))

    def _mk(self, r, uid):
        self.comm[r] = self.lib.cfx_comm_create(self.ctx[r], uid, self.W, r)
        assert self.comm[r], self.lib.cfx_last_error_string(self.ctx[r])

    def run(self, fn):
        errs = [None] * self.W

        def body(r):
            try:
                torch.cuda.set_device(0)
                fn(r)
            except BaseException as e:  # noqa: BLE001
                errs[r] = e
        ts = [threading.Thread(target=body, args=(r,)) for r in range(self.W)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ts), "a rank is stuck in a collective"
        for e in errs:
            if e is not None:
                raise e

    def close(self):
        for c in self.comm:
            if c:
                self.lib.cfx_comm_destroy(c)
        for c in self.ctx:
            self.lib.cfx_destroy(c)


def _workload(W, L, N, C, seed):
    rng = np.random.default_rng(seed)
    base0 = rng.standard_normal((W, L, 2, N, C)).astype(np.float16)                       # what WARMUP left for every shard
    xs = [(base0.astype(np.float32) + 0.1 * (s + 1) * rng.standard_normal(base0.shape).astype(np.float32)).astype(np.float16)
          for s in range(2)]
    return base0, xs


def _expected(base0, xs, steps):
    W, L = base0.shape[:2]
    want = base0.copy()
    for s in range(steps):
        for r in range(W):
            for l in range(L):
                for b in range(2):
                    _, nb = R.residual_compress("binary", xs[s & 1][r, l, b], want[r, l, b], 0)
                    want[r, l, b] = nb
    return want


@pytest.mark.parametrize("W,pattern,mode,runner,G", [
    (2, "allgather", 0, "inorder", 1), (2, "allgather", 1, "inorder", 1), (2, "allgather", 2, "inorder", 1),
    (2, "allgather", 0, "pipelined", 2), (2, "allgather", 1, "pipelined", 2), (2, "allgather", 2, "pipelined", 3),
    (3, "allgather", 0, "inorder", 1), (3, "relay", 0, "inorder", 1), (3, "relay", 1, "inorder", 1), (2, "relay", 2, "inorder", 1),
])
def test_native_exchange_multi_rank_vs_oracle(W, pattern, mode, runner, G):
    from compactfusion_amd import _lib, codecs as K
    L, N, C, STEPS = 5, 96, 1024, 3
    lib = _lib.load()
    ranks = Ranks(W)
    base0, xs = _workload(W, L, N, C, 100 + W)
    slot = (K.packet_bytes(1, N, C) + 255) // 256 * 256
    wsb = lib.cfx_workspace_bytes(1, N, C, 0, 2)
    state = {}

    def rank_main(r):
        ctx, comm, stream = ranks.ctx[r], ranks.comm[r], ranks.streams[r]
        with torch.cuda.stream(stream):
            x = [dev16(xs[s][r]) for s in range(2)]                                      # [L,2,N,C]
            own = dev16(base0[r])
            peer = {q: dev16(base0[q]) for q in range(W) if q != r}
            send = torch.zeros(L, 2, slot, dtype=torch.uint8, device="cuda")
            recv = torch.zeros(L, W, 2, slot, dtype=torch.uint8, device="cuda")       # all-gather layout per layer [rank][K|V]
            hop = torch.zeros(L, W, 2, slot, dtype=torch.uint8, device="cuda")        # relay: hop[l][h] = packet after h hops
            ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            plans = []
            for s in range(2):
                plan = lib.cfx_plan_create(ctx)
                assert lib.cfx_plan_set_exchange_stream(plan, mode) == 0

                def comp(l):
                    c = (_lib.CompItem * 2)(*[_lib.CompItem(x[s][l, b].data_ptr(), own[l, b].data_ptr(), None, send[l, b].data_ptr()) for b in range(2)])
                    assert lib.cfx_plan_add_compress(plan, 1, N, C, 0, 0, 2, c, ws.data_ptr(), wsb) >= 0

                def dec(l, items):
                    d = (_lib.DecompItem * len(items))(*items)
                    assert lib.cfx_plan_add_decompress(plan, 1, N, C, 0, len(items), d) >= 0

                def own_items(l):
                    return [_lib.DecompItem(send[l, b].data_ptr(), own[l, b].data_ptr(), own[l, b].data_ptr()) for b in range(2)]
                if pattern == "allgather":
                    for a in range(0, L, G):
                        e = min(L, a + G)
                        for l in range(a, e):
                            comp(l)
                        # one collective per group of layers: send[a:e] is contiguous, the receive region is [rank][layer][K|V]
                        g0 = lib.cfx_plan_add_all_gather(plan, comm, send[a].data_ptr(), recv[a].data_ptr(), (e - a) * 2 * slot)
                        assert g0 >= 0
                        if runner == "inorder":
                            assert lib.cfx_plan_add_wait(plan, g0) >= 0
                        region = recv[a:e].view(-1)
                        for l in range(a, e):
                            def pkt(q, b):
                                off = ((q * (e - a) + (l - a)) * 2 + b) * slot
                                return region.data_ptr() + off
                            dec(l, own_items(l) + [_lib.DecompItem(pkt(q, b), peer[q][l, b].data_ptr(), peer[q][l, b].data_ptr())
                                                   for q in range(W) if q != r for b in range(2)])
                else:
                    for l in range(L):
                        comp(l)
                        for h in range(W - 1):
                            src = send[l] if h == 0 else hop[l, h]
                            g0 = lib.cfx_plan_add_ring_hop(plan, comm, src.data_ptr(), hop[l, h + 1].data_ptr(), 2 * slot)
                            assert g0 >= 0
                            assert lib.cfx_plan_add_wait(plan, g0) >= 0
                            q = (r - h - 1) % W
                            items = [_lib.DecompItem(hop[l, h + 1, b].data_ptr(), peer[q][l, b].data_ptr(), peer[q][l, b].data_ptr()) for b in range(2)]
                            dec(l, (own_items(l) if h == 0 else []) + items)
                assert lib.cfx_plan_finalize(plan) == 0
                plans.append(plan)
            run = lib.cfx_plan_run if runner == "inorder" else lib.cfx_plan_run_pipelined
            for s in range(STEPS):
                assert run(plans[s & 1], 0, lib.cfx_plan_size(plans[s & 1]), stream.cuda_stream) == 0, lib.cfx_last_error_string(ctx)
            stream.synchronize()
            for p in plans:
                lib.cfx_plan_destroy(p)
            state[r] = (own, peer)
    try:
        ranks.run(rank_main)
        torch.cuda.synchronize()
        want = _expected(base0, xs, STEPS)
        assert not np.array_equal(want.view(np.uint16), base0.view(np.uint16))
        for r in range(W):
            own, peer = state[r]
            assert np.array_equal(bits(own), want[r].view(np.uint16)), f"rank {r}: own error-feedback state differs from the oracle"
            for q, t in peer.items():
                assert np.array_equal(bits(t), want[q].view(np.uint16)), f"rank {r}: reconstruction of rank {q}'s shard differs from the oracle"
    finally:
        ranks.close()
