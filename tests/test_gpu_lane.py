"""The flag-synchronised exchange lane (cfx_plan_run_lane / cfx_attn_merge_wait / CU-masked streams; include/cfx.h) on the GPU:
`compact_fwd` (gather schedule) with 8 logical ranks looped back through tests/fake_rccl, on the default stream AND on the lane's
CU-masked compute stream, error feedback on and off - every state a rank holds against the oracle's replay bit for bit, the
block-wise attention output against ONE attention over the K,V the rank holds (reference flow: xfuser/compact/ring.py:120-275);
plus the failure path of a wait (bounded spin -> CFX_ERR_GATE at the next call) and the flag primitives themselves."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ref_np as R

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
W = 8


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


@pytest.fixture
def loopback(monkeypatch):
    """An 8-rank ring whose peers are all this rank (fake RCCL in loop-back mode; torch.distributed is not initialised)."""
    from compactfusion_amd import _lib, codecs as K, exchange
    from compactfusion_amd.compact import ring, main as cm
    from compactfusion_amd.collector import collector
    from compactfusion_amd.prof import Profiler
    lib = _lib.load()
    monkeypatch.setenv("CFX_FAKE_RCCL_MODE", "loopback")
    monkeypatch.setenv("CFX_RING_EXCHANGE", "native")
    monkeypatch.setenv("CFX_RING_SCHEDULE", "gather")
    monkeypatch.setattr(ring.dist, "get_rank", lambda g=None: 0)
    monkeypatch.setattr(ring.dist, "get_world_size", lambda g=None: W)
    monkeypatch.setattr(ring.dist, "all_gather_into_tensor",            # WARMUP steps gather raw fp16 through torch.distributed
                        lambda recv, send, group=None: recv.view(W, -1).copy_(send.view(1, -1).expand(W, -1)))
    fake = _fake_path()

    class LoopComm:
        def __init__(self, group, device):
            ctx = K.context(device)
            assert lib.cfx_rccl_load(fake.encode()) == 0
            uid = ctypes.create_string_buffer(128)
            assert lib.cfx_comm_unique_id(ctx, uid) == 0
            self.handle = lib.cfx_comm_create(ctx, uid, W, 0)
            assert self.handle

    exchange.set_comm_factory(LoopComm)
    Profiler.instance().disable()
    collector.init(collector.Collector("/tmp/none", enabled=False))
    ring._xbuf.clear()
    ring._steady.clear()
    yield ring, cm
    exchange.set_comm_factory(None)
    ring._xbuf.clear()
    ring._steady.clear()


def _drift(seed, shape, T):
    g = torch.Generator().manual_seed(seed)
    cur = torch.randn(*shape, generator=g).half()
    out = []
    for _ in range(T):
        out.append(cur.contiguous())
        cur = (cur.float() + 0.1 * torch.randn(*shape, generator=g)).half()
    return out


def _late(t):
    """A copy of `t` that only exists after ~1 ms of queued work on the current stream: what a model's projection kernels in front of
    compact_fwd look like to the exchange stream (resident inputs let a missing cross-stream ordering pass by luck)."""
    try:
        torch.cuda._sleep(3_000_000)
    except Exception:  # noqa: BLE001
        junk = torch.ones(2048, 2048, device=t.device, dtype=torch.float16)
        for _ in range(8):
            junk = (junk @ junk) * 1e-4
    return t.clone()


@pytest.mark.parametrize("masked,ef,xmode,late", [(False, True, "lane", False), (True, True, "lane", False), (True, False, "lane", False),
                                                  (False, False, "lane", False), (False, False, "chain", False), (False, True, "chain", False),
                                                  (False, True, "lane", True), (True, True, "lane", True),
                                                  ("auto", True, "auto", True), ("auto", False, "auto", False), ("auto-side", True, "auto", True),
                                                  ("sticky", True, "auto", True)])
def test_lane_ring_forward_vs_oracle(loopback, monkeypatch, masked, ef, xmode, late):
    """masked False / True: the caller on an ordinary stream with the lane switched off (flags on unmasked streams) / on the lane's compute
    stream.  "auto" / "auto-side" / "sticky": the DEFAULT settings, caller on the default stream / a side stream - compact_fwd puts itself on
    the lane (forked from and joined to the caller's stream by flag kernels, or - sticky - the compute stream becomes the current one)."""
    ring, cm = loopback
    from compactfusion_amd import lanes
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    from compactfusion_amd.compact.attention import block_attention
    auto = isinstance(masked, str)
    if auto:
        monkeypatch.delenv("CFX_RING_EXCHANGE_STREAM", raising=False)
        monkeypatch.setenv("CFX_LANE", "sticky" if masked == "sticky" else "auto")
    else:
        monkeypatch.setenv("CFX_RING_EXCHANGE_STREAM", xmode)
        monkeypatch.setenv("CFX_LANE", "off")
    L, STEPS = 3, 5
    shape = (1, 64, 8, 64)
    N, C = 64, 512
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T.BINARY, comp_rank=-1,
                                  residual=1, ef=ef, fastpath=ef))       # the reference's rule: fastpath needs error feedback
    qs = [_drift(7 + l, shape, STEPS) for l in range(L)]
    ks = [_drift(17 + l, shape, STEPS) for l in range(L)]
    vs = [_drift(27 + l, shape, STEPS) for l in range(L)]
    dev = torch.device("cuda:0")
    if auto:
        stream = torch.cuda.Stream(dev) if masked == "auto-side" else torch.cuda.default_stream(dev)
    else:
        stream = lanes.compute_stream(0) if masked else torch.cuda.current_stream(dev)
    # oracle replay per (layer, K|V): own[s] = the sender's state after step s, peer[s] = what every receiver holds after step s
    own_w, peer_w = {}, {}
    for l in range(L):
        for name, seq in (("k", ks[l]), ("v", vs[l])):
            o = seq[0].numpy().reshape(N, C).copy()
            pr = o.copy()
            own_w[(l, name)], peer_w[(l, name)] = [o.copy()], [pr.copy()]
            for x in seq[1:]:
                x2 = x.numpy().reshape(N, C)
                pkt, nb = R.residual_compress("binary", x2, o, 0, ef=ef)
                pr = R.residual_decompress("binary", pkt, pr, N, C, 0)
                o = nb
                own_w[(l, name)].append(o.copy())
                peer_w[(l, name)].append(pr.copy())
    outs = {}
    with torch.cuda.stream(stream):
        dq = [[t.to(dev) for t in qs[l]] for l in range(L)]
        dk = [[t.to(dev) for t in ks[l]] for l in range(L)]
        dv = [[t.to(dev) for t in vs[l]] for l in range(L)]
        for s in range(STEPS):
            cm.compact_set_step(s)
            for l in range(L):
                kin, vin = (_late(dk[l][s]), _late(dv[l][s])) if late else (dk[l][s], dv[l][s])     # late: K,V produced right in front of the call
                out, lse, _ = ring.compact_fwd(dq[l][s], kin, vin, causal=False, mod_idx=l, current_iter=s)
                if auto and masked != "sticky":
                    assert torch.cuda.current_stream(dev).cuda_stream == stream.cuda_stream, "the caller's stream is the current stream again"
                elif auto:
                    assert torch.cuda.current_stream(dev).cuda_stream == lanes.compute_stream(0).cuda_stream, "sticky: the compute stream stays current"
                out = out * 1.0                        # consumed right away on whatever stream the caller now has: the join must order it
                outs[(s, l)] = (out, lse)
            torch.cuda.synchronize()
            cache = cm.compact_cache()
            for l in range(L):
                for name in ("k", "v"):
                    own = bits(cache.get_base(f"{l}-0-{name}")).reshape(N, C)
                    assert np.array_equal(own, R.bits(own_w[(l, name)][s])), (s, l, name, "own state")
                    for r in range(1, W):
                        peer = bits(cache.get_base(f"{l}-{r}-{name}")).reshape(N, C)
                        assert np.array_equal(peer, R.bits(peer_w[(l, name)][s])), (s, l, r, name, "peer reconstruction")
                # block-wise attention + merges == ONE attention over what the rank holds (own exact K,V first, then the peers in ring order)
                kk = [dk[l][s]] + [cache.get_base(f"{l}-{(0 - t) % W}-k").view(shape) for t in range(1, W)]
                vv = [dv[l][s]] + [cache.get_base(f"{l}-{(0 - t) % W}-v").view(shape) for t in range(1, W)]
                ref_o, ref_l = block_attention(dq[l][s], torch.cat(kk, dim=1), torch.cat(vv, dim=1), 0.0, None, causal=False)
                o, lse = outs[(s, l)]
                torch.testing.assert_close(o.float(), ref_o.float(), rtol=2e-3, atol=2e-3)
                torch.testing.assert_close(lse.float(), ref_l.float(), rtol=1e-3, atol=1e-3)
    exs = [e for e in ring._xbuf.values() if e.sig is not None]
    assert exs and all(e.plan is not None for e in exs), "the native per-layer plan was not used"
    assert all(e.lane == (xmode in ("lane", "auto")) for e in exs)
    assert len(ring._steady) == L, "the steady-state lane never engaged"
    from compactfusion_amd import _lib, codecs as K
    if xmode in ("lane", "auto"):
        assert _lib.load().cfx_plan_epoch(exs[0].plan) == STEPS - 1            # exactly one epoch per compressed step since the plan was bound (the general path advances it too)
    assert _lib.load().cfx_gate_errors(K.context(0)) == 0


def test_masked_streams_partition_the_cus():
    """cfx_stream_create_masked: the exchange and compute streams of a lane report disjoint CU masks covering the device."""
    from compactfusion_amd import lanes
    ln = lanes.lane(0)
    total = torch.cuda.get_device_properties(0).multi_processor_count
    assert ln.exchange_cus + ln.compute_cus == total
    hip = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                hip = ctypes.CDLL(line.split()[-1])
                break
    assert hip is not None
    masks = []
    for s in (ln.exchange, ln.compute):
        m = (ctypes.c_uint32 * 8)()
        assert hip.hipExtStreamGetCUMask(ctypes.c_void_p(s.cuda_stream), 8, m) == 0
        masks.append(int.from_bytes(bytes(m), "little"))
    assert masks[0] & masks[1] == 0
    assert bin(masks[0]).count("1") == ln.exchange_cus and bin(masks[1]).count("1") == ln.compute_cus


def test_flag_set_wait_order_two_streams():
    """Producer / consumer over two streams ordered ONLY by flags (data flag + acknowledge flag): the consumer's copy of the
    payload always sees the producer's fill of the same epoch - never an older or a newer one."""
    from compactfusion_amd import _lib, codecs as K
    from compactfusion_amd import lanes
    lib, ctx = _lib.load(), K.context(0)
    # flag-ordered streams must not share a hardware queue (pool streams may): each of these owns its queue
    a, b = lanes.exchange_stream(0), lanes.compute_stream(0)
    flags = torch.zeros(32, dtype=torch.int32, device="cuda")
    ready, ack = flags.data_ptr(), flags.data_ptr() + 64
    src = torch.zeros(1 << 20, dtype=torch.int32, device="cuda")
    seen = torch.zeros(64, 2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    E = 40
    for e in range(1, E + 1):
        with torch.cuda.stream(b):          # consumer: waits for epoch e, reads, acknowledges
            assert lib.cfx_flag_wait(ctx, ready, e, b.cuda_stream) == 0
            # the consumer must be a KERNEL (what a flag orders is kernels: the attention blocks).  A device-to-device memcpy - what
            # `seen[e, 0] = src[0]` is - was observed reading the previous epoch here (it does not take the kernel path's acquire)
            torch.add(src[0:1], 0, out=seen[e, 0:1])
            torch.add(src[-1:], 0, out=seen[e, 1:2])
            assert lib.cfx_flag_set(ctx, ack, e, b.cuda_stream) == 0
        with torch.cuda.stream(a):          # producer: waits until epoch e-1 was consumed, overwrites, publishes
            assert lib.cfx_flag_wait(ctx, ack, e - 1, a.cuda_stream) == 0
            src.fill_(e)
            assert lib.cfx_flag_set(ctx, ready, e, a.cuda_stream) == 0
    torch.cuda.synchronize()
    want = torch.arange(1, E + 1, dtype=torch.int32)
    assert torch.equal(seen[1:E + 1, 0].cpu(), want) and torch.equal(seen[1:E + 1, 1].cpu(), want)
    assert lib.cfx_gate_errors(ctx) == 0


def test_wait_timeout_is_reported_by_the_next_call():
    """A wait that never ends gives up after the gate timeout and the NEXT native call on the context fails with CFX_ERR_GATE
    (no device synchronisation needed to learn it); cfx_gate_errors reads and clears the count."""
    from compactfusion_amd import _lib
    lib = _lib.load()
    ctx = lib.cfx_create(0)
    try:
        assert lib.cfx_prepare(ctx) == 0
        assert lib.cfx_set_gate_timeout_ms(ctx, 20) == 0
        flag = torch.zeros(16, dtype=torch.int32, device="cuda")
        s = torch.cuda.Stream()
        assert lib.cfx_flag_wait(ctx, flag.data_ptr(), 5, s.cuda_stream) == 0       # nobody will ever write 5
        s.synchronize()
        out = torch.zeros(1, 4, 2, 64, dtype=torch.float32, device="cuda")
        lse = torch.zeros(1, 4, 2, 1, dtype=torch.float32, device="cuda")
        bo = torch.zeros(1, 4, 2, 64, dtype=torch.float16, device="cuda")
        bl = torch.zeros(1, 2, 4, dtype=torch.float32, device="cuda")
        rc = lib.cfx_attn_merge(ctx, out.data_ptr(), lse.data_ptr(), bo.data_ptr(), bl.data_ptr(), 1, 4, 2, 64, 1, 1, s.cuda_stream)
        assert rc == -8 and b"timed out" in lib.cfx_last_error_string(ctx)
        plan = lib.cfx_plan_create(ctx)
        assert lib.cfx_plan_run(plan, 0, 0, s.cuda_stream) == -8
        assert lib.cfx_gate_errors(ctx) == 1
        assert lib.cfx_gate_errors(ctx) == 0
        assert lib.cfx_plan_run(plan, 0, 0, s.cuda_stream) == 0
        assert lib.cfx_attn_merge(ctx, out.data_ptr(), lse.data_ptr(), bo.data_ptr(), bl.data_ptr(), 1, 4, 2, 64, 1, 1, s.cuda_stream) == 0
        # the merge launch's own wait
        assert lib.cfx_attn_merge_wait(ctx, out.data_ptr(), lse.data_ptr(), bo.data_ptr(), bl.data_ptr(), 1, 4, 2, 64, 1, 0,
                                       flag.data_ptr(), 9, s.cuda_stream) == 0
        s.synchronize()
        assert lib.cfx_gate_errors(ctx) == 1
        lib.cfx_plan_destroy(plan)
    finally:
        lib.cfx_destroy(ctx)


def test_merge_wait_releases_when_the_flag_arrives():
    from compactfusion_amd import _lib, codecs as K
    from compactfusion_amd import lanes
    lib, ctx = _lib.load(), K.context(0)
    a, b = lanes.compute_stream(0), lanes.dedicated_stream(0)          # each owns its hardware queue
    flag = torch.zeros(16, dtype=torch.int32, device="cuda")
    B, S, H, D = 1, 33, 3, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    bo = torch.randn(B, S, H, D, device="cuda", dtype=torch.float16, generator=g)
    bl = torch.randn(B, H, S, device="cuda", dtype=torch.float32, generator=g)
    out = torch.empty(B, S, H, D, dtype=torch.float32, device="cuda")
    lse = torch.empty(B, S, H, 1, dtype=torch.float32, device="cuda")
    payload = torch.zeros(1 << 22, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        assert lib.cfx_attn_merge_wait(ctx, out.data_ptr(), lse.data_ptr(), bo.data_ptr(), bl.data_ptr(), B, S, H, D, 1, 1,
                                       flag.data_ptr(), 3, a.cuda_stream) == 0
        got = payload.sum()                      # runs only after the merge launch has seen epoch 3
    with torch.cuda.stream(b):
        payload.fill_(1)                         # the data behind the flag
        assert lib.cfx_flag_set(ctx, flag.data_ptr(), 3, b.cuda_stream) == 0
    torch.cuda.synchronize()
    assert int(got) == payload.numel()
    torch.testing.assert_close(out, bo.float())
    assert lib.cfx_gate_errors(ctx) == 0
