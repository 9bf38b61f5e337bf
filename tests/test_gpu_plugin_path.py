"""The PRODUCT path of the gather schedules on the GPU (-m gpu): `compact_fwd` (ring gather schedule, reference xfuser/compact/ring.py:188-206
+ 265-269) and `compact_all_gather_kv` (what `patch_gather_fwd` calls; reference main.py:390-420, patchpara/fwd.py:88-102) issue ONE native
op per layer - `cfx_plan_add_exchange_layer[_p2p]` through compact/xlayer.py - and for the 1-bit codec that op is ONE codec launch (kernel
id 31, the gated layer launch) with no separate reconstruction launch.  Asserted through `cfx_profile_read` kernel ids, with every state
against the oracle's replay bit for bit:
  * 8 logical ranks looped back in one process (packets in the uncached IPC arena, every logical peer reads this rank's packets), on a
    side stream and on the legacy default stream;
  * two rank PROCESSES on one GPU (packets read in place through IPC mappings, first two executions validated across the ranks), two
    generations with compact_reset in between, a poisoned reconstruction that must send every rank to the next transport, and 20 resets
    that must leave the device memory flat."""
import ctypes
import os

import numpy as np
import pytest
import torch

import _dist_workers as W
from oracle import ref_np as R
from test_gpu_schedules import ONAME, _chain, _spawn

pytestmark = pytest.mark.gpu
WL = 8


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


@pytest.fixture
def loop8(monkeypatch):
    """An 8-rank group whose peers are all this rank; torch.distributed is not initialised."""
    from compactfusion_amd.compact import ring, main as cm, xlayer
    from compactfusion_amd.collector import collector
    from compactfusion_amd.prof import Profiler
    monkeypatch.setenv("CFX_RING_SCHEDULE", "gather")
    monkeypatch.setenv("CFX_LANE", "off")          # these tests are about the one-op layer exchange on the caller's stream (the default is the lane)
    monkeypatch.delenv("CFX_RING_EXCHANGE_STREAM", raising=False)
    monkeypatch.delenv("CFX_RING_EXCHANGE", raising=False)
    monkeypatch.setattr(ring.dist, "get_rank", lambda g=None: 0)
    monkeypatch.setattr(ring.dist, "get_world_size", lambda g=None: WL)
    monkeypatch.setattr(ring.dist, "all_gather_into_tensor",            # WARMUP steps gather raw fp16 through torch.distributed
                        lambda recv, send, group=None: recv.view(WL, -1).copy_(send.view(1, -1).expand(WL, -1)))
    xlayer.set_p2p_loopback(True)
    Profiler.instance().disable()
    collector.init(collector.Collector("/tmp/none", enabled=False))
    ring._xbuf.clear()
    ring._steady.clear()
    yield ring, cm, xlayer
    cm._drop_kv_exchanges()
    for e in ring._xbuf.values():
        e.close()
    ring._xbuf.clear()
    ring._steady.clear()
    xlayer.set_p2p_loopback(False)


def _replay(codec, seqs, N, C, ef=True):
    name, param = ONAME[codec]
    st = seqs[0].numpy().reshape(N, C).copy()
    out = [R.bits(st).copy()]
    for x in seqs[1:]:
        _, st = R.residual_compress(name, x.numpy().reshape(N, C), st, param, ef=ef)
        out.append(R.bits(st).copy())
    return out


def _kernel_ids(lib, ctx):
    ids = (ctypes.c_int * 8192)()
    ms = (ctypes.c_float * 8192)()
    n = lib.cfx_profile_read(ctx, ids, ms, 8192)
    return [ids[i] for i in range(n)]


@pytest.mark.parametrize("api,stream_kind", [("ring", "side"), ("ring", "default"), ("gather", "side"), ("gather", "default")])
def test_plugin_call_issues_the_gated_layer_launch(loop8, api, stream_kind):
    ring, cm, xlayer = loop8
    from compactfusion_amd import _lib, codecs as K
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig, PatchConfig
    lib, ctx = _lib.load(), K.context(0)
    assert lib.cfx_hw_queues_ok() == 1, "tests/conftest.py sets GPU_MAX_HW_QUEUES before HIP starts"
    L, STEPS = 3, 5
    shape, N, C = (1, 64, 8, 64), 64, 512
    kw = dict(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T.BINARY, comp_rank=-1, residual=1, ef=True, fastpath=True)
    if api == "gather":
        kw.update(override_with_patch_gather_fwd=True, patch_gather_fwd_config=PatchConfig(True, False, 1))
    cm.compact_init(CompactConfig(**kw))
    qs = [W.drift(7 + l, shape, STEPS) for l in range(L)]
    ks = [W.drift(17 + l, shape, STEPS) for l in range(L)]
    vs = [W.drift(27 + l, shape, STEPS) for l in range(L)]
    want = {(l, n): _replay("BINARY", seq[l], N, C) for l in range(L) for n, seq in (("k", ks), ("v", vs))}
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(dev) if stream_kind == "side" else torch.cuda.default_stream(dev)
    MASK = (1 << 31) | (1 << 4) | (1 << 16) | (1 << 27)
    with torch.cuda.stream(stream):
        dq, dk, dv = ([[t.to(dev) for t in seq[l]] for l in range(L)] for seq in (qs, ks, vs))
        for s in range(STEPS):
            cm.compact_set_step(s)
            torch.cuda.synchronize()
            assert lib.cfx_profile_enable(ctx, 8192, MASK, 1) == 0
            for l in range(L):
                ring.compact_fwd(dq[l][s], dk[l][s], dv[l][s], causal=False, mod_idx=l, current_iter=s)
            torch.cuda.synchronize()
            got = _kernel_ids(lib, ctx)
            lib.cfx_profile_enable(ctx, 0, 0, 1)
            if s > 0:
                # ONE codec launch per layer: the gated layer launch (31); no reconstruction (4), error-feedback (16) or plain compress (27) launch
                assert got.count(31) == L and got.count(4) == 0 and got.count(16) == 0 and got.count(27) == 0, (api, stream_kind, s, got)
            cache = cm.compact_cache()
            for l in range(L):
                for n in ("k", "v"):
                    for r in range(WL):
                        key = f"{l}-{r}-{n}" if api == "ring" else f"{l}-{n}-{r}"
                        assert np.array_equal(bits(cache.get_base(key)).reshape(N, C), want[(l, n)][s].reshape(N, C)), (api, s, l, n, r)
    ops = [e.xop for e in ring._xbuf.values() if e.xop is not None] + [e.xop for e in cm._kv_exchanges.values() if e.xop is not None]
    assert len(ops) == L and all(o.transport == "p2p" for o in ops), "the layer op / the IPC arena was not used"
    arena = next(iter(xlayer._arenas.values()))
    assert arena.kind == 2, "the packet arena is uncached device memory (hipExtMallocWithFlags(hipDeviceMallocUncached))"
    assert lib.cfx_gate_errors(ctx) == 0


@pytest.mark.parametrize("codec", ["INT2", "INT4", "INT8", "SPARSE"])
@pytest.mark.parametrize("api", ["ring", "gather"])
def test_plugin_call_other_codecs_one_native_op(loop8, api, codec):
    """The other streaming codecs through the same one-op-per-layer path (they run compress ; exchange ; reconstruct in stream order inside it)."""
    ring, cm, xlayer = loop8
    from compactfusion_amd import _lib, codecs as K
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig, PatchConfig
    lib, ctx = _lib.load(), K.context(0)
    L, STEPS = 2, 4
    shape, N, C = (1, 64, 16, 64), 64, 1024
    fast = codec == "INT2"
    kw = dict(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T[codec], comp_rank=-1, residual=1, ef=True, fastpath=fast, sparse_ratio=8)
    if api == "gather":
        kw.update(override_with_patch_gather_fwd=True, patch_gather_fwd_config=PatchConfig(True, False, 1))
    cm.compact_init(CompactConfig(**kw))
    qs = [W.drift(7 + l, shape, STEPS) for l in range(L)]
    ks = [W.drift(17 + l, shape, STEPS) for l in range(L)]
    vs = [W.drift(27 + l, shape, STEPS) for l in range(L)]
    want = {(l, n): _replay(codec, seq[l], N, C) for l in range(L) for n, seq in (("k", ks), ("v", vs))}
    dev = torch.device("cuda:0")
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        for s in range(STEPS):
            cm.compact_set_step(s)
            for l in range(L):
                ring.compact_fwd(qs[l][s].to(dev), ks[l][s].to(dev), vs[l][s].to(dev), causal=False, mod_idx=l, current_iter=s)
            torch.cuda.synchronize()
            cache = cm.compact_cache()
            for l in range(L):
                for n in ("k", "v"):
                    for r in range(WL):
                        key = f"{l}-{r}-{n}" if api == "ring" else f"{l}-{n}-{r}"
                        assert np.array_equal(bits(cache.get_base(key)).reshape(N, C), want[(l, n)][s].reshape(N, C)), (api, codec, s, l, n, r)
    ops = [e.xop for e in ring._xbuf.values() if e.xop is not None] + [e.xop for e in cm._kv_exchanges.values() if e.xop is not None]
    assert len(ops) == L and all(o.transport == "p2p" for o in ops)
    assert lib.cfx_gate_errors(ctx) == 0


@pytest.mark.parametrize("codec,rank", [("LOW_RANK", 8), ("LOW_RANK", 12), ("LOW_RANK_Q", 32)])
@pytest.mark.parametrize("api", ["ring", "gather"])
def test_plugin_call_low_rank_family_one_native_call_per_layer(loop8, api, codec, rank):
    """Round 6: the LOW_RANK / LOW_RANK_Q presets through compact_fwd / compact_all_gather_kv take the layer op too - a chain of native
    plan ops (factor chain of K,V ; publish-and-wait ; ONE batched reconstruction of the 14 peer tensors) replayed by one host call.
    Before, the ring forward issued one compress per tensor and one reconstruction per peer tensor from Python (16+ launches and 0.44 ms
    of host time per FLUX layer: 25 ms per step).  Checked: the launches of a step (one factor chain and one reconstruction launch per
    layer; LOW_RANK_Q: plus its int4 quantise / dequantise), every peer's state == the sender's bit for bit, and the states == the
    codec called directly, tensor by tensor, with the same pinned start matrix (the chain's sums do not depend on the batch)."""
    ring, cm, xlayer = loop8
    from compactfusion_amd import _lib, codecs as K
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig, PatchConfig, lowrank
    lib, ctx = _lib.load(), K.context(0)
    L, STEPS = 2, 4
    shape, N, C = (1, 128, 16, 64), 128, 1024
    quant = codec == "LOW_RANK_Q"
    kw = dict(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T[codec], comp_rank=rank, residual=1, ef=True, fastpath=False)
    if api == "gather":
        kw.update(override_with_patch_gather_fwd=True, patch_gather_fwd_config=PatchConfig(True, False, 1))
    cm.compact_init(CompactConfig(**kw))
    qs = [W.drift(7 + l, shape, STEPS) for l in range(L)]
    ks = [W.drift(17 + l, shape, STEPS) for l in range(L)]
    vs = [W.drift(27 + l, shape, STEPS) for l in range(L)]
    dev = torch.device("cuda:0")
    pinned = torch.randn(C, rank, generator=torch.Generator().manual_seed(5))
    q0 = torch.zeros(C, K.lr_rank_pad(rank), device=dev)
    q0[:, :rank] = pinned.to(dev)
    want = {}
    for l in range(L):
        for n, seq in (("k", ks), ("v", vs)):
            st = seq[l][0].to(dev).reshape(N, C).clone()
            outs = [bits(st).copy()]
            for x in seq[l][1:]:
                pk = torch.empty(K.lr_packet_halves(quant, N, C, rank), dtype=torch.float16, device=dev)
                K.lr_compress_batch(quant, [x.to(dev).reshape(N, C)], [st], [st], [pk], [q0], N, C, rank, update_cache=True, ef=True)
                torch.cuda.synchronize()
                outs.append(bits(st).copy())
            want[(l, n)] = outs
    lowrank.set_init_q(pinned)
    try:
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            for s in range(STEPS):
                cm.compact_set_step(s)
                torch.cuda.synchronize()
                assert lib.cfx_profile_enable(ctx, 8192, 0xffffffff, 1) == 0
                for l in range(L):
                    ring.compact_fwd(qs[l][s].to(dev), ks[l][s].to(dev), vs[l][s].to(dev), causal=False, mod_idx=l, current_iter=s)
                torch.cuda.synchronize()
                got = _kernel_ids(lib, ctx)
                lib.cfx_profile_enable(ctx, 0, 0, 1)
                if s > 0:
                    n_dec = got.count(22)
                    assert got.count(17) == L and L <= n_dec <= 2 * L, (api, codec, s, got)      # (k_lr_decode also closes the sender's own update where the chain does not fuse it)
                cache = cm.compact_cache()
                for l in range(L):
                    for n in ("k", "v"):
                        own = bits(cache.get_base(f"{l}-0-{n}" if api == "ring" else f"{l}-{n}-0")).reshape(N, C)
                        assert np.array_equal(own, want[(l, n)][s].reshape(N, C)), (api, codec, s, l, n, "vs the codec called directly")
                        for r in range(1, WL):
                            key = f"{l}-{r}-{n}" if api == "ring" else f"{l}-{n}-{r}"
                            assert np.array_equal(bits(cache.get_base(key)).reshape(N, C), own), (api, codec, s, l, n, r)
    finally:
        lowrank.set_init_q(None)
    ops = [e.xop for e in ring._xbuf.values() if e.xop is not None] + [e.xop for e in cm._kv_exchanges.values() if e.xop is not None]
    assert len(ops) == L and all(o.transport == "p2p" and o.lowrank and o.nops == 3 for o in ops)
    assert lib.cfx_gate_errors(ctx) == 0


@pytest.mark.parametrize("codec,rank", [("LOW_RANK", 8), ("LOW_RANK_Q", 32)])
@pytest.mark.parametrize("lane", ["auto", "sticky"])
def test_low_rank_layer_beside_the_attention_blocks(loop8, monkeypatch, codec, rank, lane):
    """Protocol 2 for the low-rank family (round 6): with the lane at its default, a steady LOW_RANK / LOW_RANK_Q layer keeps its factor
    chain (+ publish-and-wait) on the compute lane and leaves the peers' reconstructions to the exchange lane, peer by peer behind flags
    the merge launches wait for (xlayer.LayerOp.run(lane=True) / lane_chain).  Against the same steps with the lane off (the one-call
    layer op on the caller's stream): every state bit for bit (same pinned start matrix: the chain's sums do not depend on the stream),
    the attention output to merge-order tolerance; and the launches say the lane form ran: one reconstruction launch per peer."""
    ring, cm, xlayer = loop8
    from compactfusion_amd import _lib, codecs as K, lanes
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig, lowrank
    lib, ctx = _lib.load(), K.context(0)
    L, STEPS = 3, 5
    shape, N, C = (1, 128, 16, 64), 128, 1024
    qs = [W.drift(7 + l, shape, STEPS) for l in range(L)]
    ks = [W.drift(17 + l, shape, STEPS) for l in range(L)]
    vs = [W.drift(27 + l, shape, STEPS) for l in range(L)]
    dev = torch.device("cuda:0")
    pinned = torch.randn(C, rank, generator=torch.Generator().manual_seed(5))

    def run(lane_mode):
        monkeypatch.setenv("CFX_LANE", lane_mode)
        cm._drop_kv_exchanges()
        for e in ring._xbuf.values():
            e.close()
        ring._xbuf.clear()
        ring._steady.clear()
        ring._lane_ok.clear()
        cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T[codec], comp_rank=rank, residual=1, ef=True, fastpath=False))
        states, outs, decodes = {}, {}, []
        lowrank.set_init_q(pinned)
        caller = torch.cuda.Stream(dev)
        try:
            with torch.cuda.stream(caller):
                for s in range(STEPS):
                    cm.compact_set_step(s)
                    torch.cuda.synchronize()
                    assert lib.cfx_profile_enable(ctx, 8192, 0xffffffff, 1) == 0
                    for l in range(L):
                        out, lse, _ = ring.compact_fwd(qs[l][s].to(dev), ks[l][s].to(dev), vs[l][s].to(dev), causal=False, mod_idx=l, current_iter=s)
                        want_stream = lanes.compute_stream(0) if (lane_mode == "sticky" and s >= 2) else None
                        if lane_mode == "auto":
                            assert torch.cuda.current_stream(dev).cuda_stream == caller.cuda_stream, "the caller's stream is the current stream again"
                        elif want_stream is not None:
                            assert torch.cuda.current_stream(dev).cuda_stream == want_stream.cuda_stream
                        outs[(s, l)] = (out * 1.0, lse * 1.0)
                    torch.cuda.synchronize()
                    decodes.append(_kernel_ids(lib, ctx).count(22))
                    lib.cfx_profile_enable(ctx, 0, 0, 1)
                    cache = cm.compact_cache()
                    for l in range(L):
                        for n in ("k", "v"):
                            for r in range(WL):
                                states[(s, l, n, r)] = bits(cache.get_base(f"{l}-{r}-{n}")).copy()
        finally:
            lowrank.set_init_q(None)
            torch.cuda.set_stream(torch.cuda.default_stream(dev))
        assert lib.cfx_gate_errors(ctx) == 0
        return states, outs, decodes
    st_off, out_off, dec_off = run("off")
    st_on, out_on, dec_on = run(lane)
    ops = [e.xop for e in ring._xbuf.values() if e.xop is not None]
    assert len(ops) == L and all(o.lowrank and o.transport == "p2p" and getattr(o, "_lane", None) is not None for o in ops), "the lane form never engaged"
    assert all(d >= (WL - 1) * L for d in dec_on[2:]) and all(d <= 2 * L for d in dec_off[1:]), (dec_on, dec_off)
    for key, want in st_off.items():
        assert np.array_equal(st_on[key], want), (key, "state differs between the lane form and the one-call form")
    for key, (o, lse) in out_off.items():
        torch.testing.assert_close(out_on[key][0].float(), o.float(), rtol=2e-3, atol=2e-3)
        torch.testing.assert_close(out_on[key][1].float(), lse.float(), rtol=1e-3, atol=1e-3)


def test_plugin_call_low_rank_draws_its_start_matrices_a_chunk_at_a_time(loop8):
    """The path a model takes (no pinned start matrix): every execution of a low-rank layer op starts from its own fresh Gaussian draw
    (reference compress_lowrank.py:41 draws per call), made one launch per chunk of layers and step (compact/xlayer.py StartPool) instead
    of one per layer.  Checked over 3 compressed steps of 5 layers: the draws counted, two layers' / two steps' matrices differ, every
    peer's state == the sender's bit for bit, and the residual the rank-8 packets leave shrinks with the steps (error feedback)."""
    ring, cm, xlayer = loop8
    from compactfusion_amd import _lib, codecs as K
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    lib, ctx = _lib.load(), K.context(0)
    L, STEPS, rank = 5, 4, 8
    shape, N, C = (1, 128, 16, 64), 128, 1024
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T.LOW_RANK, comp_rank=rank, residual=1, ef=True, fastpath=False))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    low = [torch.randn(N, 6, generator=g) @ torch.randn(6, C, generator=g) for _ in range(L)]      # rank-6 activations: a rank-8 packet can carry them
    q = torch.randn(shape, generator=g).half().to(dev)
    starts, errs = [], []
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        for s in range(STEPS):
            cm.compact_set_step(s)
            for l in range(L):
                x = ((1.0 + 0.05 * s) * low[l]).half().reshape(shape).to(dev)
                ring.compact_fwd(q, x, x.clone(), causal=False, mod_idx=l, current_iter=s)
            torch.cuda.synchronize()
            ops = {e.xop.key: e.xop for e in ring._xbuf.values() if e.xop is not None}
            if s > 0:
                starts.append({k_: o._q0.clone() for k_, o in ops.items()})
                cache = cm.compact_cache()
                e = 0.0
                for l in range(L):
                    own = cache.get_base(f"{l}-0-k").reshape(N, C)
                    for r in range(1, WL):
                        assert torch.equal(cache.get_base(f"{l}-{r}-k").reshape(N, C).view(torch.int16), own.view(torch.int16)), (s, l, r)
                    want = ((1.0 + 0.05 * s) * low[l]).half().to(dev)
                    e = max(e, float((own.float() - want.float()).norm() / want.float().norm()))
                errs.append(e)
    ops = [e.xop for e in ring._xbuf.values() if e.xop is not None]
    assert len(ops) == L and all(o.lowrank and o.transport == "p2p" for o in ops)
    pool = ops[0]._pool
    assert all(o._pool is pool for o in ops) and pool.draws == STEPS - 1, (pool.draws, "one draw per chunk (32 layers) and compressed step")
    mats = [m for st in starts for m in st.values()]
    assert all(bool(m.any()) for m in mats)
    assert len({m.cpu().numpy().tobytes() for m in mats}) == len(mats), "two executions started from the same matrix"
    assert errs[-1] < 2e-3 and errs[0] < 0.05, errs
    assert lib.cfx_gate_errors(ctx) == 0


def _check_states(res, codec, mode, world, gens, L=3, STEPS=4, shape=(1, 64, 8, 64)):
    for gen in range(gens):
        for l in range(L):
            want_k = [_chain(codec, W.drift(1000 * gen + 17 + 10 * l + q, shape, STEPS)) for q in range(world)]
            want_v = [_chain(codec, W.drift(1000 * gen + 27 + 10 * l + q, shape, STEPS)) for q in range(world)]
            for r in range(world):
                for s in range(STEPS):
                    for q in range(world):
                        assert np.array_equal(res[r][f"g{gen}/s{s}/l{l}/k{q}"].reshape(-1), want_k[q][s].reshape(-1)), (mode, gen, l, r, s, q, "k")
                        assert np.array_equal(res[r][f"g{gen}/s{s}/l{l}/v{q}"].reshape(-1), want_v[q][s].reshape(-1)), (mode, gen, l, r, s, q, "v")


@pytest.mark.parametrize("mode,codec", [("ring", "BINARY"), ("gather", "BINARY"), ("ring", "INT4"), ("gather", "INT8")])
def test_two_processes_two_generations_peer_to_peer(tmp_path, mode, codec):
    """Two rank processes on one GPU, packets read in place through IPC mappings; compact_reset between two generations: the flag words
    keep counting on the device (a host-side epoch restarted at 0 and every wait of the second generation passed at once)."""
    res = _spawn(W.w_xlayer, 2, tmp_path, codec, mode, -1, 2)
    for r in range(2):
        assert int(res[r]["n_ops"][0]) == 3 and int(res[r]["p2p"][0]) == 3 and int(res[r]["fell_back"][0]) == 0, "the peer-to-peer layer op was not taken"
        assert int(res[r]["validated"][0]) == 4, "the first four executions of every layer (both parities and their first reuse) are validated across the ranks"
    _check_states(res, codec, mode, 2, 2)


@pytest.mark.parametrize("world,mode,codec", [(4, "ring", "BINARY"), (3, "gather", "INT4")])
def test_more_than_two_rank_processes_peer_to_peer(tmp_path, world, mode, codec):
    """Three / four rank processes on one GPU: every layer launch publishes its word and awaits world - 1 peers' words inside the launch, its
    reconstruction workgroups read world - 1 other processes' packets in place (which peer's packet goes to which state, and which word is
    whose, only shows with more than one peer).  States of every rank's view of every rank == the oracle's chains."""
    res = _spawn(W.w_xlayer, world, tmp_path, codec, mode, -1, 1)
    for r in range(world):
        assert int(res[r]["n_ops"][0]) == 3 and int(res[r]["p2p"][0]) == 3 and int(res[r]["fell_back"][0]) == 0, "the peer-to-peer layer op was not taken"
    _check_states(res, codec, mode, world, 1)


@pytest.mark.parametrize("poison", [0, 2])
def test_poisoned_reconstruction_sends_every_rank_to_the_next_transport(tmp_path, poison):
    """A reconstruction that differs from its owner's state (what a stale cache line would produce - from the SECOND use of an address on:
    validated execution 2 is the first reuse of parity 0) fails the validation on every rank together: states restored and re-synchronised
    from their owners, the arena marked bad, the execution repeated on the next transport (here torch.distributed over gloo: RCCL refuses
    two ranks on one device) - and nothing of it shows in the states."""
    res = _spawn(W.w_xlayer, 2, tmp_path, "BINARY", "ring", poison, 1)
    for r in range(2):
        assert int(res[r]["p2p"][0]) == 0 and int(res[r]["fell_back"][0]) >= 1, (r, res[r]["p2p"], res[r]["fell_back"])
    _check_states(res, "BINARY", "ring", 2, 1)


def _chain_no_ef(codec, xs):
    """Without error feedback (main.py:227-243, `else x`): the owner keeps the activation, a peer keeps its reconstruction
    recon <- recon + decode(compress(x - previous x)).  Returns (owner states, peer states) per step as bit patterns."""
    name, param = ONAME[codec]
    shape2 = (-1, xs[0].shape[-2] * xs[0].shape[-1])
    own = xs[0].numpy().reshape(shape2).copy()
    peer = own.copy()
    outs_own, outs_peer = [R.bits(own).copy()], [R.bits(peer).copy()]
    for x in xs[1:]:
        x2 = x.numpy().reshape(own.shape)
        pkt, _ = R.residual_compress(name, x2, own, param)
        peer = R.residual_decompress(name, pkt, peer, own.shape[0], own.shape[1], param)
        own = x2.copy()
        outs_own.append(R.bits(own).copy())
        outs_peer.append(R.bits(peer).copy())
    return outs_own, outs_peer


def test_two_processes_without_error_feedback_keep_the_peer_to_peer_transport(tmp_path):
    """own_update "x" (ring mode, error_feedback=False): the owner's state becomes the activation, its peers hold previous state + decoded
    packet - the validation compares the peers' reconstructions with THAT (round 4 compared them with the owner's state and sent every
    such layer to the fall-back transport on its first execution)."""
    codec, STEPS, shape = "INT4", 6, (1, 64, 8, 64)
    res = _spawn(W.w_xlayer, 2, tmp_path, codec, "ring", -1, 1, STEPS, False)
    for r in range(2):
        assert int(res[r]["n_ops"][0]) == 3 and int(res[r]["p2p"][0]) == 3 and int(res[r]["fell_back"][0]) == 0, (res[r]["p2p"], res[r]["fell_back"])
        assert int(res[r]["validated"][0]) == 4
    for l in range(3):
        for q in range(2):
            for nm, seed in (("k", 17), ("v", 27)):
                own, peer = _chain_no_ef(codec, W.drift(seed + 10 * l + q, shape, STEPS))
                for r in range(2):
                    for s in range(STEPS):
                        want = own[s] if r == q else peer[s]
                        assert np.array_equal(res[r][f"g0/s{s}/l{l}/{nm}{q}"].reshape(-1), want.reshape(-1)), (l, q, nm, r, s)


@pytest.mark.parametrize("late_step", [3, 6])
def test_a_rank_three_seconds_late_is_waited_for_inside_the_launch(tmp_path, late_step):
    """Rank 1 arrives 3 s late at layer 1 of a step (3: a validated execution; 6: the steady path, nothing synchronises the host): rank 0's
    layer launch waits for rank 1's word INSIDE the kernel - on the wall clock against the 5 s gate timeout (an iteration count gave up
    after ~2 s and the reconstruction groups went on without the packets).  The run completes on the peer-to-peer transport, states ==
    oracle."""
    STEPS = 8
    res = _spawn(W.w_xlayer, 2, tmp_path, "BINARY", "ring", -1, 1, STEPS, True, (late_step, 3.0, 1))
    for r in range(2):
        assert int(res[r]["p2p"][0]) == 3 and int(res[r]["fell_back"][0]) == 0, (r, res[r]["p2p"], res[r]["fell_back"])
    _check_states(res, "BINARY", "ring", 2, 1, STEPS=STEPS)


def test_a_wait_that_times_out_stores_nothing_and_the_group_recovers(tmp_path):
    """Gate timeout 1 s, rank 1 arrives 2.5 s late in the steady path: rank 0's reconstruction groups give up WITHOUT storing (its copies of
    rank 1's states stay a delta behind), the next native call reports CFX_ERR_GATE, the ranks agree on it at a step boundary, every
    layer validates again, the stale layer fails, is re-synchronised from its owner and continues on the next transport.  The last
    steps' states == oracle again on both ranks."""
    STEPS = 12
    res = _spawn(W.w_xlayer, 2, tmp_path, "BINARY", "ring", -1, 1, STEPS, True, (6, 2.5, 1), 1000)
    for r in range(2):
        assert int(res[r]["fell_back"][0]) >= 1, (r, res[r]["p2p"], res[r]["fell_back"])
    shape = (1, 64, 8, 64)
    for l in range(3):
        for q in range(2):
            for nm, seed in (("k", 17), ("v", 27)):
                want = _chain("BINARY", W.drift(seed + 10 * l + q, shape, STEPS))
                for r in range(2):
                    for s in range(6):                                  # before the late arrival: the oracle's chains
                        assert np.array_equal(res[r][f"g0/s{s}/l{l}/{nm}{q}"].reshape(-1), want[s].reshape(-1)), (l, q, nm, r, s)
                for s in (STEPS - 2, STEPS - 1):
                    # afterwards: both ranks hold the SAME state of every shard again (a copy that missed an update was replaced by its
                    # owner's state) ...
                    assert np.array_equal(res[0][f"g0/s{s}/l{l}/{nm}{q}"], res[1][f"g0/s{s}/l{l}/{nm}{q}"]), (l, q, nm, s)
                    if q == 1:      # ... and rank 1's own chain never noticed anything (rank 0's own error feedback missed the update whose
                        #                launch gave up: its chain continues from the state it really holds)
                        assert np.array_equal(res[1][f"g0/s{s}/l{l}/{nm}{q}"].reshape(-1), want[s].reshape(-1)), (l, q, nm, s)


def test_a_timeout_in_the_last_launch_before_a_revalidation_step_is_not_swallowed(tmp_path):
    """the arena is not uncached, so every 6th execution of a layer validates itself.  Rank 1 arrives 2.5 s late (gate timeout
    1 s) at the LAST layer of the step before such a step: rank 0's launch of that layer gives up and stores nothing, and the next native
    call - layer 0's VALIDATED execution - finds the context's error word: it returns without launching and clears the word.  The
    validation fails for layer 0; the layer that really missed its update never owed the group a proof (`recheck` 0).  Every layer must
    nevertheless leave the failed transport with its peers' copies taken from their owners: at the end both ranks hold the SAME state of
    every shard of every layer."""
    STEPS, EVERY = 16, 6
    res = _spawn(W.w_xlayer, 2, tmp_path, "BINARY", "ring", -1, 1, STEPS, True, ("before_revalidation", 2.5, 2), 1000, EVERY)
    assert 0 < int(res[1]["late_at_step"][0]) < STEPS - 3, "the run never reached a step before a periodic re-validation"
    for r in range(2):
        assert int(res[r]["fell_back"][0]) >= 1, (r, res[r]["p2p"], res[r]["fell_back"])
    shape = (1, 64, 8, 64)
    for l in range(3):
        for q in range(2):
            for nm, seed in (("k", 17), ("v", 27)):
                want = _chain("BINARY", W.drift(seed + 10 * l + q, shape, STEPS))
                for s in (STEPS - 2, STEPS - 1):
                    assert np.array_equal(res[0][f"g0/s{s}/l{l}/{nm}{q}"], res[1][f"g0/s{s}/l{l}/{nm}{q}"]), (l, q, nm, s)
                    if q == 1:
                        assert np.array_equal(res[1][f"g0/s{s}/l{l}/{nm}{q}"].reshape(-1), want[s].reshape(-1)), (l, q, nm, s)


def test_twenty_resets_leave_device_memory_flat(tmp_path):
    res = _spawn(W.w_xlayer, 2, tmp_path, "BINARY", "ring", -1, 22)
    for r in range(2):
        free = res[r]["free"]
        assert int(res[r]["p2p"][0]) == 3
        assert abs(int(free[-1]) - int(free[1])) <= (8 << 20), f"device memory moved by {int(free[1]) - int(free[-1])} B over 20 compact_reset()s"
