"""Schedule-level PARITY on the GPU (-m gpu): the exchange schedules of the reference - compact_all_gather
(xfuser/compact/main.py:390-420), the ring forward in the relay and gather schedules (ring.py:120-275), the patch-gather
forward (patchpara/fwd.py:20-236) and the xFuserLongContextAttention hook (attn_layer.py:55-65,173-210) - run as TWO PROCESSES
sharing GPU 0 on the real HIP kernels (collectives over gloo), and are compared with
  * the REFERENCE's own committed 2-rank trace (golden group G10, tests/golden/make_golden.py),
  * the oracle's replay of the state machine on the same inputs (bit for bit: states of every rank for every shard),
  * full attention over the K,V a rank actually holds (rtol / atol of the reference's tests/core/test_ring_flash_attn.py:75-101).
The worker bodies are the ones the CPU (gloo + oracle stand-in) tests run: tests/_dist_workers.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import _dist_workers as W
import _golden as G
from oracle import ref_np as R

pytestmark = pytest.mark.gpu
ONAME = {"BINARY": ("binary", 0), "INT2": ("int2", 0), "INT4": ("int4", 0), "INT8": ("int8", 0), "SPARSE": ("topk", 8)}


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _entry(rank, fn_name, world, port, out, args):
    W.run(getattr(W, fn_name), rank, world, port, out, *args, device="cuda")


def _spawn(fn, world, tmp_path, *args):
    out = str(tmp_path / "res")
    for attempt in range(3):
        try:
            mp.start_processes(_entry, args=(fn.__name__, world, _port(), out, args), nprocs=world, join=True, start_method="spawn")
            break
        except Exception as e:  # noqa: BLE001
            # the port _port() found free can be taken again before rank 0 binds it (other rank processes of the suite come and go)
            if "EADDRINUSE" not in str(e) or attempt == 2:
                raise
    return [dict(np.load(out + f".r{r}.npz")) for r in range(world)]


def _chain(codec, xs):
    """Oracle replay: states after each step of WARMUP, codec, codec, ... on the drift sequence xs (uint16 bit patterns out)."""
    name, param = ONAME[codec]
    state = xs[0].numpy().reshape(-1, xs[0].shape[-1] if xs[0].dim() == 2 else xs[0].shape[-2] * xs[0].shape[-1]).copy()
    out = [R.bits(state).copy()]
    for x in xs[1:]:
        x2 = x.numpy().reshape(state.shape)
        _, state = R.residual_compress(name, x2, state, param)
        out.append(R.bits(state).copy())
    return out


@pytest.mark.parametrize("codec", ["BINARY", "INT2", "INT4", "SPARSE"])
def test_compact_all_gather_on_gpu_vs_oracle_and_reference_trace(tmp_path, codec):
    res = _spawn(W.w_all_gather, 2, tmp_path, codec)
    N, C = 32, 256
    want = [_chain(codec, W.drift(100 + i, (N, C), 4)) for i in range(2)]
    for r in range(2):
        assert int(res[r]["passed_count"][0]) == 1
        for t in range(4):
            for i in range(2):
                # every rank's reconstruction of shard i == the oracle's error-feedback state, bit for bit
                assert np.array_equal(res[r][f"t{t}/out{i}"], want[i][t].reshape(N, C)), (codec, r, t, i)
    if codec in ("BINARY", "INT2"):
        fn, name = "g10_allgather_2rank_eager.npz", codec.lower()
        for r in range(2):
            for t in range(4):
                assert np.array_equal(G.get(fn, f"{name}/r{r}/t{t}/x"), res[r][f"t{t}/x"]), "input recipe drifted"
                for i in range(2):
                    gold, mine = G.get(fn, f"{name}/r{r}/t{t}/out{i}"), res[r][f"t{t}/out{i}"]
                    if t == 0:
                        assert np.array_equal(gold, mine)
                    else:
                        assert G.rel_err(mine, gold) < 1e-3, (name, r, t, i, G.rel_err(mine, gold))     # the north-star tolerance


@pytest.mark.parametrize("codec,joint", [("BINARY", "none"), ("INT2", "front"), ("BINARY", "rear"), ("INT8", "none"), ("SPARSE", "none")])
def test_ring_forward_on_gpu_vs_oracle_and_full_attention(tmp_path, codec, joint):
    (tmp_path / "relay").mkdir()
    (tmp_path / "gather").mkdir()
    relay = _spawn(W.w_ring, 2, tmp_path / "relay", "relay", codec, joint)
    gather = _spawn(W.w_ring, 2, tmp_path / "gather", "gather", codec, joint)
    shape = (1, 64, 8, 64)
    want_k = [_chain(codec, W.drift(17 + q, shape, 3)) for q in range(2)]
    want_v = [_chain(codec, W.drift(27 + q, shape, 3)) for q in range(2)]
    for sched, res in (("relay", relay), ("gather", gather)):
        for r in range(2):
            assert int(res[r]["passed_count"][0]) == 3, sched
            for s in range(3):
                for q in range(2):
                    # what rank r holds for rank q's shard == the oracle's replay of q's error-feedback state
                    assert np.array_equal(res[r][f"s{s}/state_k_{q}"].reshape(-1), want_k[q][s].reshape(-1)), (sched, r, s, q, "k")
                    assert np.array_equal(res[r][f"s{s}/state_v_{q}"].reshape(-1), want_v[q][s].reshape(-1)), (sched, r, s, q, "v")
                # block-wise attention with running log-sum-exp == ONE attention over the K,V this rank holds
                np.testing.assert_allclose(res[r][f"s{s}/out"], res[r][f"s{s}/ref_out"], rtol=2e-3, atol=2e-3)
                np.testing.assert_allclose(res[r][f"s{s}/lse"], res[r][f"s{s}/ref_lse"], rtol=1e-3, atol=1e-3)
    for r in range(2):
        for s in range(3):
            np.testing.assert_allclose(relay[r][f"s{s}/out"], gather[r][f"s{s}/out"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("codec", ["BINARY", "INT8", "SPARSE"])
def test_ring_gather_peer_to_peer_on_gpu_vs_oracle(tmp_path, codec):
    """CFX_RING_P2P=1: the gather schedule of compact_fwd with NO collective - every rank's packets stay in cfx_ipc_alloc memory, the other
    process's per-peer reconstruction launches read them in place behind a publish-and-wait op (cfx_plan_add_p2p_sync).  Two processes on
    one GPU: the states every rank holds for every shard equal the oracle's replay, and the attention equals one attention over them."""
    os.environ["CFX_RING_P2P"] = "1"
    try:
        res = _spawn(W.w_ring, 2, tmp_path, "gather", codec, "none")
    finally:
        os.environ.pop("CFX_RING_P2P", None)
    shape = (1, 64, 8, 64)
    want_k = [_chain(codec, W.drift(17 + q, shape, 3)) for q in range(2)]
    want_v = [_chain(codec, W.drift(27 + q, shape, 3)) for q in range(2)]
    for r in range(2):
        assert int(res[r]["p2p"][0]) == 1, "the peer-to-peer exchange was not taken"
        assert int(res[r]["passed_count"][0]) == 3
        for s in range(3):
            for q in range(2):
                assert np.array_equal(res[r][f"s{s}/state_k_{q}"].reshape(-1), want_k[q][s].reshape(-1)), (r, s, q, "k")
                assert np.array_equal(res[r][f"s{s}/state_v_{q}"].reshape(-1), want_v[q][s].reshape(-1)), (r, s, q, "v")
            np.testing.assert_allclose(res[r][f"s{s}/out"], res[r][f"s{s}/ref_out"], rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(res[r][f"s{s}/lse"], res[r][f"s{s}/ref_lse"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("mode", ["sync", "async", "compact"])
def test_patch_gather_forward_on_gpu(tmp_path, mode):
    res = _spawn(W.w_patch, 2, tmp_path, mode)
    for r in range(2):
        for s in range(4):
            np.testing.assert_allclose(res[r][f"s{s}/out"], res[r][f"s{s}/ref_out"], rtol=2e-3, atol=2e-3)
    if mode == "compact":
        want = [_chain("INT2", W.drift(17 + q, (1, 64, 8, 64), 4)) for q in range(2)]
        for r in range(2):
            for s in range(1, 4):
                for q in range(2):
                    assert np.array_equal(res[r][f"s{s}/state_k_{q}"].reshape(-1), want[q][s].reshape(-1)), (r, s, q)


def test_displaced_compressed_patch_gather_on_gpu_vs_oracle(tmp_path):
    res = _spawn(W.w_patch_displaced, 2, tmp_path)
    want_k = [_chain("BINARY", W.drift(17 + q, (1, 64, 8, 64), 5)) for q in range(2)]
    want_v = [_chain("BINARY", W.drift(27 + q, (1, 64, 8, 64), 5)) for q in range(2)]
    for r in range(2):
        for s in range(5):
            for q in range(2):
                assert np.array_equal(res[r][f"sync/s{s}/state_k_{q}"].reshape(-1), want_k[q][s].reshape(-1)), ("sync", r, s, q)
                assert np.array_equal(res[r][f"sync/s{s}/state_v_{q}"].reshape(-1), want_v[q][s].reshape(-1)), ("sync", r, s, q)
                if s >= 1:   # while step s is in flight the displaced run holds the states of step s-1
                    assert np.array_equal(res[r][f"disp/s{s}/state_k_{q}"].reshape(-1), want_k[q][s - 1].reshape(-1)), ("disp", r, s, q)
            if s >= 1:
                np.testing.assert_allclose(res[r][f"disp/s{s}/out"], res[r][f"disp/s{s}/ref_out"], rtol=2e-3, atol=2e-3)
        for q in range(2):
            assert np.array_equal(res[r][f"disp/final/state_k_{q}"].reshape(-1), want_k[q][4].reshape(-1))


def test_displaced_low_rank_patch_gather_on_gpu(tmp_path):
    """BASELINE config 5 as written: DistriFusion-style displaced patch gather WITH the low-rank residual codec (the reference
    forbids async + compact, df_utils.py:13-16: an extension).  Every rank applies every packet once and in order: states are
    identical on both ranks, lag the synchronous run by exactly one step and end equal to it; the output is full attention over
    own-fresh + peers-one-step-stale K,V."""
    res = _spawn(W.w_patch_displaced, 2, tmp_path, "LOW_RANK")
    for r in range(2):
        for s in range(1, 5):
            np.testing.assert_allclose(res[r][f"disp/s{s}/out"], res[r][f"disp/s{s}/ref_out"], rtol=2e-3, atol=2e-3)
            for q in range(2):
                assert np.array_equal(res[0][f"sync/s{s}/state_k_{q}"], res[1][f"sync/s{s}/state_k_{q}"])
                assert np.array_equal(res[r][f"disp/s{s}/state_k_{q}"], res[r][f"sync/s{s - 1}/state_k_{q}"])
        for q in range(2):
            assert np.array_equal(res[r][f"disp/final/state_k_{q}"], res[r][f"sync/final/state_k_{q}"])
            assert np.array_equal(res[0][f"disp/final/state_k_{q}"], res[1][f"disp/final/state_k_{q}"])
        # the codec is lossy but tracks: relative error of the low-rank reconstruction of the peer's K after 4 steps
        k4 = res[1 - r]["s4/k"].view(np.float16).astype(np.float32).reshape(-1)
        st = res[r][f"sync/s4/state_k_{1 - r}"].view(np.float16).astype(np.float32).reshape(-1)
        assert np.linalg.norm(k4 - st) / np.linalg.norm(k4) < 0.25


@pytest.mark.parametrize("ulysses,ring,compact_on", [(1, 2, True), (2, 1, True), (1, 2, False)])
def test_long_context_attention_hook_on_gpu(tmp_path, ulysses, ring, compact_on):
    res = _spawn(W.w_hook_layer, 2, tmp_path, ulysses, ring, compact_on)
    for r in range(2):
        for li in range(2):
            np.testing.assert_allclose(res[r][f"s0/l{li}/out"], res[r][f"s0/l{li}/ref"], rtol=2e-3, atol=2e-3)
            if not compact_on or ring == 1:
                np.testing.assert_allclose(res[r][f"s1/l{li}/out"], res[r][f"s1/l{li}/ref"], rtol=2e-3, atol=2e-3)
            else:
                # the compressed ring step: parity with ONE attention over the K,V the rank holds (own exact, peers' reconstructions) ...
                np.testing.assert_allclose(res[r][f"s1/l{li}/out"], res[r][f"s1/l{li}/ref_held"], rtol=2e-3, atol=2e-3)
                # ... the states against the oracle's replay, bit for bit (every rank's view of every shard, both layers) ...
                for q in range(2):
                    want_k = _chain("BINARY", W.drift(17 + q, (1, 64, 8, 64), 2))
                    want_v = _chain("BINARY", W.drift(27 + q, (1, 64, 8, 64), 2))
                    for s_ in range(2):
                        assert np.array_equal(res[r][f"s{s_}/l{li}/state_k_{q}"].reshape(-1), want_k[s_].reshape(-1)), (r, li, s_, q, "k")
                        assert np.array_equal(res[r][f"s{s_}/l{li}/state_v_{q}"].reshape(-1), want_v[s_].reshape(-1)), (r, li, s_, q, "v")
                # ... and only a sanity bound against the UNCOMPRESSED full attention (1-bit residuals of a 0.1 drift)
                assert np.abs(res[r][f"s1/l{li}/out"] - res[r][f"s1/l{li}/ref"]).max() < 0.15
        if compact_on:
            want = {f"{l}-{q}-{t}" for l in range(2) for q in range(ring) for t in "kv"}
            assert set(res[r]["keys"].tolist()) == want


# ---- the native per-layer chain (libcfx communicator, exchange stream, steady-state lane) with 4 logical ranks looped back ----
def test_native_exchange_chain_in_compact_fwd_vs_oracle(tmp_path):
    """compact_fwd's gather schedule with the layer's whole exchange issued natively (cfx_plan_run_async / cfx_plan_join, the
    library-owned communicator = tests/fake_rccl in loop-back mode: every logical peer is this rank).  Every peer state and the
    rank's own error-feedback state must equal the oracle's replay of the rank's own shard, over enough steps to cover the
    general path (binds the plan), and the steady-state lane."""
    import ctypes
    import sys
    import tempfile
    os.environ["CFX_FAKE_RCCL_MODE"] = "loopback"
    from compactfusion_amd import _lib, codecs as K, exchange
    from compactfusion_amd.collector import collector
    from compactfusion_amd.compact import ring, main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl"))
    import build as fake_build
    sys.path.pop(0); sys.modules.pop("build", None)
    fake = fake_build.build()
    lib = _lib.load()
    Wl, L, STEPS = 4, 3, 5
    B, S, Hh, Dh = 1, 64, 8, 64

    class LoopComm:
        def __init__(self, group, device):
            ctx = K.context(device)
            assert lib.cfx_rccl_load(fake.encode()) == 0
            uid = ctypes.create_string_buffer(128)
            assert lib.cfx_comm_unique_id(ctx, uid) == 0
            self.handle = lib.cfx_comm_create(ctx, uid, Wl, 0)
            assert self.handle
    saved = (ring.dist.get_rank, ring.dist.get_world_size, ring.dist.all_gather_into_tensor)
    ring.dist.get_rank = lambda g=None: 0
    ring.dist.get_world_size = lambda g=None: Wl
    ring.dist.all_gather_into_tensor = lambda recv, send, group=None: recv.view(Wl, -1).copy_(send.view(1, -1).expand(Wl, -1))
    collector.init(collector.Collector(tempfile.mkdtemp(), enabled=False))
    try:
        for xmode in ("chain", "side"):
            os.environ["CFX_RING_EXCHANGE"] = "native"
            os.environ["CFX_RING_EXCHANGE_STREAM"] = xmode
            exchange.set_comm_factory(LoopComm)
            ring._xbuf.clear(); ring._steady.clear()
            cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T.BINARY, comp_rank=-1,
                                          residual=1, ef=True, fastpath=True))
            qs = [W.drift(7 + l, (B, S, Hh, Dh), STEPS) for l in range(L)]
            ks = [W.drift(17 + l, (B, S, Hh, Dh), STEPS) for l in range(L)]
            vs = [W.drift(27 + l, (B, S, Hh, Dh), STEPS) for l in range(L)]
            steady_hits = 0
            for step in range(STEPS):
                cm.compact_set_step(step)
                for l in range(L):
                    st = ring._steady.get((l, None))
                    out, lse, _ = ring.compact_fwd(qs[l][step].cuda(), ks[l][step].cuda(), vs[l][step].cuda(), causal=False, mod_idx=l, current_iter=step)
                    steady_hits += int(st is not None)
            torch.cuda.synchronize()
            assert all(ex.plan is not None for ex in ring._xbuf.values() if ex.sig is not None), "the native plan was not used"
            assert steady_hits >= L * (STEPS - 2), "the steady-state lane was not taken"
            for l in range(L):
                wk, wv = _chain("BINARY", ks[l])[-1], _chain("BINARY", vs[l])[-1]
                for r in range(Wl):
                    gk = cm.compact_cache().get_base(f"{l}-{r}-k")
                    gv = cm.compact_cache().get_base(f"{l}-{r}-v")
                    assert np.array_equal(W.bits(gk).reshape(-1), wk.reshape(-1)), (xmode, l, r, "k")
                    assert np.array_equal(W.bits(gv).reshape(-1), wv.reshape(-1)), (xmode, l, r, "v")
            # the output of the last step: every block sees this rank's K,V (local exact, peers reconstructed)
            from compactfusion_amd.compact.attention import block_attention
            l = L - 1
            kk = [ks[l][-1].cuda()] + [cm.compact_cache().get_base(f"{l}-{r}-k").view(B, S, Hh, Dh) for r in range(1, Wl)]
            vv = [vs[l][-1].cuda()] + [cm.compact_cache().get_base(f"{l}-{r}-v").view(B, S, Hh, Dh) for r in range(1, Wl)]
            ref, _ = block_attention(qs[l][-1].cuda(), torch.cat(kk, 1), torch.cat(vv, 1), 0.0, None, causal=False)
            torch.testing.assert_close(out.float(), ref.float(), rtol=2e-3, atol=2e-3)
    finally:
        ring.dist.get_rank, ring.dist.get_world_size, ring.dist.all_gather_into_tensor = saved
        exchange.set_comm_factory(None)
        os.environ.pop("CFX_RING_EXCHANGE", None); os.environ.pop("CFX_RING_EXCHANGE_STREAM", None)
        ring._xbuf.clear(); ring._steady.clear()


# ---- observability and the quantised cache on GPU tensors ---------------------------------------------------------------------
@pytest.mark.parametrize("residual", [0, 1, 2])
def test_stats_logger_records_on_gpu_match_reference_golden(residual, monkeypatch):
    """StatsLogger fed GPU tensors reproduces the REFERENCE's records (golden G11, tests/golden/make_golden_stats.py)."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    gold = np.load(os.path.join(here, "golden", "g11_stats.npz"))
    spec = importlib.util.spec_from_file_location("make_golden_stats", os.path.join(here, "golden", "make_golden_stats.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    from compactfusion_amd.compact import stats as S
    monkeypatch.setattr(S, "CALC_SIMILARITY", True)
    monkeypatch.setattr(S, "CALC_MORE_SIMILARITY", True)
    S.stats_clear()
    try:
        cu = lambda t: None if t is None else t.cuda()        # noqa: E731
        for key, base, dbase, x, recv, comp in gen.inputs(residual):
            S.log(key, cu(base), cu(dbase), cu(x), cu(recv), cu(comp), residual)
        lg = S.stats_log()
        lowrank = gen.FIELDS.index("delta_before_feedback_lowrank_similarity")
        cols = [i for i in range(len(gen.FIELDS)) if i != lowrank]
        for k in gen.KEYS:
            want = gold[f"r{residual}/{k}"]
            got = np.array([[np.nan if row[f] is None else float(row[f]) for f in gen.FIELDS] for row in lg.stats[k]])
            assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want))
            np.testing.assert_allclose(got[:, cols], want[:, cols], rtol=1e-3, atol=1e-5, equal_nan=True)      # norms are fp16 results: one ulp = 4.9e-4
        assert [lg.total_original_volume, lg.total_compressed_volume] == list(gold[f"r{residual}/volumes"])
    finally:
        S.stats_clear()


def test_profiler_scopes_time_gpu_work_with_hip_events():
    """Profiler scopes (xfuser/prof.py) bracket stream work with HIP events: a scope around a known amount of GPU work reports a
    time consistent with torch's own event timing, on the current and on a side stream."""
    from compactfusion_amd.prof import Profiler, prof_summary
    prof = Profiler.instance()
    prof.enable(); prof.reset()
    a = torch.randn(4096, 4096, device="cuda")
    side = torch.cuda.Stream()
    try:
        for _ in range(3):
            with Profiler.scope("total"):
                with Profiler.scope("matmul"):
                    for _ in range(10):
                        a @ a
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    with Profiler.scope("side.matmul", stream=side):
                        a @ a
                torch.cuda.current_stream().wait_stream(side)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            a @ a
        e1.record(); torch.cuda.synchronize()
        ref_ms = e0.elapsed_time(e1)
        totals, means = prof.get_all_elapsed_times()
        assert set(totals) >= {"total", "matmul", "side.matmul"}
        assert 0.5 * ref_ms < means["matmul"] < 2.0 * ref_ms, (means, ref_ms)
        assert 0.02 * ref_ms < means["side.matmul"] < 0.5 * ref_ms
        assert means["total"] >= means["matmul"]
        lines = prof_summary(prof, rank=0)
        assert any("[matmul]" in ln for ln in lines)
    finally:
        prof.reset(); prof.disable()


def test_quantized_cache_on_gpu_vs_oracle(monkeypatch, tmp_path):
    """CompactCache(quantize=True) on the native INT8 codec: stored state == oracle int8 packet, handed out == oracle dequant;
    the 1-bit state machine on top keeps sender and receiver bit-identical and equal to the oracle's replay."""
    from compactfusion_amd.collector import collector
    collector.init(collector.Collector(str(tmp_path), enabled=False))
    from compactfusion_amd.compact import utils as U, main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    monkeypatch.setattr(U, "ALLOW_DEPRECATED", True)
    N, C = 64, 256
    c = U.CompactCache(quantize=True)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, C, generator=g).half()
    c.put("1-0-k", x.cuda(), None)
    want = R.decompress("int8", R.compress("int8", W.bits(x).reshape(N, C), None)[0], N, C)
    assert np.array_equal(W.bits(c.get_base("1-0-k")), R.bits(want))
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T.BINARY, comp_rank=-1,
                                  residual=1, ef=True, fastpath=True, quantized_cache=True))
    xs = W.drift(5, (N, C), 4)
    state = None
    for step, xt in enumerate(xs):
        cm.compact_set_step(step)
        typ = T.WARMUP if step == 0 else T.BINARY
        pkt = cm.compact_compress("5-0-k", xt.cuda(), typ, update_cache=True)
        cm.compact_decompress("5-1-k", pkt.clone(), typ, xt.shape, update_cache=True)
        s_state, r_state = W.bits(cm.compact_cache().get_base("5-0-k")), W.bits(cm.compact_cache().get_base("5-1-k"))
        assert np.array_equal(s_state, r_state), step
        # oracle replay: the state is stored through int8 every step
        q8 = lambda a: R.decompress("int8", R.compress("int8", R.bits(a).reshape(N, C), None)[0], N, C)      # noqa: E731
        if step == 0:
            state = q8(xt.numpy())
        else:
            _, nb = R.residual_compress("binary", xt.numpy(), state, 0)
            state = q8(nb)
        assert np.array_equal(s_state.reshape(N, C), R.bits(state)), step
