"""world_size-2 tests of the exchange schedules on CPU (gloo): compact_all_gather, the ring forward in both
schedules, the patch-gather forward in its three modes, and the committed 2-rank golden trace (G10).
Kernels are replaced by the oracle stand-in (tests/_oracle_backend.py); the collectives are real."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import _dist_workers as W
import _golden as G


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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


def _entry(rank, fn_name, world, port, out, args):
    W.run(getattr(W, fn_name), rank, world, port, out, *args)


@pytest.mark.parametrize("codec", ["BINARY", "INT2", "INT4", "SPARSE"])
def test_compact_all_gather_2rank(tmp_path, codec):
    res = _spawn(W.w_all_gather, 2, tmp_path, codec)
    for t in range(4):
        for i in range(2):
            # every rank reconstructs the same shard i (bit-identical state on all ranks)
            assert np.array_equal(res[0][f"t{t}/out{i}"], res[1][f"t{t}/out{i}"]), (t, i)
        # WARMUP step returns the raw shards
        if t == 0:
            for i in range(2):
                assert np.array_equal(res[0][f"t0/out{i}"], res[i]["t0/x"])
    for r in range(2):
        assert int(res[r]["passed_count"][0]) == 1
    # reconstruction tracks the input: relative error stays bounded with error feedback
    x3 = res[1]["t3/x"].view(np.float16).astype(np.float32)
    o3 = res[0]["t3/out1"].view(np.float16).astype(np.float32)
    assert np.linalg.norm(x3 - o3) / np.linalg.norm(x3) < 0.25


@pytest.mark.parametrize("name", ["binary", "int2"])
def test_g10_golden_all_gather_trace(tmp_path, name):
    """Committed trace of the REFERENCE's compact_all_gather on 2 gloo ranks (tests/golden/make_golden.py g10)."""
    fn = "g10_allgather_2rank_eager.npz"
    if fn not in G.manifest():
        pytest.skip("G10 golden vectors not generated")
    res = _spawn(W.w_all_gather, 2, tmp_path, name.upper())
    for r in range(2):
        assert int(G.get(fn, f"{name}/r{r}/passed_count")[0]) == int(res[r]["passed_count"][0]) == 1
        for t in range(4):
            assert np.array_equal(G.get(fn, f"{name}/r{r}/t{t}/x"), res[r][f"t{t}/x"]), "input recipe drifted"
            for i in range(2):
                gold = G.get(fn, f"{name}/r{r}/t{t}/out{i}")
                mine = res[r][f"t{t}/out{i}"]
                if t == 0:
                    assert np.array_equal(gold, mine)
                else:
                    # scales may differ by one fp16 ulp from the reference's fp32-ordered sums (oracle docstring);
                    # the reconstructed activation must agree to the north-star 1e-3
                    assert G.rel_err(mine, gold) < 1e-3, (name, r, t, i, G.rel_err(mine, gold))


@pytest.mark.parametrize("codec,joint", [("BINARY", "none"), ("INT2", "front"), ("BINARY", "rear"), ("INT8", "none")])
def test_ring_forward_2rank(tmp_path, codec, joint):
    (tmp_path / "relay").mkdir()
    (tmp_path / "gather").mkdir()
    relay = _spawn(W.w_ring, 2, tmp_path / "relay", "relay", codec, joint)
    gather = _spawn(W.w_ring, 2, tmp_path / "gather", "gather", codec, joint)
    for r in range(2):
        assert int(relay[r]["passed_count"][0]) == 3 and int(gather[r]["passed_count"][0]) == 3
        for s in range(3):
            # block-wise ring attention == one attention over the K/V the rank actually holds (rtol/atol 1e-3, the
            # tolerance of the reference's tests/core/test_ring_flash_attn.py:75-101)
            np.testing.assert_allclose(relay[r][f"s{s}/out"], relay[r][f"s{s}/ref_out"], rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(relay[r][f"s{s}/lse"], relay[r][f"s{s}/ref_lse"], rtol=1e-3, atol=1e-3)
            # the MI355X-native gather schedule gives the same numbers as the reference's relay schedule
            assert np.array_equal(relay[r][f"s{s}/out"], gather[r][f"s{s}/out"])
            assert np.array_equal(relay[r][f"s{s}/lse"], gather[r][f"s{s}/lse"])
            for q in range(2):
                assert np.array_equal(relay[r][f"s{s}/state_k_{q}"], gather[r][f"s{s}/state_k_{q}"])
                # owner's error-feedback state == every peer's reconstruction of that shard
                assert np.array_equal(relay[r][f"s{s}/state_k_{q}"], relay[q][f"s{s}/own_k_state"])
        # step 0 is WARMUP: raw K/V travelled, so every state equals the owner's K
        for q in range(2):
            assert np.array_equal(relay[r][f"s0/state_k_{q}"].reshape(-1), relay[q]["s0/k"].reshape(-1))


@pytest.mark.parametrize("mode", ["sync", "async", "compact"])
def test_patch_gather_forward_2rank(tmp_path, mode):
    res = _spawn(W.w_patch, 2, tmp_path, mode)
    for r in range(2):
        for s in range(4):
            np.testing.assert_allclose(res[r][f"s{s}/out"], res[r][f"s{s}/ref_out"], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("codec", ["BINARY", "LOW_RANK"])
def test_displaced_compressed_patch_gather_2rank(tmp_path, codec):
    """Extension: with `PatchConfig(displaced_compact=True)` the packets are the
    synchronous run's packets, applied one step later - so peers are seen one step stale, the own shard fresh, and after
    the final flush every state equals the synchronous run's and is identical on every rank."""
    res = _spawn(W.w_patch_displaced, 2, tmp_path, codec)       # LOW_RANK: BASELINE config 5 as written (displaced + low-rank)
    for r in range(2):
        assert np.array_equal(res[r]["disp/s0/out"], res[r]["sync/s0/out"])          # warm-up step is synchronous
        for s in range(1, 5):
            np.testing.assert_allclose(res[r][f"disp/s{s}/out"], res[r][f"disp/s{s}/ref_out"], rtol=2e-3, atol=2e-3)
            for q in range(2):
                # while step s is in flight the states are the synchronous run's of step s-1
                assert np.array_equal(res[r][f"disp/s{s}/state_k_{q}"], res[r][f"sync/s{s - 1}/state_k_{q}"])
        for q in range(2):
            assert np.array_equal(res[r][f"disp/final/state_k_{q}"], res[r]["sync/final/state_k_" + str(q)])
            assert np.array_equal(res[0][f"disp/final/state_k_{q}"], res[1][f"disp/final/state_k_{q}"])


@pytest.mark.parametrize("ulysses,ring,compact_on", [(1, 2, True), (2, 1, True), (1, 2, False), (2, 1, False)])
def test_long_context_attention_hook_2rank(tmp_path, ulysses, ring, compact_on):
    """The attention layer that binds compact_fwd (attn_layer.py:55-65,173-210): layer indices, Ulysses all-to-all,
    ring / compressed-ring dispatch.  Step 0 is WARMUP (raw K/V) so it must equal full attention; step 1 is 1-bit
    compressed for the remote shard, so it equals full attention only up to the codec error."""
    res = _spawn(W.w_hook_layer, 2, tmp_path, ulysses, ring, compact_on)
    for r in range(2):
        for li in range(2):
            np.testing.assert_allclose(res[r][f"s0/l{li}/out"], res[r][f"s0/l{li}/ref"], rtol=2e-3, atol=2e-3)
            if not compact_on or ring == 1:
                np.testing.assert_allclose(res[r][f"s1/l{li}/out"], res[r][f"s1/l{li}/ref"], rtol=2e-3, atol=2e-3)
            else:
                err = np.abs(res[r][f"s1/l{li}/out"] - res[r][f"s1/l{li}/ref"]).max()
                assert err < 0.15, err
        if compact_on:
            keys = set(res[r]["keys"].tolist())
            want = {f"{l}-{q}-{t}" for l in range(2) for q in range(ring) for t in "kv"}     # "{layer}-{ring rank}-{k|v}"
            assert keys == want, keys


def test_displaced_gather_refuses_what_it_cannot_do(monkeypatch):
    """No silent synchronous fallback: a displaced compressed gather with an option the fused exchange does not cover raises."""
    import torch
    import _oracle_backend as OB
    OB.install(monkeypatch)
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: None, residual=1, ef=True, fastpath=True, comp_rank=-1,
                                  log_stats=True))
    k = torch.zeros(1, 16, 4, 32, dtype=torch.float16)
    with pytest.raises(NotImplementedError, match="log_compress_stats"):
        cm.compact_all_gather_kv("1-k", "1-v", k, k.clone(), T.BINARY, displaced=True)
