"""Host-side logic on CPU: config rules, enum, cache arena, (N,C) view rule, wire sizes, the residual / error-feedback
state machine and the function-level API mirrors.  The kernels are replaced by the oracle through a TEST-ONLY
stand-in (tests/_oracle_backend.py); the GPU tests (-m gpu) run the same API on the real kernels."""
import os
import numpy as np
import pytest
import torch

import _oracle_backend as OB
from oracle import ref_np as R


@pytest.fixture(autouse=True)
def _collector(tmp_path):
    from compactfusion_amd.collector import collector
    collector.init(collector.Collector(str(tmp_path), enabled=False))
    yield


@pytest.fixture
def cpu_kernels(monkeypatch):
    OB.install(monkeypatch)


def bits(t):
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16)


def test_enum_matches_reference_values():
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T
    want = {"WARMUP": "warmup", "SPARSE": "sparse", "BINARY": "binary", "INT2": "int2", "INT2_MINMAX": "int2-minmax",
            "INT4": "int4", "IDENTITY": "identity", "LOW_RANK": "low-rank", "LOW_RANK_Q": "low-rank-int4",
            "LOW_RANK_AWL": "low-rank-awl"}                       # utils.py:19-28
    for k, v in want.items():
        assert T[k].value == v


def test_config_rules():
    from compactfusion_amd.compact import CompactConfig, PatchConfig
    CompactConfig(enabled=True, residual=1, ef=True, fastpath=True, comp_rank=-1)
    for kw in (dict(residual=0, ef=True), dict(residual=2, ef=False), dict(residual=1, ef=False, fastpath=True),
               dict(residual=1, ef=True, fastpath=True, simulate=True), dict(residual=2, ef=True, fastpath=True),
               dict(residual=3)):
        with pytest.raises(AssertionError):
            CompactConfig(enabled=True, **kw)
    with pytest.raises(AssertionError):
        CompactConfig(enabled=False, override_with_patch_gather_fwd=True, patch_gather_fwd_config=PatchConfig(True, False, 1))
    with pytest.raises(AssertionError):
        CompactConfig(enabled=True, override_with_patch_gather_fwd=True)
    with pytest.raises(AssertionError):
        CompactConfig(enabled=True, patch_gather_fwd_config=PatchConfig(False, False, 0))
    with pytest.raises(AssertionError):
        PatchConfig(use_compact=True, async_comm=True, async_warmup=1)
    c = CompactConfig(enabled=True, compress_func=lambda l, s: __import__("compactfusion_amd").compact.COMPACT_COMPRESS_TYPE.BINARY,
                      residual=1, ef=True)
    assert c.get_compress_type() == "BINARY"
    assert CompactConfig().get_compress_type() == "NO_COMPACT"


def test_cache_arena_semantics():
    from compactfusion_amd.compact import CompactCache
    c = CompactCache()
    x = torch.randn(4, 8).half()
    c.put("0-0-k", x, None)
    b = c.get_base("0-0-k")
    assert torch.equal(b, x) and b.data_ptr() != x.data_ptr()        # copied into the arena
    p0 = b.data_ptr()
    c.put("0-0-k", torch.ones(4, 8).half(), x * 2)
    assert c.get_base("0-0-k").data_ptr() == p0                       # stable pointer
    assert torch.equal(c.get_delta_base("0-0-k"), x * 2)
    c.put("0-0-k", c.get_base("0-0-k"), None)                         # handing back the arena buffer is a no-op
    assert c.get_delta_base("0-0-k") is None and c.get_base("missing") is None


def test_collector_must_be_initialised():
    from compactfusion_amd.collector import collector
    from compactfusion_amd.compact import CompactCache
    collector.instance = None
    with pytest.raises(ValueError):
        CompactCache().put("0-0-k", torch.zeros(2, 8).half(), None)


def _drift(seed, N, C, T):
    g = torch.Generator().manual_seed(seed)
    cur = torch.randn(N, C, generator=g).half()
    out = []
    for _ in range(T):
        out.append(cur.contiguous())
        cur = (cur.float() + 0.1 * torch.randn(N, C, generator=g)).half()
    return out


CASES = [
    ("binary_fast", dict(residual=1, ef=True, fastpath=True, comp_rank=-1), "BINARY", 1, dict(codec="binary")),
    ("int2_fast", dict(residual=1, ef=True, fastpath=True, comp_rank=-1), "INT2", 1, dict(codec="int2")),
    ("binary_slow_noef", dict(residual=1, ef=False, comp_rank=-1), "BINARY", 1, dict(codec="binary")),
    ("binary_res0", dict(residual=0, ef=False, comp_rank=-1), "BINARY", 0, dict(codec="binary")),
    ("binary_res2", dict(residual=2, ef=True, comp_rank=-1, delta_decay_factor=0.5), "BINARY", 2, dict(codec="binary")),
    ("int8_ef", dict(residual=1, ef=True), "INT8", 1, dict(codec="int8")),
    ("int4_ef", dict(residual=1, ef=True), "INT4", 1, dict(codec="int4")),
    ("sparse8", dict(residual=1, ef=True, sparse_ratio=8), "SPARSE", 1, dict(codec="topk", param=8)),
    ("int4_sim", dict(residual=1, ef=True, simulate=True), "INT4", 1, dict(codec="int4", simulate=True)),
]


@pytest.mark.parametrize("name,kw,tname,nwarm,okw", CASES, ids=[c[0] for c in CASES])
def test_state_machine_equals_oracle(cpu_kernels, name, kw, tname, nwarm, okw):
    """compact_compress / compact_decompress (host state machine over the native codec boundary) follow the oracle's
    restatement of main.py:169-270, :322-388 step by step: same packets, same sender and receiver state."""
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    N, C = 64, 1024
    xs = _drift(11, N, C, 5)
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: None, **kw))
    orc_s = R.OracleCompact(residual=kw.get("residual", 0), ef=kw.get("ef", False), fastpath=kw.get("fastpath", False),
                            simulate=okw.get("simulate", False), param=okw.get("param", 0), decay=kw.get("delta_decay_factor"))
    orc_r = R.OracleCompact(residual=kw.get("residual", 0), ef=kw.get("ef", False), fastpath=kw.get("fastpath", False),
                            simulate=okw.get("simulate", False), param=okw.get("param", 0), decay=kw.get("delta_decay_factor"))
    skey, rkey = "0-0-k", "0-1-k"
    for t, x in enumerate(xs):
        x4 = x.view(1, N, 8, C // 8)           # >= 4-D input exercises the (N, C) view rule
        warm = t < nwarm
        typ = T.WARMUP if warm else T[tname]
        pkt = cm.compact_compress(skey, x4, typ, update_cache=True)
        want = orc_s.compress(skey, bits(x4).reshape(1, N, 8, C // 8), "warmup" if warm else okw["codec"], True)
        assert np.array_equal(bits(pkt).reshape(-1), want), f"{name} step {t}: packet"
        rec = cm.compact_decompress(rkey, pkt.clone(), typ, x4.shape, update_cache=True)
        assert rec.shape == x4.shape
        wrec = orc_r.decompress(rkey, want, "warmup" if warm else okw["codec"], x4.shape, True)
        assert np.array_equal(bits(rec).reshape(-1), R.bits(wrec).reshape(-1)), f"{name} step {t}: recon"
        if kw.get("residual", 0) != 0:
            assert np.array_equal(bits(cm.compact_cache().get_base(skey)), R.bits(orc_s.base[skey])), f"{name} step {t}: sender state"
            assert np.array_equal(bits(cm.compact_cache().get_base(rkey)), R.bits(orc_r.base[rkey])), f"{name} step {t}: receiver state"
            if kw.get("ef", False):
                assert np.array_equal(bits(cm.compact_cache().get_base(skey)), bits(cm.compact_cache().get_base(rkey)))
        if kw.get("residual", 0) == 2 and t >= 1:
            assert np.array_equal(bits(cm.compact_cache().get_delta_base(skey)), R.bits(orc_s.dbase[skey]))


def test_fastpath_rejects_other_codecs_and_missing_warmup(cpu_kernels):
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    cm.compact_init(CompactConfig(enabled=True, residual=1, ef=True, fastpath=True, comp_rank=-1))
    x = torch.randn(1, 8, 64).half()
    with pytest.raises(AssertionError):
        cm.compact_compress("0-0-k", x, T.INT4, update_cache=True)
    with pytest.raises(AssertionError):
        cm.compact_compress("0-0-k", x, T.BINARY, update_cache=True)      # no WARMUP yet -> no base
    cm.compact_init(CompactConfig(enabled=True, residual=1, ef=True, comp_rank=4))
    cm.compact_compress("0-0-k", x, T.WARMUP, update_cache=True)
    if not os.environ.get("COMPACT_ALLOW_DEPRECATED"):
        with pytest.raises(AssertionError):
            cm.compact_compress("0-0-k", x, T.BINARY, update_cache=True)      # rank != -1 is deprecated in the reference (main.py:188-189)
    cm.compact_init(CompactConfig(enabled=True, residual=1, ef=True))
    cm.compact_compress("0-0-k", x, T.WARMUP, update_cache=True)
    with pytest.raises(ValueError):
        cm.compact_compress("0-0-k", x, T.IDENTITY, update_cache=True)    # not a wire codec (slowpath.py:80-81)
    cm.compact_init(CompactConfig(enabled=False))
    with pytest.raises(AssertionError):
        cm.compact_compress("0-0-k", x, T.BINARY)


def test_reset_and_step(cpu_kernels):
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    cm.compact_init(CompactConfig(enabled=True, residual=1, ef=True, fastpath=True, comp_rank=-1))
    cm.compact_set_step(3)
    assert cm.compact_get_step() == 3
    cm.compact_compress("5-0-v", torch.randn(2, 8, 64).half(), T.WARMUP, update_cache=True)
    assert cm.compact_cache().get_base("5-0-v") is not None and cm.compact_get_current_cache_key() == "5-0-v"
    cm.compact_reset()
    assert cm.compact_cache().get_base("5-0-v") is None and cm.compact_get_step() is None


def test_function_level_mirrors(cpu_kernels):
    """fastpath / compress_quantize / compress_topk / slowpath signatures and return layouts (views into the packet)."""
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T
    from compactfusion_amd.compact import compress_quantize as Q, compress_topk as TK, fastpath as F, slowpath as S
    torch.manual_seed(42)
    N, C = 64, 256
    x = torch.randn(N, C).half()
    base = torch.randn_like(x) * 0.1
    p, u, v, nb = F.binary_quant_fastpath(x, base, -1, True)
    assert p.shape == (N, C // 8) and p.dtype == torch.uint8 and u.shape == (N, 1) and v.shape == (C, 1) and nb.shape == (N, C)
    op, ou, ov, onb = R.binary_quant_fastpath(bits(x), bits(base), -1, True)
    assert np.array_equal(p.numpy(), op) and np.array_equal(bits(u), R.bits(ou)) and np.array_equal(bits(v), R.bits(ov))
    assert np.array_equal(bits(nb), R.bits(onb))
    assert F.binary_quant_fastpath(x, base, -1, False)[3] is None
    rec = F.binary_dequant_fastpath(p, u, v, base)
    assert np.array_equal(bits(rec), R.bits(onb))
    sp, su, sv, snb = F.sim_binary_quant_fastpath(x, base, -1, True)
    assert torch.equal(sp, p) and torch.equal(su, u) and torch.equal(sv, v) and torch.equal(snb, nb)
    assert torch.equal(F.sim_binary_dequant_fastpath(p, u, v, base), rec)
    p2, u2, v2, nb2 = F.int2_quant_fastpath(x, base, True, -1)
    assert p2.shape == (N, C // 4) and torch.equal(F.int2_dequant_fastpath(p2, u2, v2, base), nb2)
    s2 = F.sim_int2_quant_fastpath(x, base, True)
    assert torch.equal(s2[0], p2) and torch.equal(s2[3], nb2)
    d = x - base
    q, s, z = Q.quantize_int8(d)
    assert q.dtype == torch.int8 and s.shape == (1, C) and z.dtype == torch.int16
    oq, os_, oz = R.quantize_int8(bits(d))
    assert np.array_equal(q.numpy(), oq) and np.array_equal(z.numpy(), oz)
    assert np.array_equal(bits(Q.dequantize_int8(q, s, z)), R.bits(R.dequantize_int8(oq, os_, oz)))
    q4, s4, m4 = Q.quantize_int4(d)
    assert q4.shape == (N // 2, C) and np.array_equal(bits(Q.dequantize_int4(q4, s4, m4)), R.bits(R.sim_int4(bits(d))))
    assert np.array_equal(bits(Q.sim_int4(d, 0)), R.bits(R.sim_int4(bits(d))))
    pk, ch, tk = Q.quantize_int2(d)
    assert pk.shape == (N, C // 4) and ch.shape == (1, C) and tk.shape == (N, 1)
    assert np.array_equal(bits(Q.sim_int2(d)), R.bits(R.dequantize_int2(*R.quantize_int2(bits(d)))))
    assert np.array_equal(bits(Q.sim_binary(d, -1)), R.bits(R.sim_binary(bits(d))))
    val, idx = TK.topk_compress(d.view(-1, 1024), 4)
    assert val.shape == (16, 256) and idx.shape == (16, 128) and idx.dtype == torch.uint8
    assert np.array_equal(bits(TK.topk_decompress(val, idx, 4).view(N, C)), R.bits(R.sim_topk(bits(d), 4)))
    assert np.array_equal(bits(TK.sim_topk(d, 4)), R.bits(R.sim_topk(bits(d), 4)))
    pkt = S.slowpath_compress(d, T.BINARY, rank=-1)
    assert pkt.numel() == N * C // 16 + N + C
    assert np.array_equal(bits(S.slowpath_decompress(pkt, (N, C), T.BINARY, rank=-1)), R.bits(R.sim_binary(bits(d))))
    assert np.array_equal(bits(S.sim_compress(d, T.SPARSE, sparse_ratio=8)), R.bits(R.sim_topk(bits(d), 8)))
    assert S.sim_compress(d, T.IDENTITY) is d
    assert np.array_equal(bits(S.sim_compress(d, T.INT2_MINMAX)), R.bits(R.sim_int2_minmax(bits(d))))
    with pytest.raises(ValueError):
        S.slowpath_compress(d, T.IDENTITY)


def test_profiler_scopes_cpu():
    import time
    from compactfusion_amd.prof import Profiler, prof_summary
    p = Profiler()
    Profiler._singleton = p
    with Profiler.scope("total", cpu=True):
        with Profiler.scope("inner", cpu=True):
            time.sleep(0.01)
        with Profiler.scope("gpu-only"):       # skipped silently without a GPU
            pass

    @Profiler.prof_func("fn", cpu=True)
    def f():
        return 7
    assert f() == 7 and f() == 7
    tot, avg = p.elapsed_time("inner")
    assert tot >= 9.0 and avg >= 9.0
    assert p.elapsed_time("fn")[0] >= 0 and "gpu-only" not in p.events
    p.disable()
    with Profiler.scope("off", cpu=True):
        pass
    assert "off" not in p.events
    p.enable()
    lines = prof_summary(p, rank=0)
    assert any("[inner]" in l for l in lines)
    with pytest.raises(ValueError):
        p.elapsed_time("nope")
    p.reset()
    assert not p.events


def test_alias_injection_recipe_from_integration_md(cpu_kernels):
    """registering this package under the reference's module names lets unchanged call sites bind."""
    import importlib
    import sys
    saved = {k: v for k, v in sys.modules.items() if k == "xfuser" or k.startswith("xfuser.")}
    try:
        import types
        import compactfusion_amd.compact as _c
        sys.modules["xfuser"] = types.ModuleType("xfuser")
        sys.modules["xfuser.compact"] = _c
        for name in ("main", "utils", "ring", "fastpath", "slowpath", "compress_quantize", "compress_topk", "stats"):
            sys.modules[f"xfuser.compact.{name}"] = importlib.import_module(f"compactfusion_amd.compact.{name}")
        sys.modules["xfuser.compact.patchpara.df_utils"] = importlib.import_module("compactfusion_amd.compact.patchpara.df_utils")
        sys.modules["xfuser.prof"] = importlib.import_module("compactfusion_amd.prof")
        from xfuser.compact.main import compact_compress, compact_config, compact_init  # noqa: F401  (reference import lines)
        from xfuser.compact.ring import compact_fwd  # noqa: F401
        from xfuser.compact.utils import COMPACT_COMPRESS_TYPE, CompactConfig
        from xfuser.compact.patchpara.df_utils import PatchConfig  # noqa: F401
        from xfuser.prof import Profiler  # noqa: F401
        compact_init(CompactConfig(enabled=True, residual=1, ef=True, fastpath=True, comp_rank=-1,
                                   compress_func=lambda l, s: COMPACT_COMPRESS_TYPE.BINARY))
        assert compact_config().enabled and compact_config().get_compress_type() == "BINARY"
        import inspect
        sig = list(inspect.signature(compact_fwd).parameters)
        assert sig == ["q", "k", "v", "dropout_p", "softmax_scale", "causal", "window_size", "alibi_slopes", "return_attn_probs",
                       "deterministic", "attn_layer", "group", "joint_tensor_key", "joint_tensor_value", "joint_strategy",
                       "mod_idx", "current_iter"]                      # ring.py:36-54
    finally:
        for k in [k for k in sys.modules if k == "xfuser" or k.startswith("xfuser.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_bench_grouped_all_gather_layout():
    """bench.py's grouped exchange: with all-gather semantics (recv = concatenation of every rank's send buffer) the
    offset table must point at rank r's packet of layer l for every (l, r, K|V) - for any group size, ragged tail included."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    L, slot = 11, 8
    for live in (2, 3, 8):
        for G in (1, 2, 4, 5, 11, 16):
            # send[r][l][kv][slot] filled with a code identifying (r, l, kv)
            send = np.zeros((live, L, 2, slot), dtype=np.int64)
            for r in range(live):
                for l in range(L):
                    for kv in range(2):
                        send[r, l, kv, :] = (r * 1000 + l * 10 + kv)
            recv = np.zeros(L * live * 2 * slot, dtype=np.int64)
            for a in range(0, L, G):
                b = min(L, a + G)
                per_rank = (b - a) * 2 * slot
                region = a * live * 2 * slot
                for r in range(live):                                  # what ncclAllGather does for this group
                    recv[region + r * per_rank:region + (r + 1) * per_rank] = send[r, a:b].reshape(-1)
            for l in range(L):
                for r in range(live):
                    for kv in range(2):
                        o = bench.group_recv_offset(l, r, kv, G, L, live, slot)
                        assert (recv[o:o + slot] == r * 1000 + l * 10 + kv).all(), (live, G, l, r, kv)


def test_quantized_cache_int8_storage(cpu_kernels, monkeypatch):
    """CompactCache(quantize=True) (deprecated in the reference, utils.py:128-156): the base is stored as the int8 packet of
    quantize_int8 and handed out dequantised; a 1-bit residual exchange on top of it keeps sender and receiver states
    identical because both sides quantise the same reconstruction."""
    from compactfusion_amd.compact import utils as U, main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    with pytest.raises(AssertionError):
        U.CompactCache(quantize=True)
    monkeypatch.setattr(U, "ALLOW_DEPRECATED", True)
    c = U.CompactCache(quantize=True)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 256, generator=g).half()
    assert c.get_base("1-0-k") is None
    c.put("1-0-k", x, None)
    want = R.decompress("int8", R.compress("int8", bits(x).reshape(64, 256), None)[0], 64, 256)
    assert np.array_equal(bits(c.get_base("1-0-k")), R.bits(want))
    assert c.base["1-0-k"].numel() * 2 == 64 * 256 + 4 * 256            # int8 codes + fp16 scale + int16 zero point
    # the state machine on a quantised cache: two "ranks" in one process (distinct keys), same packets
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T.BINARY, comp_rank=-1,
                                  residual=1, ef=True, fastpath=True, quantized_cache=True))
    assert cm.compact_cache().quantize
    xs = [torch.randn(64, 256, generator=g).half()]
    for _ in range(3):
        xs.append((xs[-1].float() + 0.1 * torch.randn(64, 256, generator=g)).half())
    for step, xt in enumerate(xs):
        cm.compact_set_step(step)
        typ = T.WARMUP if step == 0 else T.BINARY
        pkt = cm.compact_compress("5-0-k", xt, typ, update_cache=True)
        rec = cm.compact_decompress("5-1-k", pkt.clone(), typ, xt.shape, update_cache=True)
        s_state, r_state = cm.compact_cache().get_base("5-0-k").clone(), cm.compact_cache().get_base("5-1-k").clone()
        assert np.array_equal(bits(s_state), bits(r_state)), step
        if step:
            assert pkt.numel() == 64 * 256 // 16 + 64 + 256
            assert float((rec.float() - xt.float()).norm() / xt.float().norm()) < 0.2


def test_lowrank_awl_simulate_codec_matches_the_reference(monkeypatch):
    """LOW_RANK_AWL (deprecated upstream; simulate mode only, slowpath.py:217-237 + ring.py:77-118): the importance-weighted rank-r
    approximation against golden G14 captured from the reference (tests/golden/make_golden_awl.py) - same seeded start matrix, per-token
    scale on a K key, per-channel scale on a V key, no scale; and the deprecation guard."""
    import os
    import numpy as np
    import torch
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import slowpath as sp, utils
    from compactfusion_amd.compact.utils import COMPACT_COMPRESS_TYPE as T
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14_lowrank_awl.npz"))
    x = torch.from_numpy(g["x"].copy()).view(torch.float16)
    tok, chan = torch.from_numpy(g["tok"].copy()), torch.from_numpy(g["chan"].copy())
    monkeypatch.setattr(utils, "ALLOW_DEPRECATED", False)
    monkeypatch.setattr(cm, "_current_cache_key", "0-0-k")
    with pytest.raises(AssertionError, match="deprecated"):
        sp.sim_compress(x, T.LOW_RANK_AWL, rank=8)
    monkeypatch.setattr(utils, "ALLOW_DEPRECATED", True)
    for name, key, sk, sv in (("k_token_scale", "0-0-k", tok, None), ("v_channel_scale", "0-0-v", None, chan), ("k_no_scale", "0-0-k", None, None)):
        monkeypatch.setattr(cm, "_current_cache_key", key)
        sp.set_current_lowrank_scale(sk, sv)
        torch.manual_seed(4321)
        y = sp.sim_compress(x, T.LOW_RANK_AWL, rank=8).float().numpy()
        want = g[name]
        assert y.shape == want.shape
        assert np.linalg.norm(y - want) / np.linalg.norm(want) < 1e-3, name
    sp.set_current_lowrank_scale(None, None)


def test_compact_update_awl_scale_sets_the_token_importance(monkeypatch):
    """ring.py:77-103: only with USE_AWL=1; scale_k[token] = mean |v| / |v[token]|, scale_v stays unset."""
    import torch
    from compactfusion_amd.compact import ring, slowpath as sp
    q = torch.randn(1, 12, 2, 8)
    v = torch.randn(1, 12, 2, 8)
    sp.set_current_lowrank_scale(None, None)
    monkeypatch.delenv("USE_AWL", raising=False)
    ring.compact_update_awl_scale(q, q, v)
    assert sp._current_lowrank_scale_k is None
    monkeypatch.setenv("USE_AWL", "1")
    ring.compact_update_awl_scale(q, q, v)
    n = v.reshape(12, 16).norm(dim=-1)
    assert torch.allclose(sp._current_lowrank_scale_k, n.mean() / n) and sp._current_lowrank_scale_v is None
    sp.set_current_lowrank_scale(None, None)


def test_install_xfuser_alias_resolves_the_reference_import_paths():
    """compat.install_xfuser_alias: the reference's call sites import xfuser.compact.* / xfuser.prof / xfuser.collector.collector by name
    (attn_layer.py:59-64, pipeline_flux.py:447-450, examples/flux_example.py:92-127) - after the alias those names are this package's modules.
    A process of its own: sys.modules is global."""
    import subprocess
    import sys
    import os
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from compactfusion_amd.compat import install_xfuser_alias\n"
        "names = install_xfuser_alias()\n"
        "from xfuser.compact.main import compact_init, compact_reset, compact_hello, compact_set_step, compact_compress, compact_decompress, compact_all_gather\n"
        "from xfuser.compact.utils import CompactConfig, COMPACT_COMPRESS_TYPE\n"
        "from xfuser.compact.ring import compact_fwd, compact_update_awl_scale\n"
        "from xfuser.compact.patchpara.df_utils import PatchConfig\n"
        "from xfuser.compact.slowpath import slowpath_compress, sim_compress, set_current_lowrank_scale\n"
        "from xfuser.prof import Profiler, prof_summary\n"
        "from xfuser.collector.collector import Collector, init, collect\n"
        "import compactfusion_amd.compact.main as m\n"
        "assert compact_init is m.compact_init and len(names) >= 15\n"
        "import compactfusion_amd.compact as c, importlib\n"
        "for sub in ('presets', 'plot', 'lowrank', 'xlayer', 'patchpara.state', 'patchpara.fwd', 'attention'):\n"
        "    assert sys.modules['xfuser.compact.' + sub] is importlib.import_module('compactfusion_amd.compact.' + sub), sub\n"
        "from xfuser.compact.presets import get_config\n"
        "from xfuser.compact.patchpara.state import PatchConfig as P2\n"
        "assert P2 is PatchConfig, 'PatchConfig exists twice'\n"
        "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]


def test_configure_is_the_one_place_for_host_switches(monkeypatch):
    """compactfusion_amd.configure: default < environment variable (the round 1-4 switches) < explicit call; unknown names and values are
    refused; hw_queues exports HIP's own variable and only before HIP is up."""
    import compactfusion_amd
    from compactfusion_amd import config
    config.reset()
    for v in ("CFX_EXCHANGE", "CFX_LANE", "CFX_RING_SCHEDULE", "CFX_LANE_EXCHANGE_CUS"):
        monkeypatch.delenv(v, raising=False)
    assert config.get("exchange") == "auto" and config.get("lane") == "auto" and config.get("lane_exchange_cus") == "32"
    monkeypatch.setenv("CFX_EXCHANGE", "rccl")
    assert config.get("exchange") == "rccl"
    eff = compactfusion_amd.configure(exchange="torch", lane="sticky", lane_exchange_cus=48)
    assert eff["exchange"] == "torch" and eff["lane"] == "sticky" and config.get("lane_exchange_cus") == "48"
    config.reset()
    assert config.get("exchange") == "rccl"                      # back to the environment layer
    with pytest.raises(ValueError):
        compactfusion_amd.configure(lane="sometimes")
    with pytest.raises(TypeError):
        compactfusion_amd.configure(no_such_switch=1)
    monkeypatch.setenv("CFX_LANE", "bogus")
    with pytest.raises(ValueError):
        config.get("lane")
    monkeypatch.delenv("CFX_LANE")
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    monkeypatch.setattr(config, "_hip_is_up", lambda: False)
    compactfusion_amd.configure(hw_queues=4)
    import os
    assert os.environ["GPU_MAX_HW_QUEUES"] == "4"
    monkeypatch.setattr(config, "_hip_is_up", lambda: True)
    with pytest.raises(RuntimeError, match="before the first CUDA call"):
        compactfusion_amd.configure(hw_queues=8)
    config.reset()


def test_named_configurations_equal_the_reference_presets():
    """compact/presets.py::get_config against G15 (tests/golden/make_golden_presets.py: the reference's examples/configs.py:6-200 run for
    every model x method): the same CompactConfig fields, the same PatchConfig, the same compress_func answers.  Where the reference's own
    dispatcher raises (`patch`: it passes an argument `_patch_config()` does not take) ours returns what that function builds."""
    import json
    import os
    from compactfusion_amd.compact import presets
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g15_presets.json")))
    assert len(gold) == 33
    fields = ("enabled", "override_with_patch_gather_fwd", "comp_rank", "compress_residual", "error_feedback", "simulate_compress",
              "log_compress_stats", "fastpath", "quantized_cache", "sparse_ratio", "delta_decay_factor", "check_cache_consistency")
    for key, g in gold.items():
        model, method = key.split("/")
        c = presets.get_config(model, method)
        if "raises" in g:
            assert method == "patch" and g["raises"] == "TypeError"
            assert c.enabled and c.override_with_patch_gather_fwd and c.compress_func is None and not c.error_feedback
            p = c.patch_gather_fwd_config
            assert (p.use_compact, p.async_comm, p.async_warmup) == (False, False, 0)          # configs.py:153-158
            continue
        for f in fields:
            if f in g:
                assert getattr(c, f) == g[f], (key, f, getattr(c, f), g[f])
        p = c.patch_gather_fwd_config
        assert (None if p is None else {"use_compact": p.use_compact, "async_comm": p.async_comm, "async_warmup": p.async_warmup}) == g["patch"], key
        sched = None if c.compress_func is None else [[c.compress_func(layer, step).name for step in range(4)] for layer in (0, 5)]
        assert sched == g["schedule"], key
    with pytest.raises(ValueError):
        presets.get_config("SDXL", "binary")                                                    # configs.py:35
    assert presets.get_config("Flux", "lowrank16").comp_rank == 16                              # defined upstream (:96-107), not dispatched there


def test_group_health_boundaries_with_and_without_step_numbers():
    """compact/xlayer.py GroupHealth: the ranks agree on time-outs at BOUNDARIES.  A boundary is a new step number - or, when the caller
    never sets one (or sits on one), the layer that opened the interval coming round again.  The decision is pure bookkeeping over the
    call sequence (every rank sees the same one)."""
    from compactfusion_amd.compact.xlayer import GroupHealth
    h = GroupHealth(None, 0)
    layers = [("ring", f"{l}-0-k", None) for l in range(3)]
    # explicit steps: one boundary per step, at the first layer that sees the new number
    seq = [(s, k) for s in (1, 2, 3) for k in layers]
    assert [h._new_interval(s, k) for s, k in seq] == [True, False, False] * 3
    # no step numbers at all: the first pass opens the interval, every later pass starts a boundary at the same layer
    h = GroupHealth(None, 0)
    got = [h._new_interval(None, k) for _ in range(3) for k in layers]
    assert got == [False, False, False, True, False, False, True, False, False]
    # a loop that sits on ONE step number (several forwards per step): still a boundary per pass after the first
    h = GroupHealth(None, 0)
    got = [h._new_interval(7, k) for _ in range(3) for k in layers]
    assert got == [True, False, False, True, False, False, True, False, False]
    # a single-layer model without steps: every call after the first is a boundary
    h = GroupHealth(None, 0)
    assert [h._new_interval(None, layers[0]) for _ in range(3)] == [False, True, True]


def test_start_pool_draws_once_per_chunk_and_never_hands_out_a_draw_twice():
    """compact/xlayer.py StartPool: the low-rank layer ops' start matrices, one normal_() per chunk of layers and step; every
    execution reads a draw nobody consumed before; another stream draws for itself; a pinned matrix survives until a redraw."""
    from compactfusion_amd.compact import xlayer
    sp = xlayer.StartPool(C=16, rp=8, rank=6, device=torch.device("cpu"))
    sp.CHUNK = 4
    slots = [sp.acquire() for _ in range(6)]                     # two chunks: 4 + 2 slots
    assert len(sp.chunks) == 2 and {c for c, _, _ in slots} == {0, 1}
    assert all(t.shape == (2, 16, 8) and not t.any() for _, _, t in slots)
    seen = []
    torch.manual_seed(0)
    for step in range(3):
        for c, i, t in slots:
            sp.take(c, i, "s0")
            assert not t[:, :, 6:].any(), "the padding columns stay zero"
            seen.append(t[:, :, :6].clone())
    assert sp.draws == 2 * 3, "one draw per chunk and step"
    flat = torch.stack(seen).reshape(len(seen), -1)
    assert len({tuple(r.tolist()) for r in flat}) == len(seen), "two executions read the same draw"
    # a slot used from another stream: draws for itself, and the chunk is no longer redrawn as a whole while that holds
    before = [t.clone() for _, _, t in slots[:4]]
    sp.take(0, 1, "s1")
    assert not torch.equal(slots[1][2], before[1]) and all(torch.equal(slots[j][2], before[j]) for j in (0, 2, 3))
    d0 = sp.draws
    sp.take(0, 0, "s0")                                            # consumed on s0 in step 2: needs a draw; slot 1 sits on s1 -> own slot only
    assert sp.draws == d0 + 1 and torch.equal(slots[2][2], before[2]) and not torch.equal(slots[0][2], before[0])
    # pinning: the slot holds the pinned matrix; a chunk redraw bumps the generation so the owner knows to pin again
    q = torch.arange(16 * 6, dtype=torch.float32).reshape(16, 6)
    gen = sp.pin(1, 0, q)
    assert torch.equal(slots[4][2][0, :, :6], q) and torch.equal(slots[4][2][1, :, :6], q) and not slots[4][2][:, :, 6:].any()
    sp.take(1, 1, "s0"); sp.take(1, 1, "s0")
    assert sp.chunks[1]["gen"] != gen
    # give-back and reuse
    sp.give_back(0, 1)
    c, i, t = sp.acquire()
    assert (c, i) == (0, 1) and not t[:, :, 6:].any()
    sp.take(0, 1, "s0"); sp.take(0, 0, "s0"); sp.take(0, 0, "s0")  # everything on s0 again: whole-chunk draws come back
    assert all(sp.chunks[0]["fresh"][j] for j in (1, 2, 3))
