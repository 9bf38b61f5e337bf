"""Worker bodies for the world_size-2 gloo tests (spawned processes; CPU tensors; kernels replaced by the oracle
stand-in of tests/_oracle_backend.py - test infrastructure only)."""
import os
import sys
import tempfile
import traceback

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (REPO, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


DEV = "cpu"      # "cuda": the same worker bodies on the real kernels (two processes share GPU 0; tests/test_gpu_schedules.py)


def _setup(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if DEV == "cpu":
        import _oracle_backend as OB
        OB.install_plain()
    else:
        torch.cuda.set_device(0)
    from compactfusion_amd.collector import collector
    collector.init(collector.Collector(tempfile.mkdtemp(), enabled=False))


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def TD(t):
    """Worker inputs live where the run does."""
    return t if (t is None or DEV == "cpu") else t.to("cuda")


def TH(t):
    if DEV != "cpu":
        torch.cuda.synchronize()
    return t.float().cpu().numpy()


def drift(seed, shape, T):
    g = torch.Generator().manual_seed(seed)
    cur = torch.randn(*shape, generator=g).half()
    out = []
    for _ in range(T):
        out.append(cur.contiguous())
        cur = (cur.float() + 0.1 * torch.randn(*shape, generator=g)).half()
    return out


def run(fn, rank, world, port, out_path, *args, device="cpu"):
    global DEV
    DEV = device
    try:
        _setup(rank, world, port)
        res = fn(rank, world, *args)
        dist.barrier()
        if out_path:
            np.savez(out_path + f".r{rank}.npz", **(res or {}))
    except Exception:
        traceback.print_exc()
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
def w_all_gather(rank, world, codec_name):
    """compact_all_gather over `world` ranks; recipe identical to golden group G10 when codec is binary / int2."""
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    N, C = 32, 256
    fast = codec_name in ("BINARY", "INT2")
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: None, residual=1, ef=True, fastpath=fast, comp_rank=-1,
                                  sparse_ratio=8))
    res = {}
    xs = drift(100 + rank, (N, C), 4)
    for t, x in enumerate(xs):
        typ = T.WARMUP if t == 0 else T[codec_name]
        outs = cm.compact_all_gather("3-k", TD(x).view(1, N, C), typ)
        assert len(outs) == world and all(o.shape == (1, N, C) for o in outs)
        for i, o in enumerate(outs):
            res[f"t{t}/out{i}"] = bits(o).reshape(N, C).copy()
        res[f"t{t}/x"] = bits(x)
    cm.compact_cache().check_consistency()
    res["passed_count"] = np.array([cm.compact_cache().passed_count])
    return res


def _full_attention(q, ks, vs, scale=None):
    from compactfusion_amd.compact.attention import block_attention
    return block_attention(q, torch.cat(ks, dim=1), torch.cat(vs, dim=1), 0.0, scale, causal=False)


def w_ring(rank, world, schedule, codec_name, joint):
    """Ring forward over 3 steps (WARMUP then codec); returns out/lse and the K/V every rank ended up attending to."""
    os.environ["CFX_RING_SCHEDULE"] = schedule
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    from compactfusion_amd.compact.ring import compact_fwd
    B, S, Hh, Dh = (1, 16, 4, 32) if DEV == "cpu" else (1, 64, 8, 64)
    fast = codec_name in ("BINARY", "INT2")
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T[codec_name],
                                  residual=1, ef=True, fastpath=fast, comp_rank=-1, check_consist=True, sparse_ratio=8))
    qs = drift(7 + rank, (B, S, Hh, Dh), 3)
    ks = drift(17 + rank, (B, S, Hh, Dh), 3)
    vs = drift(27 + rank, (B, S, Hh, Dh), 3)
    g = torch.Generator().manual_seed(99)
    jk = torch.randn(B, 8, Hh, Dh, generator=g).half() if joint != "none" else None
    jv = torch.randn(B, 8, Hh, Dh, generator=g).half() if joint != "none" else None
    res = {}
    for step in range(3):
        cm.compact_set_step(step)
        out, lse, _ = compact_fwd(TD(qs[step]), TD(ks[step]), TD(vs[step]), causal=False, group=None, joint_tensor_key=TD(jk),
                                  joint_tensor_value=TD(jv), joint_strategy=joint, mod_idx=5, current_iter=step)
        assert out.shape == (B, S, Hh, Dh) and out.dtype == torch.float16 and lse.shape == (B, Hh, S)
        res[f"s{step}/out"] = TH(out)
        res[f"s{step}/lse"] = TH(lse)
        # what this rank attended to: its own exact K/V + the cached reconstructions of the peers
        kk, vv = [], []
        order = [(rank - s) % world for s in range(world)]
        for r in order:
            if r == rank:
                kk.append(TD(ks[step])); vv.append(TD(vs[step]))
            else:
                kk.append(cm.compact_cache().get_base(f"5-{r}-k").view(B, S, Hh, Dh).clone())
                vv.append(cm.compact_cache().get_base(f"5-{r}-v").view(B, S, Hh, Dh).clone())
        if joint == "front":
            kk.insert(0, TD(jk)); vv.insert(0, TD(jv))
        elif joint == "rear":
            kk.append(TD(jk)); vv.append(TD(jv))
        ref_out, ref_lse = _full_attention(TD(qs[step]), kk, vv)
        res[f"s{step}/ref_out"] = TH(ref_out)
        res[f"s{step}/ref_lse"] = TH(ref_lse)
        # peers hold exactly the state the owner holds for its own shard (error feedback keeps them in lock step)
        res[f"s{step}/own_k_state"] = bits(cm.compact_cache().get_base(f"5-{rank}-k")).copy()
        for r in range(world):
            res[f"s{step}/state_k_{r}"] = bits(cm.compact_cache().get_base(f"5-{r}-k")).copy()
            res[f"s{step}/state_v_{r}"] = bits(cm.compact_cache().get_base(f"5-{r}-v")).copy()
        res[f"s{step}/k"] = bits(ks[step])
        res[f"s{step}/v"] = bits(vs[step])
    res["passed_count"] = np.array([cm.compact_cache().passed_count])
    import compactfusion_amd.compact.ring as ring_mod
    res["p2p"] = np.array([int(any(getattr(ex, "_p2p", None) is not None or (ex.xop is not None and ex.xop.transport == "p2p")
                                   for ex in ring_mod._xbuf.values()))])
    return res


def w_patch(rank, world, mode):
    """patch_gather_fwd in its three modes."""
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig, PatchConfig
    from compactfusion_amd.compact.ring import compact_fwd
    B, S, H, D = (1, 16, 4, 32) if DEV == "cpu" else (1, 64, 8, 64)
    pc = {"sync": PatchConfig(False, False, 0), "async": PatchConfig(False, True, 1), "compact": PatchConfig(True, False, 1)}[mode]
    cm.compact_init(CompactConfig(enabled=True, override_with_patch_gather_fwd=True, patch_gather_fwd_config=pc,
                                  compress_func=(lambda l, s: T.WARMUP if s == 0 else T.INT2) if mode == "compact" else None,
                                  residual=1 if mode == "compact" else 0, ef=mode == "compact", fastpath=mode == "compact", comp_rank=-1))
    qs = drift(7 + rank, (B, S, H, D), 4)
    ks = drift(17 + rank, (B, S, H, D), 4)
    vs = drift(27 + rank, (B, S, H, D), 4)
    allk = [drift(17 + r, (B, S, H, D), 4) for r in range(world)]
    allv = [drift(27 + r, (B, S, H, D), 4) for r in range(world)]
    res = {}
    for step in range(4):
        cm.compact_set_step(step)
        out, lse, _ = compact_fwd(TD(qs[step]), TD(ks[step]), TD(vs[step]), causal=False, group=None, mod_idx=2, current_iter=step)
        res[f"s{step}/out"] = TH(out)
        if mode == "sync" or (mode == "async" and step < 1) or (mode == "compact" and step == 0):
            kk = [TD(allk[r][step]) for r in range(world)]
            vv = [TD(allv[r][step]) for r in range(world)]
        elif mode == "async":
            # remote shards are one step stale, own shard is fresh (DistriFusion, fwd.py:146-159)
            kk = [TD(allk[r][step] if r == rank else allk[r][step - 1]) for r in range(world)]
            vv = [TD(allv[r][step] if r == rank else allv[r][step - 1]) for r in range(world)]
        else:
            kk = [cm.compact_cache().get_base(f"2-k-{r}").view(B, S, H, D).clone() for r in range(world)]
            vv = [cm.compact_cache().get_base(f"2-v-{r}").view(B, S, H, D).clone() for r in range(world)]
            for r in range(world):
                res[f"s{step}/state_k_{r}"] = bits(kk[r]).copy()
        ref_out, _ = _full_attention(TD(qs[step]), kk, vv)
        res[f"s{step}/ref_out"] = TH(ref_out)
        res[f"s{step}/k"] = bits(ks[step])
    return res


def w_patch_displaced(rank, world, codec_name="BINARY"):
    """Extension: displaced (one-step-stale) compressed patch gather next to the synchronous compressed gather."""
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig, PatchConfig, lowrank
    from compactfusion_amd.compact.ring import compact_fwd
    lr = codec_name.startswith("LOW_RANK")
    if lr:   # same start matrix in both runs (the reference draws a fresh one per call: the two runs could not be compared)
        lowrank.set_init_q(torch.randn(4 * 32 if DEV == "cpu" else 8 * 64, 8, generator=torch.Generator().manual_seed(77)))
    B, S, H, D, STEPS = (1, 16, 4, 32, 5) if DEV == "cpu" else (1, 64, 8, 64, 5)
    qs = drift(7 + rank, (B, S, H, D), STEPS)
    ks = drift(17 + rank, (B, S, H, D), STEPS)
    vs = drift(27 + rank, (B, S, H, D), STEPS)
    res = {}
    for mode, pc in (("sync", PatchConfig(True, False, 1)), ("disp", PatchConfig(True, True, 1, displaced_compact=True))):
        cm.compact_init(CompactConfig(enabled=True, override_with_patch_gather_fwd=True, patch_gather_fwd_config=pc,
                                      compress_func=lambda l, s: T.WARMUP if s == 0 else T[codec_name],
                                      residual=1, ef=True, fastpath=not lr, comp_rank=8 if lr else -1))
        for step in range(STEPS):
            cm.compact_set_step(step)
            out, lse, _ = compact_fwd(TD(qs[step]), TD(ks[step]), TD(vs[step]), causal=False, group=None, mod_idx=2, current_iter=step)
            res[f"{mode}/s{step}/out"] = TH(out)
            for r in range(world):
                res[f"{mode}/s{step}/state_k_{r}"] = bits(cm.compact_cache().get_base(f"2-k-{r}")).copy()
                res[f"{mode}/s{step}/state_v_{r}"] = bits(cm.compact_cache().get_base(f"2-v-{r}")).copy()
        cm.compact_flush_displaced()
        for r in range(world):
            res[f"{mode}/final/state_k_{r}"] = bits(cm.compact_cache().get_base(f"2-k-{r}")).copy()
    # what the displaced forward must have attended to: own shard fresh, peers as of the previous step
    sync_state = lambda t, kv, r: TD(torch.from_numpy(res[f"sync/s{t}/state_{kv}_{r}"].view(np.int16).copy()).view(torch.float16).view(B, S, H, D))  # noqa: E731
    for step in range(1, STEPS):
        kk = [TD(ks[step]) if r == rank else sync_state(step - 1, "k", r) for r in range(world)]
        vv = [TD(vs[step]) if r == rank else sync_state(step - 1, "v", r) for r in range(world)]
        ref_out, _ = _full_attention(TD(qs[step]), kk, vv)
        res[f"disp/s{step}/ref_out"] = TH(ref_out)
    for step in range(STEPS):
        res[f"s{step}/k"] = bits(ks[step])
        res[f"s{step}/v"] = bits(vs[step])
    lowrank.set_init_q(None)
    return res


def w_hook_layer(rank, world, ulysses, ring, compact_on):
    """xFuserLongContextAttention over a (ulysses x ring) sequence-parallel group; compares with full attention."""
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    from compactfusion_amd.core import init_sequence_parallel, xFuserLongContextAttention
    from compactfusion_amd.core import long_ctx_attention as LCA
    LCA.reset_layer_index()
    init_sequence_parallel(ulysses, ring)
    if compact_on:
        cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T.BINARY,
                                      residual=1, ef=True, fastpath=True, comp_rank=-1))
    else:
        cm.compact_init(CompactConfig(enabled=False))
    layers = [xFuserLongContextAttention(), xFuserLongContextAttention()]
    B, S, H, D = (1, 16, 4, 32) if DEV == "cpu" else (1, 64, 8, 64)      # per-rank shard
    res = {}
    allq = [drift(7 + r, (B, S, H, D), 2) for r in range(world)]
    allk = [drift(17 + r, (B, S, H, D), 2) for r in range(world)]
    allv = [drift(27 + r, (B, S, H, D), 2) for r in range(world)]
    for step in range(2):
        cm.compact_set_step(step)
        for li, layer in enumerate(layers):
            out = layer(None, TD(allq[rank][step]), TD(allk[rank][step]), TD(allv[rank][step]), causal=False)
            assert out.shape == (B, S, H, D)
            res[f"s{step}/l{li}/out"] = TH(out)
            # reference: full attention over the whole sequence (ranks in order), my query shard
            from compactfusion_amd.compact.attention import block_attention
            kk = torch.cat([TD(allk[r][step]) for r in range(world)], dim=1)
            vv = torch.cat([TD(allv[r][step]) for r in range(world)], dim=1)
            ref, _ = block_attention(TD(allq[rank][step]), kk, vv, 0.0, None, causal=False)
            res[f"s{step}/l{li}/ref"] = TH(ref)
            if compact_on and ring > 1 and ulysses == 1:
                # what the compressed ring step must equal: ONE attention over what this rank holds - its own exact K,V, then the
                # peers' cached reconstructions in ring order (the comparison w_ring makes); and the states themselves
                order = [(rank - s_) % world for s_ in range(world)]
                hk = [TD(allk[rank][step]) if r_ == rank else cm.compact_cache().get_base(f"{li}-{r_}-k").view(B, S, H, D).clone() for r_ in order]
                hv = [TD(allv[rank][step]) if r_ == rank else cm.compact_cache().get_base(f"{li}-{r_}-v").view(B, S, H, D).clone() for r_ in order]
                held, _ = block_attention(TD(allq[rank][step]), torch.cat(hk, dim=1), torch.cat(hv, dim=1), 0.0, None, causal=False)
                res[f"s{step}/l{li}/ref_held"] = TH(held)
                for r_ in range(world):
                    res[f"s{step}/l{li}/state_k_{r_}"] = bits(cm.compact_cache().get_base(f"{li}-{r_}-k")).copy()
                    res[f"s{step}/l{li}/state_v_{r_}"] = bits(cm.compact_cache().get_base(f"{li}-{r_}-v")).copy()
        assert layers[0].idx == 0 and layers[1].idx == 1
    if compact_on:
        res["keys"] = np.array(sorted(cm.compact_cache().base.keys()), dtype="U")
    return res


def w_stack(rank, world, codec_name, steady=False):
    """Golden group G13's 4-layer attention stack over two ranks through `compact_fwd` (tests/golden/make_golden_stack.py holds the
    seeded recipe): the stack's final output per step with the compressed exchange and with the exact K,V exchanged (every step
    WARMUP = raw fp16 through the same ring path).  steady: profiler scopes and collector off - what a production run has - so that the
    steady layers take the calls (with the defaults of both, as here otherwise, the general path does)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_stack", os.path.join(HERE, "golden", "make_golden_stack.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    from compactfusion_amd.compact.ring import compact_fwd
    W = [[TD(w) for w in lw] for lw in G.weights()]
    xs = [TD(x) for x in G.inputs(rank)]
    res = {}
    if steady:
        from compactfusion_amd.collector import collector
        from compactfusion_amd.prof import Profiler
        Profiler.instance().disable()
        collector.init(collector.Collector("/tmp/none", enabled=False))

    def qkv(h, Wl):
        return [(h.float() @ w).half().view(G.B, G.S, G.H, G.D).contiguous() for w in Wl]

    def nxt(h, o):
        y = h.float() + o.float().reshape(G.B, G.S, G.C)
        return (y / y.pow(2).mean(dim=-1, keepdim=True).sqrt()).half()

    finals = {}
    tname, crank, fast = G.CODECS[codec_name.lower()] if codec_name.lower() in G.CODECS else (codec_name, -1, True)
    if crank > 0:       # the low-rank presets: the start of every subspace iteration pinned to the golden run's seeded matrix
        from compactfusion_amd.compact import lowrank
        lowrank.set_init_q(G.start_matrix(crank))
    for mode in ("exact", codec_name):
        cm.compact_init(CompactConfig(enabled=True, compress_func=(lambda l, s: T.WARMUP) if mode == "exact" else
                                      (lambda l, s: T.WARMUP if s == 0 else T[tname]),
                                      residual=1, ef=True, fastpath=fast, comp_rank=crank))
        outs = []
        for t in range(G.STEPS):
            cm.compact_set_step(t)
            h = xs[t]
            for l in range(G.LAYERS):
                q, k, v = qkv(h, W[l])
                o, _, _ = compact_fwd(q, k, v, causal=False, group=None, mod_idx=l, current_iter=t)
                h = nxt(h, o)
            outs.append(h)
        finals[mode] = outs
    if crank > 0:
        lowrank.set_init_q(None)
    res["psnr"] = np.array([G.psnr(finals["exact"][t].cpu(), finals[codec_name][t].cpu()) for t in range(G.STEPS)])
    res["exact_final"] = bits(finals["exact"][-1])
    # did an in-launch wait give up on the way?  (two rank PROCESSES time-slicing one GPU can starve each other's polling kernels past the
    # gate timeout; the run recovers - states re-synchronised from their owners - but a sender whose launch gave up has skipped one
    # error-feedback update, so its chain is no longer the golden run's)
    from compactfusion_amd.compact import ring as ring_mod
    ops = [e.xop for e in ring_mod._xbuf.values() if getattr(e, "xop", None) is not None]
    res["timeouts"] = np.array([sum(1 for o in ops if o.fallback_reason is not None) + (1 if DEV != "cpu" and _gate_errors_seen() else 0)])
    # low-rank family, lane at its default: the steady layers left their peers' reconstructions to the exchange lane (xlayer.LayerOp.run(lane=True))
    res["lane_ops"] = np.array([sum(1 for o in ops if getattr(o, "_lane", None) is not None)])
    return res


def _gate_errors_seen() -> int:
    try:
        from compactfusion_amd import _lib, codecs
        return int(_lib.load().cfx_gate_errors(codecs.context(torch.cuda.current_device())) > 0)
    except Exception:  # noqa: BLE001
        return 0


def w_xlayer(rank, world, codec_name, mode, poison, gens, steps=4, ef=True, late=None, gate_timeout_ms=0, revalidate_every=0):
    """The product path of the gather schedules - ONE native op per layer (compact/xlayer.py) - over `gens` generations with
    compact_reset in between: `mode` = "ring" (compact_fwd, gather schedule) or "gather" (compact_all_gather_kv, what patch_gather_fwd
    calls).  poison >= 0: rank 1 corrupts a peer's reconstruction right before its validated p2p execution `poison` - every rank must
    fall back to the next transport together and the states must come out as if nothing had happened.  late = (step, seconds, layer):
    rank 1 arrives that late at that layer of that step - rank 0's launch waits for it INSIDE the kernel (layer >= 1: the first layer op of
    a step meets the group's health all-reduce, which absorbs a late rank on the host)."""
    import time
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig, PatchConfig, xlayer
    from compactfusion_amd.compact import ring as ring_mod
    from compactfusion_amd.compact.ring import compact_fwd
    L, STEPS = 3, steps
    B, S, Hh, Dh = (1, 16, 4, 32) if DEV == "cpu" else (1, 64, 8, 64)
    os.environ["CFX_LANE"] = "off"                 # the one-op layer exchange on the caller's stream is what these runs are about
    if gate_timeout_ms:
        from compactfusion_amd import _lib, codecs
        assert _lib.load().cfx_set_gate_timeout_ms(codecs.context(torch.cuda.current_device()), gate_timeout_ms) == 0
    if revalidate_every:
        # the periodic re-validation of arenas that did not come out uncached: ask for fine-grained memory and shorten the period
        from compactfusion_amd import _lib, codecs
        assert _lib.load().cfx_set_ipc_memory_kind(codecs.context(torch.cuda.current_device()), 1) == 0
        xlayer.REVALIDATE_EVERY = revalidate_every
    if poison >= 0:
        # test-side only: the checksum of the first peer tensor is taken over a corrupted copy of what the launch reconstructed
        orig, done = xlayer.LayerOp._checksums, []

        def poisoned(self, tensors):
            if rank == 1 and not done and self.region is not None and self.region.validated == poison and tensors[0] is self.peers[0][1]:
                tensors[0].view(torch.int16)[0, :8] += 1          # a stale line's worth of wrong bits in a peer's reconstruction
                done.append(1)
            return orig(self, tensors)
        xlayer.LayerOp._checksums = poisoned
    fast = codec_name in ("BINARY", "INT2")
    kw = dict(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T[codec_name], residual=1, ef=ef, fastpath=fast and ef,
              comp_rank=-1, sparse_ratio=8)
    if mode == "gather":
        kw.update(override_with_patch_gather_fwd=True, patch_gather_fwd_config=PatchConfig(True, False, 1))
    cm.compact_init(CompactConfig(**kw))
    res = {}
    free = []
    late_done = []
    for gen in range(gens):
        if gen:
            cm.compact_reset()
        qs = [drift(1000 * gen + 7 + 10 * l + rank, (B, S, Hh, Dh), STEPS) for l in range(L)]
        ks = [drift(1000 * gen + 17 + 10 * l + rank, (B, S, Hh, Dh), STEPS) for l in range(L)]
        vs = [drift(1000 * gen + 27 + 10 * l + rank, (B, S, Hh, Dh), STEPS) for l in range(L)]
        for step in range(STEPS):
            cm.compact_set_step(step)
            for l in range(L):
                if late is not None and rank == 1 and l == (late[2] if len(late) > 2 else 0):
                    due = step == late[0]
                    if late[0] == "before_revalidation" and not late_done:
                        # the step after which layer 0's next execution is a periodic re-validation (its region has executed this step already)
                        op0 = [e.xop for k_, e in ring_mod._xbuf.items() if k_[0] == 0 and e.xop is not None]
                        due = bool(op0) and op0[0].region is not None and op0[0].region.validated >= xlayer.VALIDATE_FIRST \
                            and op0[0].region.n_exec % xlayer.REVALIDATE_EVERY == 0 and op0[0].transport == "p2p"
                    if due:
                        late_done.append(step)
                        torch.cuda.synchronize()
                        time.sleep(late[1])
                out, lse, _ = compact_fwd(TD(qs[l][step]), TD(ks[l][step]), TD(vs[l][step]), causal=False, group=None, mod_idx=l, current_iter=step)
                assert out.shape == (B, S, Hh, Dh)
            if gens <= 2:
                for l in range(L):
                    for r in range(world):
                        kk, vk = (f"{l}-{r}-k", f"{l}-{r}-v") if mode == "ring" else (f"{l}-k-{r}", f"{l}-v-{r}")
                        res[f"g{gen}/s{step}/l{l}/k{r}"] = bits(cm.compact_cache().get_base(kk)).copy()
                        res[f"g{gen}/s{step}/l{l}/v{r}"] = bits(cm.compact_cache().get_base(vk)).copy()
        if DEV != "cpu":
            torch.cuda.synchronize()
            free.append(torch.cuda.mem_get_info()[0])
    ops = [e.xop for e in ring_mod._xbuf.values() if e.xop is not None] + [e.xop for e in cm._kv_exchanges.values() if e.xop is not None]
    res["n_ops"] = np.array([len(ops)])
    res["p2p"] = np.array([sum(1 for o in ops if o.transport == "p2p")])
    res["fell_back"] = np.array([sum(1 for o in ops if o.fallback_reason is not None)])
    res["validated"] = np.array([min([o.region.validated for o in ops if o.region is not None] or [-1])])
    res["free"] = np.array(free, dtype=np.int64)
    res["late_at_step"] = np.array(late_done or [-1])
    dist.barrier()
    xlayer.release()
    return res
