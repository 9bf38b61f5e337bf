"""Residual / error-feedback state machine of the compressed activation exchange (the plugin API).

Mirror of the reference's `xfuser/compact/main.py`: the same module-level functions with the same signatures and
state (`compact_init/reset/hello/config/set_step/get_step/cache/allgather_cache`, `compact_compress`,
`compact_decompress`, `compact_all_gather`), so `xfuser.core.long_ctx_attention` and the pipelines can bind to it
unchanged.  What differs is underneath:

  * all codec arithmetic runs in hand-written gfx950 kernels behind the C-ABI (`compactfusion_amd.codecs` ->
    libcfx.so); one call = stats pass + finalize + apply pass that write the wire packet and the error-feedback
    state directly (no eager scale prologue, no torch.cat, no fresh allocations - reference main.py:130-166,
    fastpath.py:150-166,185-186);
  * state lives in the CompactCache arena and is updated in place; packets live in per-key persistent buffers;
  * `compact_all_gather` moves ONE contiguous buffer with `all_gather_into_tensor` and reconstructs all ranks'
    shards with ONE batched launch (reference main.py:406-419: list all_gather + W Python-level decompress calls).

Semantics kept from the reference (file:line = xfuser/compact/main.py):
  WARMUP stores the activation as the new base and sends it raw (:195-209, :351-366); residual 0 compresses the
  activation itself (:214-226, :371-372); residual 1 compresses act - base and sets base <- base + decode(packet)
  when error feedback is on, base <- act otherwise (:227-243, :373-377); residual 2 adds the decayed second-order
  predictor (:244-266, :378-384; `cfx_residual2_delta` / `cfx_residual2_update` around the codec, states in place); `simulate` ships the dequantised tensor (:117-119, :126-127); the fastpath accepts
  only BINARY / INT2 (:131, :277).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .. import codecs
from ..collector import collector as _collector_mod
from ..prof import Profiler
from .patchpara.df_cache import AllGatherCache
from .utils import ALLOW_DEPRECATED, COMPACT_COMPRESS_TYPE, CompactCache, CompactConfig

T = COMPACT_COMPRESS_TYPE

_config: Optional[CompactConfig] = None
_cache: Optional[CompactCache] = None
_step = None
_allgather_cache: Optional[AllGatherCache] = None
_current_cache_key = None
_packets: Dict[Tuple, torch.Tensor] = {}     # persistent packet / scratch buffers keyed by (key, role, numel)
_generation = 0                               # bumped by compact_init / compact_reset: invalidates cached pointer tables


# ------------------------------------------------------------------------------------------------------------
# global state (main.py:37-113)
# ------------------------------------------------------------------------------------------------------------
def compact_init(config: CompactConfig):
    global _config, _cache, _step, _allgather_cache, _current_cache_key, _generation
    _generation += 1
    _drop_kv_exchanges()
    _config = config
    _cache = CompactCache(quantize=config.quantized_cache)
    _step = None
    if config.override_with_patch_gather_fwd:
        _allgather_cache = AllGatherCache()
    _current_cache_key = None
    _packets.clear()
    import sys
    ring_ = sys.modules.get(__package__ + ".ring")
    if ring_ is not None:
        ring_._lane_ok.clear()                 # (process groups and the lane's streams may have changed between generations)


def compact_reset():
    """Fresh state for a new generation (main.py:93-106)."""
    global _cache, _step, _allgather_cache, _current_cache_key, _generation
    _generation += 1
    _drop_kv_exchanges()
    _cache = CompactCache(quantize=_config.quantized_cache)
    from .stats import stats_clear
    stats_clear()
    _step = None
    if _config.override_with_patch_gather_fwd:
        _allgather_cache = AllGatherCache()
    _current_cache_key = None
    _packets.clear()


def compact_hello():
    if dist.is_initialized() and dist.get_rank() != 0:
        return
    c = _config
    print("--- compactfusion_amd initialised ---")
    print("compact enabled" if c.enabled else "compact disabled")
    if c.enabled:
        if not c.override_with_patch_gather_fwd:
            print(f"fastpath={c.fastpath} simulate={c.simulate_compress} log_stats={c.log_compress_stats} "
                  f"check_consistency={c.check_cache_consistency} residual={c.compress_residual} ef={c.error_feedback}")
        else:
            pc = c.patch_gather_fwd_config
            print(f"patch-gather forward: async(DistriFusion)={pc.async_comm} compact={pc.use_compact}")
    print("--------------------------------------")


def compact_config():
    return _config


def compact_set_step(step):
    global _step
    _step = step


def compact_get_step():
    return _step


def compact_cache():
    return _cache


def allgather_cache():
    return _allgather_cache


def compact_get_current_cache_key():
    """FOR TESTING ONLY (main.py:108-113)."""
    return _current_cache_key


# ------------------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------------------
def _nc_shape(shape) -> Tuple[int, int]:
    """(N, C) view rule of main.py:180-185 / :333-342."""
    shape = tuple(shape)
    if len(shape) >= 4:
        n = 1
        for d in shape[:-2]:
            n *= d
        return n, shape[-2] * shape[-1]
    if len(shape) == 3:
        return shape[0] * shape[1], shape[2]
    assert len(shape) == 2
    return shape


def _native(compress_type: T) -> Tuple[int, int]:
    """COMPACT_COMPRESS_TYPE -> (libcfx codec id, param)."""
    if compress_type == T.BINARY:
        rank = _config.comp_rank
        if rank is not None and rank != -1:
            assert ALLOW_DEPRECATED, "Binary compression with rank != -1 is deprecated"      # main.py:188-189
            assert 1 <= rank <= 32, "1-bit with subspace-iteration scales: comp_rank must be 1..32 (the factor chain's limit)"
            from . import lowrank
            return lowrank.BINARY_RANK_ID, int(rank)
        return int(codecs.Codec.BINARY), 0
    if compress_type == T.INT2:
        return int(codecs.Codec.INT2), 0
    if compress_type == T.INT4:
        return int(codecs.Codec.INT4), 0
    if compress_type == T.INT8:
        return int(codecs.Codec.INT8), 0
    if compress_type == T.SPARSE:
        assert _config.sparse_ratio is not None, "sparse_ratio must be provided for SPARSE compression"
        return int(codecs.Codec.TOPK), int(_config.sparse_ratio)
    if compress_type in (T.LOW_RANK, T.LOW_RANK_Q):
        from . import lowrank
        return lowrank.native_id(compress_type), int(_config.comp_rank)
    raise ValueError(f"Invalid compress_type value: {compress_type}")


def compact_bind_packet(cache_key, buffer: torch.Tensor) -> None:
    """Extension: make `compact_compress(cache_key, ...)` write its packet straight into `buffer` (a 1-D fp16 view,
    16-byte aligned, of exactly the packet length) - e.g. a slot of an exchange buffer - instead of a private one."""
    assert buffer.dtype == torch.float16 and buffer.dim() == 1 and buffer.is_contiguous() and buffer.data_ptr() % 16 == 0
    _packets[(cache_key, "pkt", buffer.numel(), buffer.device, torch.float16)] = buffer


def _buf(key, role: str, numel: int, like: torch.Tensor, dtype=torch.float16) -> torch.Tensor:
    k = (key, role, numel, like.device, dtype)
    b = _packets.get(k)
    if b is None:
        b = torch.empty(numel, dtype=dtype, device=like.device)
        _packets[k] = b
    return b


def _codec_compress(cid: int, param: int, x: torch.Tensor, base: Optional[torch.Tensor], new_base: Optional[torch.Tensor],
                    packet: torch.Tensor, update: bool, ef: bool = True) -> None:
    N, C = x.shape
    if cid >= 100:
        from . import lowrank
        lowrank.compress(cid, param, x, base, new_base, packet, update, ef)
        return
    codecs.compress_batch(cid, [x], [base], [new_base if update else None], [packet], N, C, param, update_cache=update, ef=ef)


def _codec_decompress(cid: int, param: int, packet: torch.Tensor, base: Optional[torch.Tensor], out: torch.Tensor) -> None:
    N, C = out.shape
    if packet.data_ptr() % 16:
        packet = packet.clone()
    if cid >= 100:
        from . import lowrank
        lowrank.decompress(cid, param, packet, base, out)
        return
    codecs.decompress_batch(cid, [packet], [base], [out], N, C, param)


def _packet_halves(cid: int, param: int, N: int, C: int) -> int:
    if cid >= 100:
        from . import lowrank
        return lowrank.packet_halves(cid, param, N, C)
    return codecs.packet_halves(cid, N, C, param)


def _sim(x_nc: torch.Tensor, compress_type: T, key) -> torch.Tensor:
    """simulate=True: the 'compressed' tensor is decode(encode(x)) at full size (slowpath.py:185-239)."""
    if compress_type == T.IDENTITY:
        return x_nc
    if compress_type == T.INT2_MINMAX:
        from .compress_quantize import sim_int2_minmax
        return sim_int2_minmax(x_nc)
    cid, param = _native(compress_type)
    N, C = x_nc.shape
    pkt = _buf(key, "simpkt", _packet_halves(cid, param, N, C), x_nc)
    out = torch.empty_like(x_nc)
    _codec_compress(cid, param, x_nc, None, None, pkt, update=False)
    _codec_decompress(cid, param, pkt, None, out)
    return out


def _decay(delta_base: torch.Tensor) -> torch.Tensor:
    return delta_base * _config.delta_decay_factor


def _log(key, base, dbase, x, recon, compressed):
    if _config.log_compress_stats:
        from .stats import stats_log
        stats_log().log(key, base, dbase, x, recon, compressed, _config.compress_residual)


# ------------------------------------------------------------------------------------------------------------
# compress (main.py:169-270)
# ------------------------------------------------------------------------------------------------------------
@Profiler.prof_func("compact.compact_compress")
def compact_compress(cache_key, x: torch.Tensor, compress_type: COMPACT_COMPRESS_TYPE, update_cache: bool = False):
    global _current_cache_key
    _current_cache_key = cache_key
    assert x.is_contiguous()
    assert _config.enabled
    original_shape = x.shape
    x = x.view(_nc_shape(x.shape))
    N, C = x.shape
    cfg, cache = _config, _cache

    if compress_type == T.WARMUP:
        if update_cache:
            if cfg.fastpath:
                assert cfg.compress_residual == 1
                cache.put(cache_key, x, None)
            elif cfg.compress_residual == 1:
                cache.put(cache_key, x, None)
            elif cfg.compress_residual == 2:
                old = cache.get_base(cache_key)
                cache.put(cache_key, x, None if old is None else x - old)
        return x.view(original_shape)

    if cfg.fastpath:
        assert compress_type in (T.BINARY, T.INT2)
        assert cfg.compress_residual == 1

    # ---- simulate: ship the dequantised tensor -----------------------------------------------------------------
    if cfg.simulate_compress:
        if cfg.compress_residual == 0:
            return _sim(x, compress_type, cache_key)
        base = cache.get_base(cache_key)
        if cfg.compress_residual == 1:
            recv = _sim(x - base, compress_type, cache_key)
            if update_cache:
                cache.put(cache_key, (base + recv) if cfg.error_feedback else x, None)
            _log(cache_key, base, None, x, base + recv, recv)
            return recv
        dbase = cache.get_delta_base(cache_key)
        recv = _sim(x - base - dbase, compress_type, cache_key)
        if update_cache:
            cache.put(cache_key, base + dbase + recv, _decay(dbase + recv))
        return recv

    # ---- native wire codecs -------------------------------------------------------------------------------------
    cid, param = _native(compress_type)
    pkt = _buf(cache_key, "pkt", _packet_halves(cid, param, N, C), x)
    if cfg.compress_residual == 0:
        _codec_compress(cid, param, x, None, None, pkt, update=False)
        if cfg.log_compress_stats:
            rec = torch.empty_like(x)
            _codec_decompress(cid, param, pkt, None, rec)
            _log(cache_key, None, None, x, rec, pkt)
        return pkt
    base = cache.get_base(cache_key)
    assert base is not None, f"no cached base for key {cache_key}: a WARMUP step must precede residual compression"
    if cfg.compress_residual == 1:
        log_base = base.clone() if cfg.log_compress_stats else None
        # one fused call: packet + in-place error-feedback update of the arena buffer
        _codec_compress(cid, param, x, base, base, pkt, update=update_cache, ef=cfg.error_feedback)
        if update_cache:
            cache.put(cache_key, base, None)
        if cfg.log_compress_stats:
            rec = base if (update_cache and cfg.error_feedback) else None
            if rec is None:
                rec = torch.empty_like(x)
                _codec_decompress(cid, param, pkt, log_base, rec)
            _log(cache_key, log_base, None, x, rec, pkt)
        return pkt
    # residual 2: second-order predictor - two native elementwise passes around the native codec, states updated in place
    dbase = cache.get_delta_base(cache_key)
    assert dbase is not None, f"no second-order state for key {cache_key}: residual 2 needs two WARMUP steps"
    dd = _buf(cache_key, "dd", N * C, x).view(N, C)
    codecs.residual2_delta(x, base, dbase, dd)
    _codec_compress(cid, param, dd, None, None, pkt, update=False)
    if update_cache or cfg.log_compress_stats:
        recv = _buf(cache_key, "recv", N * C, x).view(N, C)
        _codec_decompress(cid, param, pkt, None, recv)
        if update_cache and not cfg.log_compress_stats:
            codecs.residual2_update(base, dbase, recv, base, dbase, cfg.delta_decay_factor)     # in place on the arena
            cache.put(cache_key, base, dbase)
        else:
            log_base, log_dbase = base.clone(), dbase.clone()
            new_base, new_dbase = torch.empty_like(base), torch.empty_like(dbase)
            codecs.residual2_update(base, dbase, recv, new_base, new_dbase, cfg.delta_decay_factor)
            if update_cache:
                cache.put(cache_key, new_base, new_dbase)
            _log(cache_key, log_base, log_dbase, x, new_base, pkt)
    return pkt


# ------------------------------------------------------------------------------------------------------------
# decompress (main.py:322-388)
# ------------------------------------------------------------------------------------------------------------
@Profiler.prof_func("compact.compact_decompress")
def compact_decompress(cache_key, compressed: torch.Tensor, compress_type: COMPACT_COMPRESS_TYPE, shape: tuple,
                       update_cache: bool = False):
    global _current_cache_key
    _current_cache_key = cache_key
    assert _config.enabled
    original_shape = tuple(shape)
    N, C = _nc_shape(shape)
    cfg, cache = _config, _cache

    if compress_type == T.WARMUP:
        val = compressed.view(N, C)
        if update_cache:
            if cfg.fastpath:
                assert cfg.compress_residual == 1
                cache.put(cache_key, val, None)
            elif cfg.compress_residual == 1:
                cache.put(cache_key, val, None)
            elif cfg.compress_residual == 2:
                old = cache.get_base(cache_key)
                cache.put(cache_key, val, None if old is None else val - old)
        return val.view(original_shape)

    if cfg.fastpath:
        assert compress_type in (T.BINARY, T.INT2)
        assert cfg.compress_residual == 1

    if cfg.simulate_compress:
        recv = compressed.view(N, C)
        if cfg.compress_residual == 0:
            return recv.view(original_shape)
        base = cache.get_base(cache_key)
        if cfg.compress_residual == 1:
            rec = base + recv
            if update_cache:
                cache.put(cache_key, rec, None)
            return rec.view(original_shape)
        dbase = cache.get_delta_base(cache_key)
        rec = base + dbase + recv
        if update_cache:
            cache.put(cache_key, rec, _decay(dbase + recv))
        return rec.view(original_shape)

    cid, param = _native(compress_type)
    expected = _packet_halves(cid, param, N, C)
    assert compressed.numel() == expected, \
        f"Mismatch in compressed tensor size: expected {expected}, got {compressed.numel()}, Shape (N,C)=({N},{C})"
    if cfg.compress_residual == 0:
        out = torch.empty((N, C), dtype=torch.float16, device=compressed.device)
        _codec_decompress(cid, param, compressed, None, out)
        return out.view(original_shape)
    base = cache.get_base(cache_key)
    assert base is not None, f"no cached base for key {cache_key}: a WARMUP step must precede residual decompression"
    if cfg.compress_residual == 1:
        if update_cache:
            _codec_decompress(cid, param, compressed, base, base)       # in place: recon IS the new base
            cache.put(cache_key, base, None)
            return base.view(original_shape)
        out = _buf(cache_key, "recon", N * C, base).view(N, C)
        _codec_decompress(cid, param, compressed, base, out)
        return out.view(original_shape)
    dbase = cache.get_delta_base(cache_key)
    assert dbase is not None, f"no second-order state for key {cache_key}: residual 2 needs two WARMUP steps"
    recv = _buf(cache_key, "recv", N * C, base).view(N, C)
    _codec_decompress(cid, param, compressed, None, recv)
    if update_cache:
        codecs.residual2_update(base, dbase, recv, base, dbase, cfg.delta_decay_factor)         # in place on the arena
        cache.put(cache_key, base, dbase)
        return base.view(original_shape)
    rec = torch.empty((N, C), dtype=torch.float16, device=compressed.device)
    codecs.residual2_update(base, dbase, recv, rec, _buf(cache_key, "ndb", N * C, base).view(N, C), cfg.delta_decay_factor)
    return rec.view(original_shape)


# ------------------------------------------------------------------------------------------------------------
# all-gather of compressed shards (main.py:390-420)
# ------------------------------------------------------------------------------------------------------------
def compact_all_gather(tag, x: torch.Tensor, comp_type: COMPACT_COMPRESS_TYPE, group=None) -> List[torch.Tensor]:
    """Every rank contributes its shard; returns the list of W reconstructed shards (own shard included - it is
    replaced by its lossy reconstruction, as in the reference).  Keys: f"{tag}-{rank}"."""
    assert _config.enabled
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    cfg = _config
    to_send = compact_compress(f"{tag}-{rank}", x, comp_type, update_cache=False)
    flat = to_send.reshape(-1)
    native_batch = (comp_type != T.WARMUP and not cfg.simulate_compress and cfg.compress_residual == 1
                    and world <= codecs.CFX_MAX_BATCH and not _cache.quantize)
    # one contiguous receive buffer; each rank's slot starts 256-byte aligned
    slot = (flat.numel() + 127) // 128 * 128
    sendbuf = _buf(tag, "agsend", slot, flat, flat.dtype)
    sendbuf[:flat.numel()].copy_(flat)
    recvbuf = _buf(tag, "agrecv", slot * world, flat, flat.dtype)
    with Profiler.scope("compact.all_gather"):
        dist.all_gather_into_tensor(recvbuf, sendbuf, group=group)
    bufs = [recvbuf[i * slot:i * slot + flat.numel()] for i in range(world)]
    if not native_batch:
        return [compact_decompress(f"{tag}-{i}", bufs[i], comp_type, x.shape, update_cache=True) for i in range(world)]
    # batched native reconstruction: one launch for all W shards, in place on the state arena
    global _current_cache_key
    N, C = _nc_shape(x.shape)
    cid, param = _native(comp_type)
    bases = []
    for i in range(world):
        b = _cache.get_base(f"{tag}-{i}")
        assert b is not None, f"no cached base for key {tag}-{i}"
        bases.append(b)
    if cid >= 100:
        for i in range(world):
            _codec_decompress(cid, param, bufs[i], bases[i], bases[i])
    else:
        with Profiler.scope("compact.decompress_batch"):
            codecs.decompress_batch(cid, bufs, bases, bases, N, C, param)
    for i in range(world):
        _cache.put(f"{tag}-{i}", bases[i], None)
        _current_cache_key = f"{tag}-{i}"
    return [b.view(x.shape) for b in bases]


# ------------------------------------------------------------------------------------------------------------
# fused K+V gather (what patch_gather_fwd calls) and its displaced variant
# ------------------------------------------------------------------------------------------------------------
_kv_exchanges: Dict[Tuple, "_KVExchange"] = {}


def _drop_kv_exchanges():
    for ex in _kv_exchanges.values():
        ex.drain()
    _kv_exchanges.clear()


class _KVExchange:
    """Per-layer state of the fused K+V packet gather: double-buffered exchange buffers, key strings and the prepared
    native batches (pointer tables bound once to the state arena).

    sync      - compress K,V (no state update) -> ONE all-gather of [K packet | V packet] -> ONE batched reconstruction
                of all W ranks' K and V (own shard included, as compact_all_gather does, main.py:406-419).
    displaced - extension (the reference forbids async + compact, df_utils.py:13-16): the gather of step t is left in
                flight and applied at the start of step t+1, so peers' K/V are one step stale (DistriFusion's
                staleness) while every rank still applies every packet exactly once and in order - the replicated
                states stay bit-identical across ranks; the own shard is used fresh."""

    def __init__(self, tag_k, tag_v, rank, world, slot, like, group):
        self.rank, self.world, self.slot, self.group = rank, world, slot, group
        self.send = [torch.empty(2 * slot, dtype=torch.float16, device=like.device) for _ in range(2)]
        self.recv = [torch.empty(world * 2 * slot, dtype=torch.float16, device=like.device) for _ in range(2)]
        self.kkeys = [f"{tag_k}-{i}" for i in range(world)]
        self.vkeys = [f"{tag_v}-{i}" for i in range(world)]
        self.sig = None
        self.comp, self.dec = [None, None], [[], []]
        self.kviews, self.vviews = [], []
        self.parity = 0
        self.pending = None
        self.cuda = like.is_cuda
        self.device = like.device
        self.tags = (tag_k, tag_v)
        self.xop = None              # the synchronous exchange as ONE native op per layer (xlayer.LayerOp)
        self._xop_args = None
        self.steady = None           # (codec type, config, shape, generation, arena version, device) the bound op is valid for

    def bind(self, sig, cid, param, N, C, n_half, shape, ef):
        def state(key):
            b = _cache.get_base(key)
            assert b is not None, f"no cached base for key {key}"
            return b
        kb = [state(k) for k in self.kkeys]
        vb = [state(k) for k in self.vkeys]
        own = [kb[self.rank], vb[self.rank]]
        bases = [t for pair in zip(kb, vb) for t in pair]
        step = codecs.CFX_MAX_BATCH
        for p in range(2):
            s_, r_, sl = self.send[p], self.recv[p], self.slot
            own_pkts = [s_[:n_half], s_[sl:sl + n_half]]
            pkts = [r_[i * sl:i * sl + n_half] for i in range(2 * self.world)]
            if cid >= 100:
                # low-rank family (BASELINE config 5): the batched native chain, K and V together; the start matrices are drawn
                # per call as the reference does (compress_lowrank.py:41)
                from . import lowrank
                quant = cid == lowrank.LOW_RANK_Q_ID

                def comp(xs, sh, own_pkts=own_pkts):
                    dev = xs[0].device
                    codecs.lr_compress_batch(quant, [x.view(N, C) for x in xs], own, [None, None], own_pkts,
                                             [lowrank._start(C, param, dev), lowrank._start(C, param, dev)], N, C, param,
                                             update_cache=False, ef=ef)
                self.comp[p] = comp
                self.dec[p] = [(lambda sh, a=pkts[i:i + step], b=bases[i:i + step]: codecs.lr_decompress_batch(quant, a, b, b, N, C, param))
                               for i in range(0, len(bases), step)]
                continue
            self.comp[p] = codecs.prepare_compress(cid, own, [None, None], own_pkts, N, C, param, update_cache=False, ef=ef)
            self.dec[p] = [codecs.prepare_decompress(cid, pkts[i:i + step], bases[i:i + step], bases[i:i + step], N, C, param)
                           for i in range(0, len(bases), step)]
        self.kviews = [b.view(shape) for b in kb]
        self.vviews = [b.view(shape) for b in vb]
        self.sig = sig
        # sync steps: compress ; exchange ; reconstruct all W shards (the own shard is replaced by its reconstruction too, main.py:406-419)
        # as ONE native call (built on first use)
        from . import xlayer
        self._drop_xop()
        if xlayer.usable(cid, self.world, self.cuda, ef=ef):
            peers = [(r, kb[r], vb[r]) for r in range(self.world) if r != self.rank]
            self._xop_args = (("gather",) + self.tags + (id(self.group) if self.group is not None else None,), cid, param, N, C,
                              self.rank, self.world, self.group, self.device, own, peers)

    def _drop_xop(self):
        if self.xop is not None:
            self.xop.close()
        self.xop, self._xop_args, self.steady = None, None, None

    def step_steady(self, k, v):
        """`step` of the synchronous exchange with nothing to re-check: one native call."""
        global _current_cache_key
        self.xop.run(k, v, torch.cuda.current_stream(self.device).cuda_stream)
        inst = _collector_mod.instance
        if inst is None or inst.enabled:
            for key in self.kkeys + self.vkeys:
                _cache.touch(key)
        _current_cache_key = self.vkeys[-1]
        return self.kviews[:], self.vviews[:]

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.cuda else None

    def _apply(self, p):
        global _current_cache_key
        sh = self._stream()
        with Profiler.scope("compact.decompress_batch"):
            for run in self.dec[p]:
                run(sh)
        from ..collector import collector
        if collector.instance is None or collector.instance.enabled:
            for key in self.kkeys + self.vkeys:
                _cache.touch(key)
        _current_cache_key = self.vkeys[-1]

    def flush(self):
        """Apply a displaced gather that is still in flight: all states advance to the step it was issued at."""
        if self.pending is not None:
            handle, p = self.pending
            self.pending = None
            handle.wait()
            self._apply(p)

    def drain(self):
        if self.pending is not None:
            self.pending[0].wait()
            self.pending = None
        self._drop_xop()

    def step(self, k, v, displaced: bool):
        self.flush()
        if not displaced and self._xop_args is not None:
            global _current_cache_key
            if self.xop is None:
                from . import xlayer
                self.xop = xlayer.LayerOp(*self._xop_args, own_update="ef")
            with Profiler.scope("compact.exchange_layer"):
                self.xop.run(k, v, self._stream())
            from ..collector import collector
            if collector.instance is None or collector.instance.enabled:
                for key in self.kkeys + self.vkeys:
                    _cache.touch(key)
            _current_cache_key = self.vkeys[-1]
            return list(self.kviews), list(self.vviews)
        p = self.parity
        self.parity ^= 1
        with Profiler.scope("compact.compress_batch"):
            self.comp[p]((k, v), self._stream())
        if displaced:
            with Profiler.scope("df.all_gather"):
                handle = dist.all_gather_into_tensor(self.recv[p], self.send[p], group=self.group, async_op=True)
            self.pending = (handle, p)
            ks, vs = list(self.kviews), list(self.vviews)
            ks[self.rank], vs[self.rank] = k, v               # own shard is always fresh
            return ks, vs
        with Profiler.scope("compact.all_gather"):
            dist.all_gather_into_tensor(self.recv[p], self.send[p], group=self.group)
        self._apply(p)
        return list(self.kviews), list(self.vviews)


def compact_flush_displaced() -> None:
    """Apply every displaced gather still in flight (end of a generation, or before inspecting the states)."""
    for ex in _kv_exchanges.values():
        ex.flush()


def compact_all_gather_kv(tag_k, tag_v, k: torch.Tensor, v: torch.Tensor, comp_type: COMPACT_COMPRESS_TYPE, group=None,
                          displaced: bool = False) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """`compact_all_gather(tag_k, k)` + `compact_all_gather(tag_v, v)` as ONE exchange (one collective, one batched
    reconstruction) whenever the codec runs natively with first-order residuals; the two separate calls otherwise.
    `displaced=True` selects the one-step-stale variant described in `_KVExchange`."""
    assert _config.enabled
    cfg = _config
    xkey = (tag_k, tag_v, id(group) if group is not None else None)
    ex = _kv_exchanges.get(xkey)
    # steady state of the synchronous exchange: the layer is bound to ONE native op and nothing it was bound against has changed - straight to it
    if ex is not None and not displaced:
        st = ex.steady
        # (identity of the config object AND the mutable fields the op was bound against; a displaced step in between cleared `steady`)
        if (st is not None and ex.pending is None and st[0] is comp_type and st[1] is cfg and k.shape == st[2] and v.shape == st[2]
                and _generation == st[3] and _cache.version == st[4] and k.device == st[5] and st[6] == _steady_flags(cfg, k)
                and not cfg.log_compress_stats and k.is_contiguous() and v.is_contiguous()):
            return ex.step_steady(k, v)
    fusable = (comp_type != T.WARMUP and not cfg.simulate_compress and cfg.compress_residual == 1
               and not cfg.log_compress_stats and k.shape == v.shape and k.is_contiguous() and v.is_contiguous()
               and not _cache.quantize)
    if fusable:
        cid, param = _native(comp_type)
    if not fusable:
        if displaced and comp_type != T.WARMUP:
            # never degrade silently to a synchronous gather: the displaced exchange exists for native first-order codecs only
            why = ("simulate_compress" if cfg.simulate_compress else "compress_residual != 1" if cfg.compress_residual != 1 else
                   "log_compress_stats" if cfg.log_compress_stats else "quantized_cache" if _cache.quantize else
                   "K and V of different shape or not contiguous")
            raise NotImplementedError(f"displaced (one-step-stale) compressed gather is not available with {why}; "
                                      "use PatchConfig(displaced_compact=False) or change that option")
        if ex is not None:
            ex.flush()
        return (compact_all_gather(tag_k, k, comp_type, group=group), compact_all_gather(tag_v, v, comp_type, group=group))
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    N, C = _nc_shape(k.shape)
    n_half = _packet_halves(cid, param, N, C)
    slot = (n_half + 127) // 128 * 128
    if ex is None or ex.slot != slot or ex.world != world or ex.device != k.device:
        if ex is not None:
            ex.flush()
        ex = _kv_exchanges[xkey] = _KVExchange(tag_k, tag_v, rank, world, slot, k, group)
    ex.flush()
    sig = (_generation, _cache.version, cid, param, N, C, tuple(k.shape), cfg.error_feedback)
    if ex.sig != sig:
        ex.bind(sig, cid, param, N, C, n_half, k.shape, cfg.error_feedback)
    out = ex.step(k, v, displaced)
    if not displaced and ex.xop is not None and ex.pending is None and not cfg.simulate_compress and cfg.compress_residual == 1:
        ex.steady = (comp_type, cfg, k.shape, _generation, _cache.version, k.device, _steady_flags(cfg, k))
    elif displaced or ex.pending is not None:
        ex.steady = None             # a displaced delta is in flight: the next synchronous call takes the general path (which flushes it)
    return out


def _steady_flags(cfg, k):
    return (cfg.simulate_compress, cfg.compress_residual, cfg.error_feedback, bool(_cache.quantize), k.dtype)
