"""Ring-attention forward with residual-compressed K/V exchange - the hook `xFuserLongContextAttention` binds.

Mirror of the reference's `xfuser/compact/ring.py` (`compact_fwd` :36-70 with the same 17-argument signature and
(out, lse, None) return; `_compact_ring_fwd` :120-275).  Two schedules produce the same numbers:

  relay  - the reference's schedule restated: each rank compresses its K and V once, the *packets* hop W-1 times
           around the ring with batched isend/irecv (own `RingComm`, the reference uses yunchang's), every hop's
           packet is decoded against the sender's cached state (keys "{layer}-{rank}-{k|v}"), attention blocks are
           merged with the running log-sum-exp.
  gather - the MI355X-native schedule (default on GPUs): xGMI is a full point-to-point mesh and a packet is ~0.2 MB,
           so all ranks' K+V packets are exchanged with ONE `all_gather_into_tensor` on a side HIP stream while the
           local attention block runs, then ALL peers' K and V are reconstructed with ONE batched dequant+add launch,
           and the attention blocks are visited in the same ring order (rank-1, rank-2, ...) so the merge order - and
           therefore the result - is identical to the relay schedule.
Step 0 uses the exact local K/V in both (ring.py:207-208); WARMUP steps move raw fp16 (main.py:195-209).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import config as _settings
from ..collector import collector as _collector_mod
from ..prof import Profiler
from . import main as cm
from .attention import block_attention, update_out_and_lse
from .main import compact_cache, compact_compress, compact_config, compact_decompress
from .utils import COMPACT_COMPRESS_TYPE

T = COMPACT_COMPRESS_TYPE
_profiler = Profiler.instance()


class RingComm:
    """One hop of a ring: post send-to-next / receive-from-previous pairs, commit them as one batch, wait.
    (restates the interface `yunchang.ring.utils.RingComm` offers at the call sites ring.py:172,193-195,267)"""

    def __init__(self, process_group=None):
        self._pg = process_group
        self.rank = dist.get_rank(process_group)
        self.world_size = dist.get_world_size(process_group)
        nxt, prv = (self.rank + 1) % self.world_size, (self.rank - 1) % self.world_size
        if process_group is not None:
            nxt, prv = dist.get_global_rank(process_group, nxt), dist.get_global_rank(process_group, prv)
        self.send_rank, self.recv_rank = nxt, prv
        self._ops: List[dist.P2POp] = []
        self._reqs = None

    def send_recv(self, to_send: torch.Tensor, recv_tensor: Optional[torch.Tensor] = None) -> torch.Tensor:
        res = torch.empty_like(to_send) if recv_tensor is None else recv_tensor
        self._ops.append(dist.P2POp(dist.isend, to_send, self.send_rank, group=self._pg))
        self._ops.append(dist.P2POp(dist.irecv, res, self.recv_rank, group=self._pg))
        return res

    def commit(self):
        assert self._reqs is None, "commit called twice"
        self._reqs = dist.batch_isend_irecv(self._ops)

    def wait(self):
        assert self._reqs is not None, "wait called before commit"
        for r in self._reqs:
            r.wait()
        self._reqs, self._ops = None, []


def compact_fwd(q, k, v, dropout_p=0, softmax_scale=None, causal=True, window_size=(-1, -1), alibi_slopes=None,
                return_attn_probs=None, deterministic=False, attn_layer=None, group=None, joint_tensor_key=None,
                joint_tensor_value=None, joint_strategy="none", mod_idx=None, current_iter=None):
    """Dispatch: patch-gather forward when `override_with_patch_gather_fwd`, else the ring forward (ring.py:36-70)."""
    args = (q, k, v, dropout_p, softmax_scale, causal, window_size, alibi_slopes, return_attn_probs, deterministic,
            attn_layer, group, joint_tensor_key, joint_tensor_value, joint_strategy, mod_idx, current_iter)
    if compact_config().override_with_patch_gather_fwd:
        from .patchpara.fwd import patch_gather_fwd
        return patch_gather_fwd(*args)
    # The overlapped form is the DEFAULT: the reference posts the exchange and then attends locally (ring.py:191-208); here the layer's
    # chain runs on the exchange lane beside the attention blocks whenever the layer has peers to hear from.  A caller that is not on the
    # lane's compute stream is put there for the duration of the call (configure(lane="auto"): forked from and joined to its own stream
    # by flag kernels) or for good (lane="sticky"); lane="off" keeps the exchange as one op on the caller's stream.
    token = dev = key = None
    if q.is_cuda and _auto_lane(q, group) and _lane_has_chain(mod_idx, current_iter, group):
        from .. import lanes
        dev = q.device.index if q.device.index is not None else torch.cuda.current_device()
        key = (mod_idx, id(group) if group is not None else None)
        st = _steady.get(key)
        if st is not None and st.ex.lane and st.ex.plan is not None and not lanes.on_compute_stream(dev):
            # steady layer: ONE flag word does both jobs - "K,V exist", published by the caller's stream (what cfx_plan_lane_begin does on
            # the compute stream when the caller is there already), is also what the compute lane waits for before it goes on.  The
            # compute lane runs neither a wait-for-fork nor a publish kernel of its own in front of the local attention block.
            token = lanes.fork_to_compute(dev, begin=st.ex.lane_begin, flag=lambda e: st.ex.flag_ptr(0))
            _prebegun[key] = token[2]
        else:
            token = lanes.fork_to_compute(dev)
        if token is not None and _settings.get("lane") == "sticky":
            token = None                       # the compute stream stays the thread's current stream
    try:
        return _compact_ring_fwd(*args)
    finally:
        if key is not None:
            _prebegun.pop(key, None)           # (not consumed: the call took the general path, which publishes its own epoch)
        if token is not None:
            lanes.join_from_compute(dev, token)


def _lane_has_chain(mod_idx, current_iter, group=None) -> bool:
    """The exchange lane runs the layer's chain of the STREAMING codecs (compress ; all-gather ; per-peer reconstruction, flag-ordered
    against the attention blocks).  The low-rank family's factor chain is one persistent launch that wants the whole chip: it has no
    chain of that kind.  A LOW_RANK / LOW_RANK_Q layer goes to the lane once it is a steady layer whose one-call layer op can leave the
    peers' reconstructions to the exchange lane (xlayer.LayerOp.lane_capable: peer-to-peer transport, configure(lowrank_lane="on")) -
    the factor chain then runs on the compute lane, exposed, the reconstructions beside the attention blocks; otherwise it keeps the
    caller's stream and takes the one-call layer op there (round 6: on the lane it fell through to one Python call per tensor, 25 ms per
    FLUX step)."""
    cfg = compact_config()
    try:
        ctype = cfg.compress_func(mod_idx, current_iter if current_iter is not None else cm.compact_get_step())
    except Exception:  # noqa: BLE001  (a compress_func that needs arguments this call does not have: the forward itself will say so)
        return True
    if ctype not in (COMPACT_COMPRESS_TYPE.LOW_RANK, COMPACT_COMPRESS_TYPE.LOW_RANK_Q):
        return True
    # (... and only while the steady layer is what will run: with the profiler's scopes or a live collector the general path takes the
    # call, which runs the layer op in stream order - it would pay the hand-over for nothing)
    st = _steady.get((mod_idx, id(group) if group is not None else None))
    return (st is not None and st.ctype is ctype and st.ex.xop is not None and st.ex.xop.lane_capable()
            and not _profiler.enabled and not _collector_live())


_lane_ok = {}        # (device, id(group)) -> world size >= 2 and the lane's streams usable: asked once, not on every layer call


def _auto_lane(q, group) -> bool:
    if _settings.get("lane") == "off" or _settings.get("ring_exchange_stream") not in ("auto", "lane") or _schedule(q) != "gather":
        return False
    if torch.cuda.is_current_stream_capturing():
        return False                           # the lane's flag kernels spin on words another stream writes: not capturable - the one-op path is
    dev = q.device.index if q.device.index is not None else torch.cuda.current_device()
    key = (dev, id(group) if group is not None else None)
    ok = _lane_ok.get(key)
    if ok is None:
        try:
            ok = dist.get_world_size(group) >= 2
        except Exception:  # noqa: BLE001  (no process group: a single rank has nobody to overlap with)
            ok = False
        if ok:
            from .. import lanes
            ok = bool(lanes.usable(dev))
        _lane_ok[key] = ok
    return ok


def compact_update_awl_scale(q, k, v) -> None:
    """ring.py:77-118 (deprecated upstream, only active with USE_AWL=1): key-token importance for LOW_RANK_AWL - tokens whose V row is
    small get a larger weight (mean |v| over |v| per token) - handed to the simulate-mode codec (slowpath.set_current_lowrank_scale)."""
    if os.getenv("USE_AWL", "0") != "1":
        return
    from .slowpath import set_current_lowrank_scale
    with torch.no_grad():
        bs, seq_len, head_cnt, head_size = q.shape
        v_norm = torch.norm(v.reshape(bs * seq_len, head_cnt * head_size), dim=-1).flatten()
        set_current_lowrank_scale((v_norm.mean() / v_norm).flatten(), None)


def _joint_mode(joint_tensor_key, joint_tensor_value, joint_strategy) -> Optional[str]:
    if (joint_tensor_key is None) != (joint_tensor_value is None):
        raise ValueError("joint_tensor_key and joint_tensor_value should be None or not None simultaneously.")
    if joint_tensor_key is None:
        return None
    if joint_strategy not in ("front", "rear"):
        raise ValueError(f"joint_strategy: {joint_strategy} not supprted. supported joint strategy: ['front', 'rear']")
    return joint_strategy


def _with_joint(k, v, jk, jv, mode, step, world):
    if mode == "front" and step == 0:
        return torch.cat([jk, k], dim=1), torch.cat([jv, v], dim=1)
    if mode == "rear" and step + 1 == world:
        return torch.cat([k, jk], dim=1), torch.cat([v, jv], dim=1)
    return k, v


def _schedule(q: torch.Tensor) -> str:
    s = _settings.get("ring_schedule")
    if s == "auto":
        return "gather" if q.is_cuda else "relay"
    return s


@Profiler.prof_func("compact._compact_ring_fwd")
def _compact_ring_fwd(q, k, v, dropout_p=0, softmax_scale=None, causal=True, window_size=(-1, -1), alibi_slopes=None,
                      return_attn_probs=None, deterministic=False, attn_layer=None, group=None, joint_tensor_key=None,
                      joint_tensor_value=None, joint_strategy="none", mod_idx=None, current_iter=None):
    assert alibi_slopes is None
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    # steady-state lane: this layer's exchange is a bound native plan and nothing it was bound against has changed -> two
    # native calls around the local attention block, no per-call bookkeeping (the reference rebuilds keys, shapes and
    # communicator state on every call, ring.py:172-190)
    st = _steady.get((mod_idx, id(group) if group is not None else None))
    if st is not None and joint_tensor_key is None and joint_tensor_value is None:
        cfg = compact_config()
        ctype = cfg.compress_func(mod_idx, current_iter)
        if st.matches(q, k, v, ctype, cfg, causal, dropout_p):
            return st.run(q, k, v, softmax_scale)
    jmode = _joint_mode(joint_tensor_key, joint_tensor_value, joint_strategy)
    comm = RingComm(group)
    rank, world = comm.rank, comm.world_size
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    ctype = compact_config().compress_func(mod_idx, current_iter)
    kshape, vshape = k.shape, v.shape
    kkey = lambda r: f"{mod_idx}-{r % world}-k"   # noqa: E731
    vkey = lambda r: f"{mod_idx}-{r % world}-v"   # noqa: E731

    def attend(out, lse, kk, vv, step):
        if causal and step > rank:
            return out, lse
        kk, vv = _with_joint(kk, vv, joint_tensor_key, joint_tensor_value, jmode, step, world)
        bo, bl = block_attention(q, kk, vv, dropout_p, softmax_scale, causal=causal and step == 0)
        return update_out_and_lse(out, lse, bo, bl)

    out = lse = None
    if world == 1 or _schedule(q) == "relay":
        k_send = compact_compress(kkey(rank), k, ctype, update_cache=True)
        v_send = compact_compress(vkey(rank), v, ctype, update_cache=True)
        for step in range(world):
            if step + 1 != world:
                buf_k = comm.send_recv(k_send)
                buf_v = comm.send_recv(v_send)
                comm.commit()
            if step != 0:
                src = rank - step
                k = compact_decompress(kkey(src), k_send, ctype, kshape, update_cache=True).contiguous()
                v = compact_decompress(vkey(src), v_send, ctype, vshape, update_cache=True).contiguous()
            out, lse = attend(out, lse, k, v, step)
            if step + 1 != world:
                with Profiler.scope("compact.ring.wait"):
                    comm.wait()
                k_send, v_send = buf_k, buf_v
    else:
        out, lse = _gather_schedule(q, k, v, ctype, mod_idx, rank, world, group, kkey, vkey, attend)

    out = out.to(q.dtype)
    lse = lse.squeeze(dim=-1).transpose(1, 2)
    if compact_config().check_cache_consistency:
        compact_cache().check_consistency(group=group)
    return out, lse, None


_xbuf = {}
_xstreams = {}
_steady = {}
_prebegun = {}           # (layer, group) -> the lane epoch compact_fwd already published from the caller's stream (auto lane)


class _SteadyLayer:
    """What `_gather_schedule` established for a layer, frozen: valid while the config object, the state arena generation, the
    codec and the tensor geometry stay what they were (any change falls back to the general path, which re-binds)."""

    def __init__(self, ex, q, k, v, ctype, cfg, rank, world):
        cache = compact_cache()
        self.ex, self.ctype, self.cfg, self.rank, self.world = ex, ctype, cfg, rank, world
        self.qs, self.ks, self.vs, self.device = q.shape, k.shape, v.shape, q.device
        self.gen, self.cver, self.sig = cm._generation, cache.version, ex.sig
        self.flags = (cfg.error_feedback, cfg.log_compress_stats, cfg.check_cache_consistency, cfg.simulate_compress, cfg.compress_residual)
        self.N, self.C = cm._nc_shape(k.shape)
        self.kk, self.vk = ex.kkeys[rank], ex.vkeys[rank]
        self.last_key = ex.vkeys[ex.peers[-1]]
        self._key = None             # set by whoever files the layer under _steady

    def matches(self, q, k, v, ctype, cfg, causal, dropout_p) -> bool:
        # (cheap comparisons only: this runs once per layer and step in front of ONE native call)
        ex = self.ex
        if ctype is not self.ctype or cfg is not self.cfg or causal or dropout_p or (ex.plan is None and ex.xop is None):
            return False
        if q.shape != self.qs or k.shape != self.ks or v.shape != self.vs or q.device != self.device:
            return False
        if cm._generation != self.gen or cm._cache.version != self.cver or ex.sig is not self.sig:
            return False
        if (cfg.error_feedback, cfg.log_compress_stats, cfg.check_cache_consistency, cfg.simulate_compress, cfg.compress_residual) != self.flags \
                or cfg.check_cache_consistency:
            return False
        return q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and not _profiler.enabled and not _collector_live()

    def _fast_ok(self, q) -> bool:
        """The lean host path applies when the fused SDPA op takes this shape and returns the layouts the native merge reads
        (decided once per layer by trying it)."""
        ok = getattr(self, "_fast", None)
        if ok is not None:
            return ok
        self._fast = False
        try:
            if q.dtype == torch.float16 and q.shape[-1] % 8 == 0 and q.shape[-1] <= 512 and q.dim() == 4:
                from .. import _lib, codecs
                sdpa = torch.ops.aten._scaled_dot_product_flash_attention
                qt = q.transpose(1, 2)
                r = sdpa(qt, qt, qt, 0.0, False, False, scale=1.0)
                if (r[0].dtype == torch.float16 and r[0].transpose(1, 2).is_contiguous() and r[1].dtype == torch.float32 and r[1].is_contiguous()
                        and r[0].data_ptr() % 16 == 0):
                    dev = q.device.index if q.device.index is not None else torch.cuda.current_device()
                    self._sdpa, self._merge, self._ctx = sdpa, _lib.load().cfx_attn_merge_wait, codecs.context(dev)
                    self._peer_t = [(kk.transpose(1, 2), vv.transpose(1, 2)) for kk, vv in self.ex.peer_views]
                    self._fast = True
        except (RuntimeError, NotImplementedError):
            self._fast = False
        return self._fast

    def run(self, q, k, v, softmax_scale):
        ex = self.ex
        sh = torch.cuda.current_stream(q.device).cuda_stream
        if ex.xop is not None:
            # ONE native call: the layer's whole exchange on this stream, then the blocks over own K,V and the peers' states
            xop = ex.xop
            epoch = None
            if xop.lowrank and xop.lane_capable():
                from .. import lanes
                dev = q.device.index if q.device.index is not None else torch.cuda.current_device()
                if lanes.on_compute_stream(dev):
                    # low-rank family on the lane: the factor chain + publish-and-wait here, the peers' reconstructions on the exchange lane
                    epoch = xop.run(k, v, sh, lane=True)
                else:
                    xop.run(k, v, sh)
            else:
                xop.run(k, v, sh)
            cm._current_cache_key = self.last_key
            if epoch is not None:
                return self._lowrank_lane_blocks(q, k, v, softmax_scale, sh, epoch)
            if self._fast_ok(q):
                sdpa, merge, ctx = self._sdpa, self._merge, self._ctx
                B, S, H, D = q.shape
                qt = q.transpose(1, 2)
                res = sdpa(qt, k.transpose(1, 2), v.transpose(1, 2), 0.0, False, False, scale=softmax_scale)
                out = torch.empty((B, S, H, D), dtype=torch.float32, device=q.device)
                lse = torch.empty((B, S, H, 1), dtype=torch.float32, device=q.device)
                op, lp = out.data_ptr(), lse.data_ptr()
                first = 1
                for kt, vt in [(None, None)] + self._peer_t:
                    if kt is not None:
                        res = sdpa(qt, kt, vt, 0.0, False, False, scale=softmax_scale)
                    if merge(ctx, op, lp, res[0].data_ptr(), res[1].data_ptr(), B, S, H, D, 1, first, None, 0, sh) != 0:
                        raise RuntimeError("cfx_attn_merge failed: " + (ex.xop.lib.cfx_last_error_string(ctx) or b"").decode())
                    first = 0
                return out.to(q.dtype), lse.squeeze(dim=-1).transpose(1, 2), None
            bo, bl = block_attention(q, k, v, 0.0, softmax_scale, causal=False)
            out, lse = update_out_and_lse(None, None, bo, bl)
            for kk, vv in ex.peer_views:
                bo, bl = block_attention(q, kk, vv, 0.0, softmax_scale, causal=False)
                out, lse = update_out_and_lse(out, lse, bo, bl)
            return out.to(q.dtype), lse.squeeze(dim=-1).transpose(1, 2), None
        if ex.lane:
            # exchange lane: ONE native call issues the layer's whole chain on the exchange stream; the compute stream never sees
            # an event - the merge launch of block s also waits (in-kernel, on a flag) for peer s+1's reconstruction
            epoch = _prebegun.pop((self._key), None)        # (auto lane: already published, from the caller's stream, before the hand-over)
            if epoch is None:
                epoch = ex.lane_begin(sh)                   # "K, V exist" on the compute stream ...
            fast = self._fast_ok(q)
            if fast:
                # lean host path (the step is host-bound long before it is GPU-bound: 8 attention + 8 merge calls per layer): cached
                # transposed views of the peers' states, the fused SDPA op called directly, the merge through its cached entry point
                sdpa, merge, ctx = self._sdpa, self._merge, self._ctx
                B, S, H, D = q.shape
                qt = q.transpose(1, 2)
                res = sdpa(qt, k.transpose(1, 2), v.transpose(1, 2), 0.0, False, False, scale=softmax_scale)
                ex.run_lane_head(k, v)                     # ... the chain up to the first peer's flag is issued while the local block runs
                cm._current_cache_key = self.last_key
                out = torch.empty((B, S, H, D), dtype=torch.float32, device=q.device)
                lse = torch.empty((B, S, H, 1), dtype=torch.float32, device=q.device)
                op, lp, last = out.data_ptr(), lse.data_ptr(), self.world - 1
                if merge(ctx, op, lp, res[0].data_ptr(), res[1].data_ptr(), B, S, H, D, 1, 1, ex.flag_ptr(1), epoch, sh) != 0:
                    raise RuntimeError("cfx_attn_merge_wait failed: " + (ex._lib.cfx_last_error_string(ctx) or b"").decode())
                keep = [res]
                for s_, (kt, vt) in enumerate(self._peer_t, start=1):
                    res = sdpa(qt, kt, vt, 0.0, False, False, scale=softmax_scale)
                    keep.append(res)
                    if s_ == 1:
                        ex.run_lane_tail()                 # the rest of the chain, behind the first peer's block on the compute stream
                    if merge(ctx, op, lp, res[0].data_ptr(), res[1].data_ptr(), B, S, H, D, 1, 0,
                             None if s_ == last else ex.flag_ptr(s_ + 1), epoch, sh) != 0:
                        raise RuntimeError("cfx_attn_merge_wait failed: " + (ex._lib.cfx_last_error_string(ctx) or b"").decode())
                return out.to(q.dtype), lse.squeeze(dim=-1).transpose(1, 2), None
            bo, bl = block_attention(q, k, v, 0.0, softmax_scale, causal=False)     # ... the local block behind it ...
            ex.run_lane(k, v, None)                        # ... and the chain's dozen launches are issued while that block runs
            cm._current_cache_key = self.last_key
            out, lse = update_out_and_lse(None, None, bo, bl, wait=(ex.flag_ptr(1), epoch))
            last = self.world - 1
            for s, (kk, vv) in enumerate(ex.peer_views, start=1):
                bo, bl = block_attention(q, kk, vv, 0.0, softmax_scale, causal=False)
                out, lse = update_out_and_lse(out, lse, bo, bl, wait=None if s == last else (ex.flag_ptr(s + 1), epoch))
            return out.to(q.dtype), lse.squeeze(dim=-1).transpose(1, 2), None
        ex.run_front(k, v, sh)
        if not self.flags[0] and not ex.plan_updates_state:    # no error feedback: the state becomes the activation (main.py:240-243)
            cache = compact_cache()
            cache.put(self.kk, k.view(self.N, self.C), None)
            cache.put(self.vk, v.view(self.N, self.C), None)
        bo, bl = block_attention(q, k, v, 0.0, softmax_scale, causal=False)
        out, lse = update_out_and_lse(None, None, bo, bl)
        ex.run_back(sh)
        cm._current_cache_key = self.last_key
        for kk, vv in ex.peer_views:
            bo, bl = block_attention(q, kk, vv, 0.0, softmax_scale, causal=False)
            out, lse = update_out_and_lse(out, lse, bo, bl)
        return out.to(q.dtype), lse.squeeze(dim=-1).transpose(1, 2), None


def _lowrank_lane_blocks(self, q, k, v, softmax_scale, sh, epoch):
    """The attention blocks of a low-rank layer whose peers are being reconstructed on the exchange lane: the local block first (the
    chain's launches are issued while it runs), then peer s's block behind a merge launch that waited in-kernel for peer s's flag."""
    xop = self.ex.xop
    last = self.world - 1
    if self._fast_ok(q):
        sdpa, merge, ctx = self._sdpa, self._merge, self._ctx
        B, S, H, D = q.shape
        qt = q.transpose(1, 2)
        res = sdpa(qt, k.transpose(1, 2), v.transpose(1, 2), 0.0, False, False, scale=softmax_scale)
        xop.lane_chain(1, 2)                           # peers 1 and 2 while the local block runs, peer s + 2 behind peer s's block
        out = torch.empty((B, S, H, D), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, S, H, 1), dtype=torch.float32, device=q.device)
        op, lp = out.data_ptr(), lse.data_ptr()
        if merge(ctx, op, lp, res[0].data_ptr(), res[1].data_ptr(), B, S, H, D, 1, 1, xop.lane_flag(1), epoch, sh) != 0:
            raise RuntimeError("cfx_attn_merge_wait failed: " + (xop.lib.cfx_last_error_string(ctx) or b"").decode())
        keep = [res]
        for s_, (kt, vt) in enumerate(self._peer_t, start=1):
            res = sdpa(qt, kt, vt, 0.0, False, False, scale=softmax_scale)
            keep.append(res)
            xop.lane_chain(s_ + 2, 1)
            if merge(ctx, op, lp, res[0].data_ptr(), res[1].data_ptr(), B, S, H, D, 1, 0,
                     None if s_ == last else xop.lane_flag(s_ + 1), epoch, sh) != 0:
                raise RuntimeError("cfx_attn_merge_wait failed: " + (xop.lib.cfx_last_error_string(ctx) or b"").decode())
        return out.to(q.dtype), lse.squeeze(dim=-1).transpose(1, 2), None
    bo, bl = block_attention(q, k, v, 0.0, softmax_scale, causal=False)
    xop.lane_chain(1, 2)
    out, lse = update_out_and_lse(None, None, bo, bl, wait=(xop.lane_flag(1), epoch))
    for s_, (kk, vv) in enumerate(self.ex.peer_views, start=1):
        bo, bl = block_attention(q, kk, vv, 0.0, softmax_scale, causal=False)
        xop.lane_chain(s_ + 2, 1)
        out, lse = update_out_and_lse(out, lse, bo, bl, wait=None if s_ == last else (xop.lane_flag(s_ + 1), epoch))
    return out.to(q.dtype), lse.squeeze(dim=-1).transpose(1, 2), None


_SteadyLayer._lowrank_lane_blocks = _lowrank_lane_blocks


def _exchange_stream(device) -> "torch.cuda.Stream":
    s = _xstreams.get(device)
    if s is None:
        s = _xstreams[device] = torch.cuda.Stream(device, priority=int(_settings.get("ring_exchange_priority")))
    return s


class _RawHalves:
    """fp16 view of device memory that torch did not allocate (cfx_ipc_alloc / cfx_ipc_open), through the CUDA array interface"""

    def __init__(self, ptr: int, n_halves: int):
        self.__cuda_array_interface__ = {"shape": (n_halves,), "typestr": "<f2", "data": (ptr, False), "version": 2}


def _raw_halves(ptr: int, n_halves: int, device) -> torch.Tensor:
    return torch.as_tensor(_RawHalves(ptr, n_halves), device=device)


class _LayerExchange:
    """Everything about one layer's packet exchange that does not change from step to step: the exchange buffers, the
    side stream, key strings, the peers' packet views and - once the state arena is populated - the prepared native
    batches (cached pointer tables, `codecs.prepare_*`), so the steady-state host work per layer is one compress call,
    one collective and one reconstruct call."""

    def __init__(self, mod_idx, rank: int, world: int, slot: int, like: torch.Tensor, group=None):
        self.slot, self.world, self.rank, self.group = slot, world, rank, group
        self.plan = None             # native per-layer plan: compress, all-gather on the exchange stream | wait, reconstruct
        self.lane = False            # plan built for the flag-synchronised exchange lane (cfx_plan_run_lane)
        self.plan_updates_state = False   # error feedback off: the plan itself copies the activation into the state, in stream order
        self._lib = None
        self.send = torch.empty(2 * slot, dtype=torch.float16, device=like.device)
        self.recv = torch.empty(world * 2 * slot, dtype=torch.float16, device=like.device)
        self.side = _exchange_stream(like.device) if like.is_cuda else None
        self.peers = [(rank - s) % world for s in range(1, world)]
        self.kkeys = [f"{mod_idx}-{r}-k" for r in range(world)]
        self.vkeys = [f"{mod_idx}-{r}-v" for r in range(world)]
        self.sig = None
        self.comp = None
        self.dec = []
        self.peer_views = []
        self.xop = None              # the layer's exchange as ONE native op (xlayer.LayerOp): the default off the exchange lane
        self._p2p = None             # CFX_RING_P2P: {"send": own packets in IPC memory, "peer": {rank: its packets, mapped}, "flag": ptr, "peer_flag": {rank: ptr}}
        self._p2p_tried = False

    def _p2p_setup(self):
        """CFX_RING_P2P=1 (ranks of ONE node): the packets stay in cfx_ipc_alloc memory of the rank that produced them and the peers'
        reconstruction launches read them in place; the all-gather of the layer's plan becomes a publish-and-wait op
        (cfx_plan_add_p2p_sync).  Collective over the group (the handles travel once); all ranks or none."""
        self._p2p_tried = True
        if _settings.get("ring_p2p") not in ("1", "on", "auto") or self.world < 2 or not self.send.is_cuda or self.world - 1 > 15:
            return
        import ctypes
        from .. import _lib, codecs
        lib = _lib.load()
        dev = self.send.device.index if self.send.device.index is not None else torch.cuda.current_device()
        ctx = codecs.context(dev)
        halves = 2 * self.slot
        nbytes = halves * 2 + 128                                       # [K packet | V packet] + two flag words on lines of their own
        ptr, handle = ctypes.c_void_p(), ctypes.create_string_buffer(64)
        mine = bytes(handle.raw) if lib.cfx_ipc_alloc(ctx, nbytes, ctypes.byref(ptr), handle) == 0 else None
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=self.group)
        opened, good = {}, all(h is not None for h in handles)
        if good:
            for r in range(self.world):
                if r != self.rank:
                    pq = ctypes.c_void_p()
                    if lib.cfx_ipc_open(ctx, handles[r], ctypes.byref(pq)) != 0:
                        good = False
                        break
                    opened[r] = pq.value
        votes = [None] * self.world
        dist.all_gather_object(votes, bool(good), group=self.group)
        if not all(votes):
            for pq in opened.values():
                lib.cfx_ipc_close(ctx, ctypes.c_void_p(pq))
            if mine is not None:
                lib.cfx_ipc_free(ctx, ptr)
            return
        d = self.send.device
        self._lib = lib
        self._p2p = {"ptr": ptr.value, "peer_ptr": dict(opened), "send": _raw_halves(ptr.value, halves, d),
                     "peer": {r: _raw_halves(pq, halves, d) for r, pq in opened.items()},
                     "flag": ptr.value + halves * 2, "peer_flag": {r: pq + halves * 2 for r, pq in opened.items()}}

    def packet(self, r: int, kv: int, n_half: int) -> torch.Tensor:
        o = (2 * r + kv) * self.slot
        return self.recv[o:o + n_half]

    def bind(self, sig, cid, param, N, C, n_half, kshape, vshape, ef):
        """(Re)build the pointer tables against the current state arena."""
        from .. import codecs
        cache = compact_cache()

        def state(key):
            b = cache.get_base(key)
            assert b is not None, f"no cached base for key {key}: a WARMUP step must precede residual compression"
            return b
        own = [state(self.kkeys[self.rank]), state(self.vkeys[self.rank])]
        self._drop_xop()
        if self._xop_wanted(cid, ef):
            # ONE native op per layer (xlayer.LayerOp): compress ; exchange ; reconstruct all peers - on the caller's stream, in front of
            # the local attention block.  The states are updated in place; the consumer reads them as the peers' K,V.
            from . import xlayer
            self._drop_plan()
            self.lane, self.plan_updates_state = False, not ef
            peers, self.peer_views = [], []
            for r in self.peers:
                bk, bv = state(self.kkeys[r]), state(self.vkeys[r])
                peers.append((r, bk, bv))
                self.peer_views.append((bk.view(kshape), bv.view(vshape)))
            self.xop = xlayer.LayerOp(("ring", self.kkeys[self.rank], id(self.group) if self.group is not None else None), cid, param, N, C,
                                      self.rank, self.world, self.group, self.send.device, own, peers, own_update="ef" if ef else "x")
            self.comp, self.dec = None, []
            self.sig = sig
            return
        if not self._p2p_tried:
            self._p2p_setup()
        p2p = self._p2p
        own_src = p2p["send"] if p2p else self.send
        own_pkts = [own_src[:n_half], own_src[self.slot:self.slot + n_half]]
        # the sender's error-feedback update IS the receiver's dequant+add run on its own packet (fastpath.py:88-120), so
        # it rides in the batched reconstruction launch (own K,V + all peers' K,V = 16 tensors at W = 8) instead of a
        # launch of its own; the compress sequence is then stats -> finalize only
        self.comp = codecs.prepare_compress(cid, own, [None, None], own_pkts, N, C, param, update_cache=False, ef=ef)
        bases, pkts, self.peer_views = list(own), list(own_pkts), []
        if not ef:
            # no error feedback: the state becomes the activation itself (main.py:240-243) - done by the caller
            bases, pkts = [], []
        for r in self.peers:
            bk, bv = state(self.kkeys[r]), state(self.vkeys[r])
            bases += [bk, bv]
            if p2p:                 # read in place from the peer's own buffer
                pkts += [p2p["peer"][r][:n_half], p2p["peer"][r][self.slot:self.slot + n_half]]
            else:
                pkts += [self.packet(r, 0, n_half), self.packet(r, 1, n_half)]
            self.peer_views.append((bk.view(kshape), bv.view(vshape)))
        step = codecs.CFX_MAX_BATCH
        self.dec = [codecs.prepare_decompress(cid, pkts[i:i + step], bases[i:i + step], bases[i:i + step], N, C, param)
                    for i in range(0, len(bases), step)]
        self.sig = sig
        self._bind_native(cid, param, N, C, own, own_pkts, bases, pkts, ef)

    def _drop_plan(self):
        if self.plan is not None and self._lib is not None:
            self._lib.cfx_plan_destroy(self.plan)
        self.plan = None

    def _drop_xop(self):
        if self.xop is not None:
            self.xop.close()
        self.xop = None

    def _xop_wanted(self, cid, ef: bool = True) -> bool:
        """The layer's exchange as one native op on the caller's stream - unless the caller runs on the exchange lane's compute stream
        (then the chain runs beside the attention blocks on the CU-masked exchange stream) or asked for one of the multi-launch forms."""
        from . import xlayer
        from .. import lanes
        xmode = _settings.get("ring_exchange_stream")     # auto | xlayer | lane | chain | side | main
        if xmode not in ("auto", "xlayer") or not xlayer.usable(cid, self.world, self.send.is_cuda, ef=ef):
            return False
        dev = self.send.device.index if self.send.device.index is not None else torch.cuda.current_device()
        # (the low-rank family has no chain of its own on the lane: on the compute stream too it takes the layer op, which leaves the peers'
        # reconstructions to the exchange lane - xlayer.LayerOp.run(lane=True))
        return xmode == "xlayer" or cid >= 100 or not lanes.on_compute_stream(dev)

    def close(self):
        """Release what the layer holds outside torch's allocator: native plans and (legacy CFX_RING_P2P chain) its IPC buffer + mappings."""
        self._drop_plan()
        self._drop_xop()
        p2p, self._p2p = self._p2p, None
        if p2p is not None and self._lib is not None:
            import ctypes
            from .. import codecs
            dev = self.send.device.index if self.send.device.index is not None else torch.cuda.current_device()
            ctx = codecs.context(dev)
            torch.cuda.synchronize(dev)
            for pq in p2p["peer_ptr"].values():
                self._lib.cfx_ipc_close(ctx, ctypes.c_void_p(pq))
            self._lib.cfx_ipc_free(ctx, ctypes.c_void_p(p2p["ptr"]))

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def _bind_native(self, cid, param, N, C, own, own_pkts, bases, pkts, ef):
        """The layer's exchange as ONE native plan replayed in two host calls: [compress K,V ; all-gather on the exchange
        stream] - local attention block - [wait ; batched reconstruction].  The collective is issued by libcfx's own
        communicator (the reference drives torch.distributed P2P per hop, ring.py:193-195,265-269; through torch.distributed
        the per-layer all-gather costs ~50 us of host time, more than the GPU work it overlaps)."""
        self._drop_plan()
        self.lane = False
        self.plan_updates_state = False
        mode = _settings.get("ring_exchange")
        if mode == "torch" or not self.send.is_cuda or cid >= 100:
            return
        import ctypes
        from .. import _lib, codecs
        from ..exchange import native_comm_for
        dev = self.send.device.index if self.send.device.index is not None else torch.cuda.current_device()
        p2p = self._p2p
        comm = None if p2p else native_comm_for(self.group, dev)
        if comm is None and not p2p:
            assert mode != "native", "CFX_RING_EXCHANGE=native but the library-owned communicator cannot be created"
            return
        lib = self._lib = _lib.load()
        ctx = codecs.context(dev)
        plan = lib.cfx_plan_create(ctx)
        # ONE exchange stream per device, shared by every layer's plan (a stream per plan would be a hardware queue per layer)
        xmode = _settings.get("ring_exchange_stream")     # auto | xlayer | lane | chain | side | main
        if xmode in ("auto", "xlayer"):
            xmode = "lane"           # (here: the caller is on the lane's compute stream, or the shape has no layer op)
        assert xmode in ("lane", "chain", "side", "main"), "CFX_RING_EXCHANGE_STREAM must be auto | xlayer | lane | chain | side | main"
        self._async = xmode == "chain"
        self.lane = xmode == "lane"
        xs_handle = None
        if xmode == "main":
            assert lib.cfx_plan_set_exchange_stream(plan, 0) == 0
        else:
            from .. import lanes
            # the CU-masked exchange stream when the model runs on the lane's compute stream (disjoint CU sets), else an unmasked one
            if self.lane:
                # flags order the two streams, so the exchange stream must OWN its hardware queue (lanes.dedicated_stream)
                # (on the lane's compute stream: the CU-masked exchange stream.  Otherwise a high-priority pool stream: non-blocking, and
                # high-priority streams do not share a hardware queue with normal-priority ones - a CU-masked stream would own its queue too,
                # but hipExtStreamCreateWithCUMask makes BLOCKING streams, which synchronise implicitly with the null stream most models run on)
                xstream = lanes.exchange_stream(dev) if lanes.on_compute_stream(dev) else _exchange_stream(self.send.device)
            else:
                xstream = _exchange_stream(self.send.device)
            xs_handle = xstream.cuda_stream
            assert lib.cfx_plan_use_exchange_stream(plan, xs_handle) == 0
        # the compress launches run on the exchange stream in the lane / chain modes: their statistics workspace belongs to THAT stream
        ws = codecs.workspace(cid, N, C, param, 2, dev, stream_handle=xs_handle if (self.lane or self._async) else None)
        self._plan_keep = (ws, list(own), list(own_pkts), list(bases), list(pkts), comm)
        # error feedback off: the state becomes the activation (main.py:240-243) - done by the compress op itself, in stream order
        # behind the statistics pass that reads the old state (a copy issued from Python on another stream would race with it)
        self.plan_updates_state = not ef
        flags = 0 if ef else (_lib.FLAG_NO_EF | _lib.FLAG_UPDATE_CACHE)
        c = (_lib.CompItem * 2)(*[_lib.CompItem(own[i].data_ptr(), own[i].data_ptr(), None if ef else own[i].data_ptr(), own_pkts[i].data_ptr())
                                  for i in range(2)])
        wsp, wsn = (None, 0) if ws is None else (ws.data_ptr(), ws.numel())
        step = codecs.CFX_MAX_BATCH

        def dec_items(lo, hi):
            return [_lib.DecompItem(p_.data_ptr(), b_.data_ptr(), b_.data_ptr()) for p_, b_ in zip(pkts[lo:hi], bases[lo:hi])]

        def exchange_op():
            """the op between compress and reconstruction: ncclAllGather, or (p2p) publish this rank's word and wait for the peers'"""
            if not p2p:
                return lib.cfx_plan_add_all_gather(plan, comm.handle, self.send.data_ptr(), self.recv.data_ptr(), 2 * self.slot * 2)
            pf = (ctypes.c_void_p * (self.world - 1))(*[p2p["peer_flag"][r] for r in sorted(p2p["peer_flag"])])
            return lib.cfx_plan_add_p2p_sync(plan, p2p["flag"], self.world - 1, pf)

        def done_op():
            """p2p: nobody rewrites its packets before every peer has finished reading them (a second word per rank)"""
            if not p2p:
                return 0
            pf = (ctypes.c_void_p * (self.world - 1))(*[p2p["peer_flag"][r] + 64 for r in sorted(p2p["peer_flag"])])
            return lib.cfx_plan_add_p2p_sync(plan, p2p["flag"] + 64, self.world - 1, pf)
        if self.lane:
            # flags: 0 = the activations exist (set on the compute stream), s = 1..W-1 = peer of ring step s reconstructed, W = chain done
            W = self.world
            self._flags = lib.cfx_plan_flags(plan, W + 1)
            ok = bool(self._flags)
            ok = ok and lib.cfx_plan_add_flag_wait(plan, 0) == 0
            ok = ok and lib.cfx_plan_add_compress(plan, cid, N, C, param, flags, 2, c, wsp, wsn) == 1
            ok = ok and exchange_op() == 2
            n_own = 2 if ef else 0                 # bases / pkts start with the rank's own K,V when error feedback is on
            for s_ in range(1, W):                 # just in time: peer s's K,V in the order the attention blocks visit them; its flag is
                lo = n_own + 2 * (s_ - 1)          # published by the NEXT launch of the chain as the first thing it does (no launch of its own)
                it = dec_items(lo, lo + 2)
                op = lib.cfx_plan_add_decompress(plan, cid, N, C, param, 2, (_lib.DecompItem * 2)(*it))
                ok = ok and op >= 0 and (s_ == 1 or lib.cfx_plan_set_pre_flag(plan, op, s_ - 1) == 0)
            if ef:                                 # the rank's own error-feedback update: nobody waits for it before the next step
                it = dec_items(0, 2)
                op = lib.cfx_plan_add_decompress(plan, cid, N, C, param, 2, (_lib.DecompItem * 2)(*it))
                ok = ok and op >= 0 and lib.cfx_plan_set_pre_flag(plan, op, W - 1) == 0
            else:
                ok = ok and lib.cfx_plan_add_flag_set(plan, W - 1) >= 0
            ok = ok and lib.cfx_plan_add_flag_set(plan, W) >= 0
            ok = ok and done_op() >= 0
            self._epoch = ctypes.c_uint(0)
        else:
            ok = lib.cfx_plan_add_compress(plan, cid, N, C, param, flags, 2, c, wsp, wsn) == 0
            g0 = exchange_op()
            # (front = ops [0, 2), back = the rest: with the collective the wait op is op 2; the publish-and-wait op needs none, and
            # run_back then starts at the first reconstruction - a no-op flag set keeps the indices the same)
            ok = ok and g0 == 1 and (lib.cfx_plan_add_wait(plan, g0) if not p2p else lib.cfx_plan_add_p2p_sync(plan, p2p["flag"] + 96, 0, None)) == 2
            for i in range(0, len(bases), step):
                items = dec_items(i, i + step)
                ok = ok and lib.cfx_plan_add_decompress(plan, cid, N, C, param, len(items), (_lib.DecompItem * len(items))(*items)) >= 0
            ok = ok and done_op() >= 0
        ok = ok and lib.cfx_plan_finalize(plan) == 0
        if not ok:
            lib.cfx_plan_destroy(plan)
            raise _lib.CfxError("building the layer's native exchange plan failed: " + (lib.cfx_last_error_string(ctx) or b"").decode())
        self.plan, self._ctx = plan, ctx
        self._n_ops = lib.cfx_plan_size(plan)
        self._xs = (ctypes.c_void_p * 2)()

    def flag_ptr(self, i: int) -> int:
        return self._flags + 64 * i

    def lane_begin(self, sh) -> int:
        rc = self._lib.cfx_plan_lane_begin(self.plan, 0, sh, self._epoch)
        if rc != 0:
            raise RuntimeError("native exchange lane failed: " + (self._lib.cfx_last_error_string(self._ctx) or b"").decode())
        return self._epoch.value

    def run_lane_head(self, k, v) -> None:
        """After `lane_begin`: the chain up to the launch that publishes the first peer's flag (wait, compress, all-gather, peers 1 and
        2) - what the first merge waits for.  `run_lane_tail` issues the rest; splitting the dozen launches lets the caller put the
        next attention block on the compute stream in between instead of behind 50 us of host time."""
        self._xs[0], self._xs[1] = k.data_ptr(), v.data_ptr()
        n = min(5, self._n_ops)
        rc = self._lib.cfx_plan_run_lane(self.plan, 0, n, self._xs, 2, 0, None, self._epoch)
        if rc != 0:
            raise RuntimeError("native exchange lane failed: " + (self._lib.cfx_last_error_string(self._ctx) or b"").decode())

    def run_lane_tail(self) -> None:
        n = min(5, self._n_ops)
        if self._n_ops > n and self._lib.cfx_plan_run_lane(self.plan, n, self._n_ops - n, None, 0, 0, None, self._epoch) != 0:
            raise RuntimeError("native exchange lane failed: " + (self._lib.cfx_last_error_string(self._ctx) or b"").decode())

    def run_lane(self, k, v, sh) -> int:
        """The layer's whole chain on the exchange lane, one host call; returns the epoch its flags will carry.  sh = None: the
        ready flag was already launched (`lane_begin`)."""
        self._xs[0], self._xs[1] = k.data_ptr(), v.data_ptr()
        rc = self._lib.cfx_plan_run_lane(self.plan, 0, self._n_ops, self._xs, 2, 0, sh, self._epoch)
        if rc != 0:
            raise RuntimeError("native exchange lane failed: " + (self._lib.cfx_last_error_string(self._ctx) or b"").decode())
        return self._epoch.value

    def run_front(self, k, v, sh):
        if self.xop is not None:
            self.xop.run(k, v, sh)
            return
        if self.lane:
            # (two calls: cfx_plan_run_lane takes compute_stream = NULL to mean "cfx_plan_lane_begin was already called", and the
            # legacy default stream IS the NULL handle - the ready flag would never be launched and the chain's wait would compare
            # against a stale epoch: K,V read before the compute stream has produced them)
            self.lane_begin(sh)
            self._last_epoch = self.run_lane(k, v, None)
            return
        self._xs[0], self._xs[1] = k.data_ptr(), v.data_ptr()
        if self._async:
            # the whole chain - compress, all-gather, reconstruction - on the exchange stream, beside the local attention block
            rc = self._lib.cfx_plan_run_async(self.plan, 0, self._n_ops, self._xs, 2, sh)
        else:
            rc = self._lib.cfx_plan_run_x(self.plan, 0, 2, self._xs, 2, sh)
        if rc != 0:
            raise RuntimeError("native exchange (compress + all-gather) failed: " + (self._lib.cfx_last_error_string(self._ctx) or b"").decode())

    def run_back(self, sh):
        if self.xop is not None:
            return                    # (the layer op reconstructed every peer in front of the local block, in stream order)
        if self.lane:
            # general path: one wait for the whole chain (the steady-state lane waits per peer, inside the merge launches)
            from .attention import flag_wait
            flag_wait((self.flag_ptr(self.world), self._last_epoch), self.send.device)
            return
        if self._async:
            rc = self._lib.cfx_plan_join(self.plan, sh)
        else:
            rc = self._lib.cfx_plan_run(self.plan, 2, self._n_ops - 2, sh)
        if rc != 0:
            raise RuntimeError("native exchange (wait + reconstruct) failed: " + (self._lib.cfx_last_error_string(self._ctx) or b"").decode())


def _layer_exchange(mod_idx, rank: int, world: int, slot: int, like: torch.Tensor, group=None) -> _LayerExchange:
    key = (mod_idx, rank, world, slot, like.device, id(group) if group is not None else None)
    ex = _xbuf.get(key)
    if ex is None:
        ex = _xbuf[key] = _LayerExchange(mod_idx, rank, world, slot, like, group)
    return ex


def _collector_live() -> bool:
    inst = _collector_mod.instance
    return inst is None or inst.enabled


def _gather_schedule(q, k, v, ctype, mod_idx, rank, world, group, kkey, vkey, attend):
    """One all-gather of [K packet | V packet] on a side stream + one batched reconstruction of all peers."""
    cfg = compact_config()
    kshape, vshape = k.shape, v.shape
    N, C = cm._nc_shape(k.shape)
    warm = ctype == T.WARMUP
    native = (not warm) and (not cfg.simulate_compress) and cfg.compress_residual == 1
    cid, param = cm._native(ctype) if native else (0, 0)
    n_half = cm._packet_halves(cid, param, N, C) if native else \
        (N * C if (warm or cfg.simulate_compress) else cm._packet_halves(*cm._native(ctype), N, C))
    slot = (n_half + 127) // 128 * 128
    ex = _layer_exchange(mod_idx, rank, world, slot, k, group)
    send, recv, side = ex.send, ex.recv, ex.side
    # steady state: K and V in ONE native compress sequence straight into the send slots, in-place EF state update
    # (the low-rank family - ids >= 100 - has the one-op form only: where that is not wanted, e.g. on the exchange lane, the general path below)
    fast = (native and (cid < 100 or ex._xop_wanted(cid, bool(cfg.error_feedback))) and not cfg.log_compress_stats and v.shape == k.shape
            and not compact_cache().quantize)
    cur = torch.cuda.current_stream(k.device) if side is not None else None
    sh = cur.cuda_stream if cur is not None else None
    if fast:
        cache = compact_cache()
        sig = (cm._generation, cache.version, cid, param, N, C, tuple(kshape), cfg.error_feedback)
        if ex.sig != sig:
            ex.bind(sig, cid, param, N, C, n_half, kshape, vshape, cfg.error_feedback)
        native_x = ex.plan is not None or ex.xop is not None
        with Profiler.scope("compact.compress_batch"):
            if native_x:
                ex.run_front(k, v, sh)         # compress K,V + the all-gather on the exchange stream: one host call
            else:
                ex.comp((k, v), sh)
        if not cfg.error_feedback and not (native_x and ex.plan_updates_state):
            cache.put(ex.kkeys[rank], k.view(N, C), None)
            cache.put(ex.vkeys[rank], v.view(N, C), None)
        cm._current_cache_key = ex.vkeys[rank]
        live = _collector_live()
        if live:
            cache.touch(ex.kkeys[rank])
            cache.touch(ex.vkeys[rank])
    else:
        if native:
            cm.compact_bind_packet(kkey(rank), send[:n_half])
            cm.compact_bind_packet(vkey(rank), send[slot:slot + n_half])
        pk = compact_compress(kkey(rank), k, ctype, update_cache=True)
        pv = compact_compress(vkey(rank), v, ctype, update_cache=True)
        if pk.reshape(-1).data_ptr() != send.data_ptr():
            send[:n_half].copy_(pk.reshape(-1))
            send[slot:slot + n_half].copy_(pv.reshape(-1))
    # exchange on the side stream, overlapped with the local attention block
    if fast and native_x:
        pass
    elif side is not None:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            with Profiler.scope("compact.all_gather", stream=side):
                dist.all_gather_into_tensor(recv, send, group=group)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    out, lse = attend(None, None, k, v, 0)
    if side is not None and not (fast and native_x):
        cur.wait_stream(side)
    peers = ex.peers
    if fast:
        with Profiler.scope("compact.decompress_batch"):
            if native_x:
                ex.run_back(sh)                # wait for the gather + ONE batched reconstruction: one host call
            else:
                for run in ex.dec:
                    run(sh)
        if live:
            for r in peers:
                cache.touch(ex.kkeys[r])
                cache.touch(ex.vkeys[r])
        cm._current_cache_key = ex.vkeys[peers[-1]]
        for step, (kk, vv) in enumerate(ex.peer_views, start=1):
            out, lse = attend(out, lse, kk, vv, step)
        if native_x and cfg.error_feedback is not None:
            skey = (mod_idx, id(group) if group is not None else None)
            _steady[skey] = _SteadyLayer(ex, q, k, v, ctype, cfg, rank, world)
            _steady[skey]._key = skey
        return out, lse
    if native and compact_cache().quantize:
        native = False           # state lives as int8 packets: go through compact_decompress (get_base / put) per tensor
    if native:
        from .. import codecs
        bases, pkts = [], []
        for r in peers:
            for kv, keyf in ((0, kkey), (1, vkey)):
                b = compact_cache().get_base(keyf(r))
                assert b is not None, f"no cached base for key {keyf(r)}"
                bases.append(b)
                pkts.append(ex.packet(r, kv, n_half))
        with Profiler.scope("compact.decompress_batch"):
            for i in range(0, len(bases), codecs.CFX_MAX_BATCH):
                j = i + codecs.CFX_MAX_BATCH
                if cid >= 100:
                    for p_, b_ in zip(pkts[i:j], bases[i:j]):
                        cm._codec_decompress(cid, param, p_, b_, b_)
                else:
                    codecs.decompress_batch(cid, pkts[i:j], bases[i:j], bases[i:j], N, C, param)
        for r in peers:
            compact_cache().put(kkey(r), compact_cache().get_base(kkey(r)), None)
            compact_cache().put(vkey(r), compact_cache().get_base(vkey(r)), None)
    for step, r in enumerate(peers, start=1):
        if native:
            kk = compact_cache().get_base(kkey(r)).view(kshape)
            vv = compact_cache().get_base(vkey(r)).view(vshape)
        else:
            kk = compact_decompress(kkey(r), ex.packet(r, 0, n_half), ctype, kshape, update_cache=True).contiguous()
            vv = compact_decompress(vkey(r), ex.packet(r, 1, n_half), ctype, vshape, update_cache=True).contiguous()
        out, lse = attend(out, lse, kk, vv, step)
    return out, lse
