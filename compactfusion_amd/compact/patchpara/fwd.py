"""Patch-parallel forward: gather every rank's K/V shard, then one attention over the full sequence.

Mirror of the reference's `xfuser/compact/patchpara/fwd.py` (`patch_gather_fwd` :20-236), three communication modes
selected by `PatchConfig`:
  compact - `compact_all_gather` of residual-compressed shards (own shard is replaced by its reconstruction too), K and V
            fused into one exchange (`compact_all_gather_kv`); with `PatchConfig(displaced_compact=True)` (extension) the
            gather of step t is consumed at step t+1 - DistriFusion's staleness on top of residual compression;
  sync    - plain all-gather of raw fp16 K and V (the "Patch Parallel" baseline);
  async   - DistriFusion: consume the buffers gathered during the PREVIOUS step (own shard fresh), launch this
            step's gather asynchronously for the next one; the first `async_warmup` steps gather synchronously.
K and V of a layer travel in ONE collective here (one contiguous [K|V] buffer) instead of two."""
from __future__ import annotations

import torch
import torch.distributed as dist

from ...prof import Profiler
from ..attention import block_attention
from ..main import allgather_cache, compact_all_gather_kv, compact_config
from .df_cache import DummyHandle
from .df_utils import PatchConfig

_buffers = {}


def _gather_raw_kv(k, v, group, world, key, async_op=False):
    """all-gather [K|V] of every rank into one persistent buffer; returns (handle, k_list, v_list)."""
    n = k.numel()
    send = torch.cat([k.reshape(-1), v.reshape(-1)])
    ent = _buffers.get(key)
    if ent is None or ent.numel() != world * 2 * n or ent.device != k.device:
        ent = torch.empty(world * 2 * n, dtype=k.dtype, device=k.device)
        _buffers[key] = ent
    handle = dist.all_gather_into_tensor(ent, send, group=group, async_op=async_op)
    ks = [ent[(2 * r) * n:(2 * r + 1) * n].view(k.shape) for r in range(world)]
    vs = [ent[(2 * r + 1) * n:(2 * r + 2) * n].view(v.shape) for r in range(world)]
    return handle, ks, vs


@Profiler.prof_func("patch_gather_fwd.gather_patch_fwd")
def patch_gather_fwd(q, k, v, dropout_p=0, softmax_scale=None, causal=True, window_size=(-1, -1), alibi_slopes=None,
                     return_attn_probs=None, deterministic=False, attn_layer=None, group=None, joint_tensor_key=None,
                     joint_tensor_value=None, joint_strategy="none", mod_idx=None, current_iter=None):
    assert alibi_slopes is None, "Alibi slopes not supported in this basic gather impl."
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    cfg = compact_config()
    assert cfg.override_with_patch_gather_fwd, "Patch gather fwd is not enabled"
    pc: PatchConfig = cfg.patch_gather_fwd_config
    assert mod_idx is not None, "mod_idx is required for caching"
    assert current_iter is not None, "current_iter is required for async logic"
    if (joint_tensor_key is None) != (joint_tensor_value is None):
        raise ValueError("joint_tensor_key and joint_tensor_value should be None or not None simultaneously.")
    if joint_tensor_key is not None and joint_strategy not in ("front", "rear", "none"):
        raise ValueError(f"joint_strategy: {joint_strategy} not supprted. supported joint strategy: ['front', 'rear', 'none']")
    joint = joint_strategy if joint_tensor_key is not None else "none"

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()

    if pc.use_compact:
        ctype = cfg.compress_func(mod_idx, current_iter)
        displaced = pc.displaced_compact and current_iter >= pc.async_warmup
        ks, vs = compact_all_gather_kv(f"{mod_idx}-k", f"{mod_idx}-v", k, v, ctype, group=group, displaced=displaced)
    elif not pc.async_comm:
        with Profiler.scope("compact.gather.all_gather_sync"):
            _, ks, vs = _gather_raw_kv(k, v, group, world, ("sync", mod_idx))
    else:
        cache = allgather_cache()
        ck = f"{mod_idx}-kv"
        with Profiler.scope("df.all_gather"):
            if current_iter < pc.async_warmup:
                _, ks, vs = _gather_raw_kv(k, v, group, world, ("df", mod_idx, current_iter & 1))
                cache.put(ck, DummyHandle(), ks + vs, k)
            else:
                if not cache.contains(ck):
                    raise RuntimeError(f"DistriFusion cache miss for key {ck} at iter {current_iter}. Check async_warmup steps.")
                handle, prev, _ = cache.get(ck)
                handle.wait()
                ks = [t.clone() for t in prev[:world]]
                vs = [t.clone() for t in prev[world:]]
                ks[rank], vs[rank] = k.clone(), v.clone()      # own shard is always fresh
                # double-buffered: this step's gather lands in the other buffer and is consumed next step
                h, nk, nv = _gather_raw_kv(k, v, group, world, ("df", mod_idx, current_iter & 1), async_op=True)
                cache.put(ck, h, nk + nv, k)

    gk = torch.cat(ks, dim=1).contiguous()
    gv = torch.cat(vs, dim=1).contiguous()
    if joint == "front":
        gk, gv = torch.cat([joint_tensor_key, gk], dim=1), torch.cat([joint_tensor_value, gv], dim=1)
    elif joint == "rear":
        gk, gv = torch.cat([gk, joint_tensor_key], dim=1), torch.cat([gv, joint_tensor_value], dim=1)
    out, lse = block_attention(q, gk, gv, dropout_p, softmax_scale, causal=causal)
    return out.to(q.dtype), lse, None
