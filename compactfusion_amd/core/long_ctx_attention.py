"""`xFuserLongContextAttention` - the attention layer through which the compressed exchange is reached.

Mirror of `xfuser/core/long_ctx_attention/hybrid/attn_layer.py:19-243` (constructor keywords, `forward` signature,
joint-tensor handling, global layer index, the `compact_config().enabled` switch read ONCE at construction, the
`mod_idx` / `current_iter` plumbing, Q/K/V collection), with the pieces the reference inherits from yunchang restated:
the Ulysses all-to-all (`SeqAllToAll4D`, never compressed - collective C7 of SURVEY.md §2.3) and the uncompressed ring
attention used when compaction is disabled (`xdit_ring_flash_attn_func`, ring_flash_attn.py:16-137).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor

from ..compact.attention import block_attention, update_out_and_lse
from ..prof import Profiler
from .distributed import get_sp_group

ATTN_LAYER_IDX = 0      # global layer counter (attn_layer.py:18): the cache key prefix of the compressed exchange


def reset_layer_index() -> None:
    global ATTN_LAYER_IDX
    ATTN_LAYER_IDX = 0


def _all_to_all(chunks, group):
    """out[i] = chunk sent to me by rank i.  gloo has no all_to_all: fall back to W gathers there (tests only)."""
    world = dist.get_world_size(group)
    # contiguous receive buffers (empty_like would inherit the permuted strides of a transposed attention output)
    out = [torch.empty(chunks[0].shape, dtype=chunks[0].dtype, device=chunks[0].device) for _ in range(world)]
    if dist.get_backend(group) == "gloo":
        me = dist.get_rank(group)
        for src in range(world):
            # rank src scatters its chunks
            dist.scatter(out[src], scatter_list=[c.contiguous() for c in chunks] if me == src else None,
                         src=dist.get_global_rank(group, src) if group is not None else src, group=group)
        return out
    dist.all_to_all(out, [c.contiguous() for c in chunks], group=group)
    return out


def seq_all_to_all_4d(x: Tensor, scatter_idx: int, gather_idx: int, group) -> Tensor:
    """Ulysses exchange on (bs, seq, heads, dim): split dimension `scatter_idx` across the group and concatenate the
    received pieces along `gather_idx` (2,1: heads -> sequence before attention; 1,2: back afterwards)."""
    if group is None or dist.get_world_size(group) == 1:
        return x
    world = dist.get_world_size(group)
    assert x.shape[scatter_idx] % world == 0, "the scattered dimension must divide by the Ulysses degree"
    pieces = _all_to_all(list(torch.chunk(x, world, dim=scatter_idx)), group)
    return torch.cat(pieces, dim=gather_idx).contiguous()


def ring_attention_fwd(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, group=None,
                       joint_tensor_key=None, joint_tensor_value=None, joint_strategy="none"):
    """Uncompressed ring attention: raw K/V hop around the ring (the baseline the compressed path is compared to)."""
    from ..compact.ring import RingComm, _joint_mode, _with_joint
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    jmode = _joint_mode(joint_tensor_key, joint_tensor_value, joint_strategy)
    comm = RingComm(group)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    out = lse = None
    for step in range(comm.world_size):
        if step + 1 != comm.world_size:
            nk, nv = comm.send_recv(k), comm.send_recv(v)
            comm.commit()
        if not causal or step <= comm.rank:
            kk, vv = _with_joint(k, v, joint_tensor_key, joint_tensor_value, jmode, step, comm.world_size)
            bo, bl = block_attention(q, kk, vv, dropout_p, softmax_scale, causal=causal and step == 0)
            out, lse = update_out_and_lse(out, lse, bo, bl)
        if step + 1 != comm.world_size:
            comm.wait()
            k, v = nk, nv
    return out.to(q.dtype), lse.squeeze(dim=-1).transpose(1, 2), None


class xFuserLongContextAttention(torch.nn.Module):
    ring_impl_type_supported_kv_cache = ["basic"]

    def __init__(self, scatter_idx: int = 2, gather_idx: int = 1, ring_impl_type: str = "basic", use_pack_qkv: bool = False,
                 use_kv_cache: bool = False, attn_type=None) -> None:
        super().__init__()
        sp = get_sp_group()
        self.ring_pg, self.ulysses_pg = sp.ring_group, sp.ulysses_group
        self.scatter_idx, self.gather_idx = scatter_idx, gather_idx
        self.use_pack_qkv, self.use_kv_cache = use_pack_qkv, use_kv_cache
        if use_kv_cache and ring_impl_type not in self.ring_impl_type_supported_kv_cache:
            raise RuntimeError(f"ring_impl_type: {ring_impl_type} do not support SP kv cache.")
        from ..compact.main import compact_config
        from ..compact.ring import compact_fwd
        cfg = compact_config()
        # bound ONCE, like the reference (attn_layer.py:59-64): compact_init must precede model construction
        self.compact_enabled = bool(cfg is not None and cfg.enabled)
        self.ring_attn_fn = compact_fwd if self.compact_enabled else ring_attention_fwd
        self.idx: Optional[int] = None

    @torch.compiler.disable
    def forward(self, attn, query: Tensor, key: Tensor, value: Tensor, *, joint_tensor_query=None, joint_tensor_key=None,
                joint_tensor_value=None, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                alibi_slopes=None, deterministic=False, return_attn_probs=False, joint_strategy="none") -> Tensor:
        joint = [joint_tensor_query, joint_tensor_key, joint_tensor_value]
        if any(t is not None for t in joint) and not all(t is not None for t in joint):
            raise ValueError("joint_tensor_query, joint_tensor_key, and joint_tensor_value should be None or not None simultaneously.")
        is_joint = joint_tensor_query is not None
        if is_joint:
            if joint_strategy not in ("front", "rear"):
                raise ValueError(f"joint_strategy: {joint_strategy} not supprted. supported joint strategy: ['front', 'rear']")
            query = torch.cat([query, joint_tensor_query], dim=1) if joint_strategy == "rear" else torch.cat([joint_tensor_query, query], dim=1)
            uw, ur = dist.get_world_size(self.ulysses_pg), dist.get_rank(self.ulysses_pg)
            per = joint_tensor_key.shape[-2] // uw                     # each Ulysses rank keeps its heads of the joint K/V
            joint_tensor_key = joint_tensor_key[..., per * ur:per * (ur + 1), :]
            joint_tensor_value = joint_tensor_value[..., per * ur:per * (ur + 1), :]
        with Profiler.scope("ulysses.all2all"):
            q = seq_all_to_all_4d(query, self.scatter_idx, self.gather_idx, self.ulysses_pg)
            k = seq_all_to_all_4d(key, self.scatter_idx, self.gather_idx, self.ulysses_pg)
            v = seq_all_to_all_4d(value, self.scatter_idx, self.gather_idx, self.ulysses_pg)
        if self.idx is None:
            global ATTN_LAYER_IDX
            self.idx = ATTN_LAYER_IDX
            ATTN_LAYER_IDX += 1
        from ..collector.collector import collect
        from ..compact.main import compact_get_step
        step = compact_get_step()
        collect(q, "q", step, self.idx)
        collect(k, "k", step, self.idx)
        collect(v, "v", step, self.idx)
        common = dict(dropout_p=dropout_p, softmax_scale=softmax_scale, causal=causal, group=self.ring_pg,
                      joint_tensor_key=joint_tensor_key, joint_tensor_value=joint_tensor_value, joint_strategy=joint_strategy)
        if self.compact_enabled:
            out = self.ring_attn_fn(q, k, v, window_size=window_size, alibi_slopes=alibi_slopes, deterministic=deterministic,
                                    return_attn_probs=return_attn_probs, attn_layer=attn if self.use_kv_cache else None,
                                    mod_idx=self.idx, current_iter=step, **common)
        else:
            out = self.ring_attn_fn(q, k, v, **common)
        context = out[0] if isinstance(out, tuple) else out
        with Profiler.scope("ulysses.all2all"):
            return seq_all_to_all_4d(context, self.gather_idx, self.scatter_idx, self.ulysses_pg)
