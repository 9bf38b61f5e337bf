"""Block attention + running log-sum-exp merge used by the ring / gather consumers.

The reference takes these from un-vendored third-party packages (`yunchang.ring.utils.update_out_and_lse`,
`yunchang.kernels.attention.pytorch_attn_forward`, `flash_attn._flash_attn_forward`; call sites ring.py:225-263).
They are restated here from the published ring-attention algorithm:
    out <- out - sigmoid(lse_b - lse) * (out - out_b) ;  lse <- lse - logsigmoid(lse - lse_b)
The attention math itself is not part of the compressed-exchange hot path; on the GPU it is PyTorch-ROCm's fused
SDPA kernel, elsewhere an explicit fp32 softmax."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F


def block_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dropout_p: float = 0.0,
                    softmax_scale: Optional[float] = None, causal: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """q (B, Sq, H, D), k/v (B, Sk, H, D) -> out (B, Sq, H, D) in q.dtype, lse (B, H, Sq) fp32."""
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    qt, kt, vt = (t.transpose(1, 2) for t in (q, k, v))
    if q.is_cuda and dropout_p == 0.0:
        try:
            res = torch.ops.aten._scaled_dot_product_flash_attention(qt, kt, vt, 0.0, causal, False, scale=softmax_scale)
            return res[0].transpose(1, 2), res[1].float()
        except (RuntimeError, NotImplementedError):
            pass
    s = torch.matmul(qt.float(), kt.float().transpose(-1, -2)) * softmax_scale
    if causal:
        Sq, Sk = s.shape[-2], s.shape[-1]
        mask = torch.ones(Sq, Sk, dtype=torch.bool, device=s.device).tril(diagonal=Sk - Sq)
        s = s.masked_fill(~mask, float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse.unsqueeze(-1))
    if dropout_p > 0.0:
        p = F.dropout(p, dropout_p)
    out = torch.matmul(p, vt.float()).to(q.dtype)
    return out.transpose(1, 2), lse


def update_out_and_lse(out: Optional[torch.Tensor], lse: Optional[torch.Tensor], block_out: torch.Tensor,
                       block_lse: torch.Tensor, wait=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge one attention block into the running (out fp32 (B,S,H,D), lse fp32 (B,S,H,1)).  On the GPU this is ONE native
    launch (`cfx_attn_merge`) that reads the SDPA kernel's own (B,H,S,D) / (B,H,S) outputs and updates out / lse in place;
    elsewhere (CPU tests of the host logic) the same formula in eager torch.
    `wait` = (flag device address, epoch): the exchange lane's "next peer reconstructed" flag - the merge launch also waits for
    it (`cfx_attn_merge_wait`), so the attention block that follows finds the peer's K,V complete without a stream event."""
    if block_out.is_cuda and block_out.dtype == torch.float16 and block_lse.dtype == torch.float32 and block_out.shape[-1] % 8 == 0 \
            and block_out.shape[-1] <= 512 and (out is None or (out.is_contiguous() and lse.is_contiguous() and out.dtype == torch.float32)):
        if block_lse.is_contiguous() and block_out.data_ptr() % 16 == 0:
            if block_out.is_contiguous():                  # the fused SDPA output, (B,S,H,D) contiguous underneath
                return _merge_native(out, lse, block_out, block_lse, 1, wait)
            if block_out.transpose(1, 2).is_contiguous():  # ... or (B,H,S,D)
                return _merge_native(out, lse, block_out, block_lse, 0, wait)
    block_out = block_out.to(torch.float32)
    block_lse = block_lse.transpose(-2, -1).unsqueeze(-1)
    if out is not None:
        out = out - torch.sigmoid(block_lse - lse) * (out - block_out)
        lse = lse - F.logsigmoid(lse - block_lse)
    else:
        out, lse = block_out, block_lse
    if wait is not None:
        flag_wait(wait, block_out.device)
    return out, lse


def flag_wait(wait, device) -> None:
    """A stand-alone wait launch on the current stream for (flag address, epoch) - what `update_out_and_lse(wait=...)` folds into
    its merge launch when it can."""
    from .. import _lib, codecs
    dev = device.index if device.index is not None else torch.cuda.current_device()
    ctx = codecs.context(dev)
    rc = _lib.load().cfx_flag_wait(ctx, wait[0], wait[1], torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        codecs._check(ctx, rc, "cfx_flag_wait")


_merge_fn = None


def _merge_native(out, lse, bo, bl, bshd, wait=None):
    global _merge_fn
    from .. import _lib, codecs
    B, S, H, D = bo.shape
    dev = bo.device.index if bo.device.index is not None else torch.cuda.current_device()
    first = out is None
    if first:
        out = torch.empty((B, S, H, D), dtype=torch.float32, device=bo.device)
        lse = torch.empty((B, S, H, 1), dtype=torch.float32, device=bo.device)
    if _merge_fn is None:
        _merge_fn = _lib.load().cfx_attn_merge_wait
    ctx = codecs.context(dev)
    rc = _merge_fn(ctx, out.data_ptr(), lse.data_ptr(), bo.data_ptr(), bl.data_ptr(), B, S, H, D, bshd, 1 if first else 0,
                   wait[0] if wait is not None else None, wait[1] if wait is not None else 0,
                   torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        codecs._check(ctx, rc, "cfx_attn_merge")
    return out, lse
