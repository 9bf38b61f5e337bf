"""Torch-tensor front end of the C-ABI (include/cfx.h): PyTorch-ROCm tensors in, device buffers out.

PyTorch is plumbing here (device memory, streams); all arithmetic happens in libcfx.so.  Every function
raises if the tensors are not on a GPU or the library is missing - there is no CPU path in the product.
"""
from __future__ import annotations

import ctypes
from enum import IntEnum
from typing import Callable, Optional, Sequence

import torch

from . import _lib
from ._lib import CFX_MAX_BATCH, FLAG_NO_EF, FLAG_UPDATE_CACHE, CfxError, CompItem, DecompItem


class Codec(IntEnum):
    BINARY = 1   # COMPACT_COMPRESS_TYPE.BINARY (comp_rank = -1)
    INT2 = 2     # COMPACT_COMPRESS_TYPE.INT2
    INT4 = 3     # residual int4 (compress_quantize.py:522-640 on delta)
    INT8 = 4     # residual int8 (compress_quantize.py:428-484 on delta)
    TOPK = 5     # COMPACT_COMPRESS_TYPE.SPARSE, param = sparse_ratio m


_ctx = {}
_ws = {}


def _device_index(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise CfxError("compactfusion_amd codecs run on the GPU only (tensor is on %s); there is no CPU fallback" % t.device)
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def context(device: int):
    lib = _lib.load()
    c = _ctx.get(device)
    if c is None:
        c = lib.cfx_create(device)
        if not c:
            raise CfxError("cfx_create failed")
        if torch.cuda.is_available():
            rc = lib.cfx_prepare(c)          # ticket blocks of the in-launch finalize (once; keeps later calls graph-capturable)
            if rc != 0:
                raise CfxError("cfx_prepare failed: " + (lib.cfx_last_error_string(c) or b"").decode())
        _ctx[device] = c
    return c


def set_fused_finalize(on: bool, device: Optional[int] = None) -> None:
    """Compress statistics + finalize in one launch (default) or as two kernels (bit-identical; used by the tests)."""
    device = torch.cuda.current_device() if device is None else device
    _lib.load().cfx_set_fused_finalize(context(device), 1 if on else 0)


def set_rows_per_tile(rows: int, device: Optional[int] = None) -> None:
    device = torch.cuda.current_device() if device is None else device
    _lib.load().cfx_set_rows_per_tile(context(device), int(rows))


def _check(ctx, rc: int, what: str) -> None:
    if rc != 0:
        msg = _lib.load().cfx_last_error_string(ctx)
        msg = msg.decode() if msg else ""
        err = _lib.ERR_NAMES.get(rc, str(rc))
        if rc in (-2, -4, -5):
            raise ValueError(f"{what}: {err}: {msg}")
        raise CfxError(f"{what}: {err}: {msg}")


def packet_bytes(codec: int, N: int, C: int, param: int = 0) -> int:
    n = _lib.load().cfx_packet_bytes(int(codec), N, C, param)
    if n == 0:
        raise ValueError(f"invalid shape for codec {Codec(codec).name}: N={N} C={C} param={param}")
    return n


def packet_halves(codec: int, N: int, C: int, param: int = 0) -> int:
    b = packet_bytes(codec, N, C, param)
    assert b % 2 == 0
    return b // 2


def workspace(codec: int, N: int, C: int, param: int, batch: int, device: int,
              stream_handle: Optional[int] = None) -> Optional[torch.Tensor]:
    """Statistics workspace of a compress call, one per (device, stream): calls on different streams may run concurrently
    and must not share partial sums.  `stream_handle` = the stream the call will be launched on (default: the current one)."""
    need = _lib.load().cfx_workspace_bytes(int(codec), N, C, param, batch)
    if need == 0:
        return None
    sh = torch.cuda.current_stream(device).cuda_stream if stream_handle is None else stream_handle
    key = (device, sh)
    w = _ws.get(key)
    if w is None or w.numel() < need:
        w = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=f"cuda:{device}")
        _ws[key] = w
    return w


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream_handle(stream: Optional[torch.cuda.Stream], device: int) -> int:
    s = stream if stream is not None else torch.cuda.current_stream(device)
    return s.cuda_stream


def _check_nc(t: torch.Tensor, N: int, C: int, name: str) -> None:
    if t.dtype != torch.float16 or not t.is_contiguous() or t.numel() != N * C:
        raise ValueError(f"{name}: expected contiguous fp16 with {N}x{C} elements, got {t.dtype} {tuple(t.shape)}")


def compress_batch(codec: int, xs: Sequence[torch.Tensor], bases: Sequence[Optional[torch.Tensor]],
                   new_bases: Sequence[Optional[torch.Tensor]], packets: Sequence[torch.Tensor],
                   N: int, C: int, param: int = 0, update_cache: bool = True, ef: bool = True,
                   stream: Optional[torch.cuda.Stream] = None, ws: Optional[torch.Tensor] = None) -> None:
    """One launch sequence for a batch of (x, base) -> (packet, new_base).  new_base may alias base."""
    B = len(xs)
    if not (1 <= B <= CFX_MAX_BATCH):
        raise ValueError(f"batch {B} out of range 1..{CFX_MAX_BATCH}")
    dev = _device_index(xs[0])
    ctx = context(dev)
    items = (CompItem * B)()
    for i in range(B):
        _check_nc(xs[i], N, C, "x")
        if bases[i] is not None:
            _check_nc(bases[i], N, C, "base")
        if new_bases[i] is not None:
            _check_nc(new_bases[i], N, C, "new_base")
        _device_index(packets[i])
        items[i] = CompItem(_ptr(xs[i]), _ptr(bases[i]), _ptr(new_bases[i]), _ptr(packets[i]))
    flags = (FLAG_UPDATE_CACHE if update_cache else 0) | (0 if ef else FLAG_NO_EF)
    sh = _stream_handle(stream, dev)
    if ws is None:
        ws = workspace(codec, N, C, param, B, dev, sh)
    rc = _lib.load().cfx_compress_batch(ctx, int(codec), N, C, param, flags, B, items,
                                        _ptr(ws), 0 if ws is None else ws.numel(), sh)
    _check(ctx, rc, "cfx_compress_batch")


def decompress_batch(codec: int, packets: Sequence[torch.Tensor], bases: Sequence[Optional[torch.Tensor]],
                     recons: Sequence[torch.Tensor], N: int, C: int, param: int = 0,
                     stream: Optional[torch.cuda.Stream] = None) -> None:
    """recon_i = base_i + decode(packet_i) for a batch, one launch.  recon may alias base."""
    B = len(packets)
    if not (1 <= B <= CFX_MAX_BATCH):
        raise ValueError(f"batch {B} out of range 1..{CFX_MAX_BATCH}")
    dev = _device_index(recons[0])
    ctx = context(dev)
    items = (DecompItem * B)()
    for i in range(B):
        _check_nc(recons[i], N, C, "recon")
        if bases[i] is not None:
            _check_nc(bases[i], N, C, "base")
        _device_index(packets[i])
        items[i] = DecompItem(_ptr(packets[i]), _ptr(bases[i]), _ptr(recons[i]))
    rc = _lib.load().cfx_decompress_batch(ctx, int(codec), N, C, param, B, items, _stream_handle(stream, dev))
    _check(ctx, rc, "cfx_decompress_batch")


def prepare_compress(codec: int, bases: Sequence[Optional[torch.Tensor]], new_bases: Sequence[Optional[torch.Tensor]],
                     packets: Sequence[torch.Tensor], N: int, C: int, param: int = 0, update_cache: bool = True,
                     ef: bool = True) -> Callable[[Sequence[torch.Tensor]], None]:
    """`compress_batch` with the state / packet operands bound once (persistent arena and exchange buffers): the
    returned `run(xs, stream_handle=None)` only patches the activation pointers into a cached item array - the
    per-call host cost of the exchange hot loop.  The bound tensors are kept alive by the closure."""
    B = len(packets)
    if not (1 <= B <= CFX_MAX_BATCH):
        raise ValueError(f"batch {B} out of range 1..{CFX_MAX_BATCH}")
    dev = _device_index(packets[0])
    ctx = context(dev)
    items = (CompItem * B)()
    for i in range(B):
        if bases[i] is not None:
            _check_nc(bases[i], N, C, "base")
        if new_bases[i] is not None and update_cache:
            _check_nc(new_bases[i], N, C, "new_base")
        _device_index(packets[i])
        items[i] = CompItem(None, _ptr(bases[i]), _ptr(new_bases[i]) if update_cache else None, _ptr(packets[i]))
    flags = (FLAG_UPDATE_CACHE if update_cache else 0) | (0 if ef else FLAG_NO_EF)
    fn = _lib.load().cfx_compress_batch
    keep = (list(bases), list(new_bases), list(packets))
    codec = int(codec)
    ws_by_stream = {}        # the workspace belongs to the stream the call is launched on, looked up at call time

    def run(xs: Sequence[torch.Tensor], stream_handle: Optional[int] = None) -> None:
        assert len(xs) == B and keep
        for i in range(B):
            _check_nc(xs[i], N, C, "x")
            items[i].x = xs[i].data_ptr()
        sh = torch.cuda.current_stream(dev).cuda_stream if stream_handle is None else stream_handle
        w = ws_by_stream.get(sh)
        if w is None:
            ws = workspace(codec, N, C, param, B, dev, sh)
            w = ws_by_stream[sh] = (ws, _ptr(ws), 0 if ws is None else ws.numel())
        _check(ctx, fn(ctx, codec, N, C, param, flags, B, items, w[1], w[2], sh), "cfx_compress_batch")
    return run


def prepare_decompress(codec: int, packets: Sequence[torch.Tensor], bases: Sequence[Optional[torch.Tensor]],
                       recons: Sequence[torch.Tensor], N: int, C: int, param: int = 0) -> Callable[..., None]:
    """`decompress_batch` with every operand bound once; `run(stream_handle=None)` is a single C call."""
    B = len(packets)
    if not (1 <= B <= CFX_MAX_BATCH):
        raise ValueError(f"batch {B} out of range 1..{CFX_MAX_BATCH}")
    dev = _device_index(recons[0])
    ctx = context(dev)
    items = (DecompItem * B)()
    for i in range(B):
        _check_nc(recons[i], N, C, "recon")
        if bases[i] is not None:
            _check_nc(bases[i], N, C, "base")
        _device_index(packets[i])
        items[i] = DecompItem(_ptr(packets[i]), _ptr(bases[i]), _ptr(recons[i]))
    fn = _lib.load().cfx_decompress_batch
    keep = (list(packets), list(bases), list(recons))
    codec = int(codec)

    def run(stream_handle: Optional[int] = None) -> None:
        assert keep
        sh = torch.cuda.current_stream(dev).cuda_stream if stream_handle is None else stream_handle
        _check(ctx, fn(ctx, codec, N, C, param, B, items, sh), "cfx_decompress_batch")
    return run


def compress(codec: int, x: torch.Tensor, base: Optional[torch.Tensor], N: int, C: int, param: int = 0,
             update_cache: bool = True, ef: bool = True, new_base: Optional[torch.Tensor] = None,
             packet: Optional[torch.Tensor] = None):
    """Single tensor convenience.  Returns (packet fp16 1-D, new_base | None)."""
    if packet is None:
        packet = torch.empty(packet_halves(codec, N, C, param), dtype=torch.float16, device=x.device)
    if update_cache and new_base is None:
        new_base = torch.empty((N, C), dtype=torch.float16, device=x.device)
    compress_batch(codec, [x], [base], [new_base if update_cache else None], [packet], N, C, param, update_cache, ef)
    return packet, (new_base if update_cache else None)


def decompress(codec: int, packet: torch.Tensor, base: Optional[torch.Tensor], N: int, C: int, param: int = 0,
               recon: Optional[torch.Tensor] = None) -> torch.Tensor:
    if recon is None:
        recon = torch.empty((N, C), dtype=torch.float16, device=packet.device)
    if packet.data_ptr() % 16:
        packet = packet.clone()   # the kernels want a 16-byte aligned packet; a slice of a gather buffer may not be
    decompress_batch(codec, [packet], [base], [recon], N, C, param)
    return recon


def residual2_delta(x: torch.Tensor, base: torch.Tensor, delta_base: torch.Tensor, out: torch.Tensor,
                    stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
    """out = (x - base) - delta_base (second-order residual, main.py:247), elementwise fp16."""
    dev = _device_index(x)
    n = x.numel()
    for t, name in ((x, "x"), (base, "base"), (delta_base, "delta_base"), (out, "out")):
        if t.dtype != torch.float16 or not t.is_contiguous() or t.numel() != n:
            raise ValueError(f"{name}: expected contiguous fp16 with {n} elements")
    ctx = context(dev)
    rc = _lib.load().cfx_residual2_delta(ctx, _ptr(x), _ptr(base), _ptr(delta_base), _ptr(out), n, _stream_handle(stream, dev))
    _check(ctx, rc, "cfx_residual2_delta")
    return out


def residual2_update(base: torch.Tensor, delta_base: torch.Tensor, recv: torch.Tensor, new_base: torch.Tensor,
                     new_delta_base: torch.Tensor, decay: float, stream: Optional[torch.cuda.Stream] = None) -> None:
    """new_base = (base + delta_base) + recv ; new_delta_base = (delta_base + recv) * decay (main.py:250-256, 272-273).
    new_base / new_delta_base may be base / delta_base themselves (in-place state update)."""
    dev = _device_index(base)
    n = base.numel()
    for t, name in ((base, "base"), (delta_base, "delta_base"), (recv, "recv"), (new_base, "new_base"), (new_delta_base, "new_delta_base")):
        if t.dtype != torch.float16 or not t.is_contiguous() or t.numel() != n:
            raise ValueError(f"{name}: expected contiguous fp16 with {n} elements")
    ctx = context(dev)
    rc = _lib.load().cfx_residual2_update(ctx, _ptr(base), _ptr(delta_base), _ptr(recv), _ptr(new_base), _ptr(new_delta_base),
                                          float(decay), n, _stream_handle(stream, dev))
    _check(ctx, rc, "cfx_residual2_update")


def copy_probe(dst: torch.Tensor, src: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> None:
    dev = _device_index(dst)
    ctx = context(dev)
    rc = _lib.load().cfx_copy_probe(ctx, dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size(), _stream_handle(stream, dev))
    _check(ctx, rc, "cfx_copy_probe")


# ---- low-rank family (cfx_lowrank.hip) -----------------------------------------------------------------------------------
_lr_ws = {}


def lr_rank_pad(rank: int) -> int:
    return 8 if rank <= 8 else (16 if rank <= 16 else 32)


def lr_packet_halves(quantized: bool, N: int, C: int, rank: int) -> int:
    n = _lib.load().cfx_lr_packet_bytes(int(quantized), N, C, rank)
    if n == 0:
        raise ValueError(f"invalid shape for the low-rank codec: N={N} C={C} rank={rank} quantized={quantized} "
                         "(rank even and <= 32; LOW_RANK_Q: N, C even and rank % 8 == 0)")
    return n // 2


def _lr_workspace(quantized: bool, N: int, C: int, rank: int, batch: int, device: int) -> torch.Tensor:
    need = _lib.load().cfx_lr_workspace_bytes(int(quantized), N, C, rank, batch)
    if need == 0:
        raise ValueError("invalid shape for the low-rank codec")
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    w = _lr_ws.get(key)
    if w is None or w.numel() < need:
        w = torch.empty(need, dtype=torch.uint8, device=f"cuda:{device}")
        _lr_ws[key] = w
    return w


def lr_compress_batch(quantized: bool, xs, bases, new_bases, packets, init_qs, N: int, C: int, rank: int,
                      update_cache: bool = True, ef: bool = True, stream: Optional[torch.cuda.Stream] = None) -> None:
    """Low-rank residual compress of a batch: packet_i, new_base_i from (x_i, base_i) and the start matrix init_q_i
    (C x lr_rank_pad(rank) fp32, columns >= rank zero)."""
    B = len(xs)
    if not (1 <= B <= CFX_MAX_BATCH):
        raise ValueError(f"batch {B} out of range 1..{CFX_MAX_BATCH}")
    dev = _device_index(xs[0])
    ctx = context(dev)
    items = (CompItem * B)()
    qptr = (ctypes.c_void_p * B)()
    rp = lr_rank_pad(rank)
    for i in range(B):
        _check_nc(xs[i], N, C, "x")
        q = init_qs[i]
        if q.dtype != torch.float32 or tuple(q.shape) != (C, rp) or not q.is_contiguous():
            raise ValueError(f"init_q must be a contiguous fp32 ({C}, {rp}) tensor")
        _device_index(q)
        items[i] = CompItem(_ptr(xs[i]), _ptr(bases[i]), _ptr(new_bases[i]) if update_cache else None, _ptr(packets[i]))
        qptr[i] = q.data_ptr()
    ws = _lr_workspace(quantized, N, C, rank, B, dev)
    flags = (FLAG_UPDATE_CACHE if update_cache else 0) | (0 if ef else FLAG_NO_EF)
    rc = _lib.load().cfx_lr_compress_batch(ctx, int(quantized), N, C, rank, flags, B, items, qptr, ws.data_ptr(), ws.numel(),
                                           _stream_handle(stream, dev))
    _check(ctx, rc, "cfx_lr_compress_batch")


def lr_decompress_batch(quantized: bool, packets, bases, recons, N: int, C: int, rank: int,
                        stream: Optional[torch.cuda.Stream] = None) -> None:
    B = len(packets)
    if not (1 <= B <= CFX_MAX_BATCH):
        raise ValueError(f"batch {B} out of range 1..{CFX_MAX_BATCH}")
    dev = _device_index(recons[0])
    ctx = context(dev)
    items = (DecompItem * B)()
    for i in range(B):
        _check_nc(recons[i], N, C, "recon")
        _device_index(packets[i])
        items[i] = DecompItem(_ptr(packets[i]), _ptr(bases[i]), _ptr(recons[i]))
    ws = _lr_workspace(quantized, N, C, rank, B, dev) if quantized else None
    rc = _lib.load().cfx_lr_decompress_batch(ctx, int(quantized), N, C, rank, B, items, _ptr(ws), 0 if ws is None else ws.numel(),
                                             _stream_handle(stream, dev))
    _check(ctx, rc, "cfx_lr_decompress_batch")


# ---- 1-bit codec with rank-K scales (deprecated in the reference; main.py:188-189) ----------------------------------------
def binary_rank_packet_halves(N: int, C: int, rank: int) -> int:
    n = _lib.load().cfx_binary_rank_packet_bytes(N, C, rank)
    if n == 0:
        raise ValueError(f"invalid shape / rank for the rank-K 1-bit codec: ({N}, {C}), rank {rank} (1..32)")
    return n // 2


def binary_rank_compress_batch(xs, bases, new_bases, packets, init_qs, N: int, C: int, rank: int, update_cache: bool = True,
                               ef: bool = True, stream: Optional[torch.cuda.Stream] = None) -> None:
    """bits + rank-K scale factors of |x - base| (+ error-feedback state) for a batch; init_q_i: (C, lr_rank_pad(rank)) fp32, columns >= rank zero."""
    B = len(xs)
    if not (1 <= B <= CFX_MAX_BATCH):
        raise ValueError(f"batch {B} out of range 1..{CFX_MAX_BATCH}")
    dev = _device_index(xs[0])
    ctx = context(dev)
    lib = _lib.load()
    items = (CompItem * B)()
    qptr = (ctypes.c_void_p * B)()
    for i in range(B):
        _check_nc(xs[i], N, C, "x")
        q = init_qs[i]
        if q.dtype != torch.float32 or tuple(q.shape) != (C, lr_rank_pad(rank)) or not q.is_contiguous():
            raise ValueError(f"init_q must be a contiguous fp32 ({C}, {lr_rank_pad(rank)}) tensor")
        items[i] = CompItem(_ptr(xs[i]), _ptr(bases[i]), _ptr(new_bases[i]) if update_cache else None, _ptr(packets[i]))
        qptr[i] = q.data_ptr()
    need = lib.cfx_binary_rank_workspace_bytes(N, C, rank, B)
    if need == 0:
        raise ValueError("invalid shape / rank for the rank-K 1-bit codec")
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    w = _lr_ws.get(key)
    if w is None or w.numel() < need:
        w = _lr_ws[key] = torch.empty(need, dtype=torch.uint8, device=f"cuda:{dev}")
    flags = (FLAG_UPDATE_CACHE if update_cache else 0) | (0 if ef else FLAG_NO_EF)
    rc = lib.cfx_binary_rank_compress_batch(ctx, N, C, rank, flags, B, items, qptr, w.data_ptr(), w.numel(), _stream_handle(stream, dev))
    _check(ctx, rc, "cfx_binary_rank_compress_batch")


def binary_rank_decompress_batch(packets, bases, recons, N: int, C: int, rank: int, stream: Optional[torch.cuda.Stream] = None) -> None:
    B = len(packets)
    if not (1 <= B <= CFX_MAX_BATCH):
        raise ValueError(f"batch {B} out of range 1..{CFX_MAX_BATCH}")
    dev = _device_index(recons[0])
    ctx = context(dev)
    items = (DecompItem * B)()
    for i in range(B):
        _check_nc(recons[i], N, C, "recon")
        _device_index(packets[i])
        items[i] = DecompItem(_ptr(packets[i]), _ptr(bases[i]), _ptr(recons[i]))
    rc = _lib.load().cfx_binary_rank_decompress_batch(ctx, N, C, rank, B, items, _stream_handle(stream, dev))
    _check(ctx, rc, "cfx_binary_rank_decompress_batch")
