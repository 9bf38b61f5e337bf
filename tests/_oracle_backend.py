"""TEST-ONLY stand-in for the HIP kernels: implements `codecs.compress_batch` / `codecs.decompress_batch` on CPU
tensors with the oracle, so the HOST logic (state machine, wire sizes, cache, ring / gather schedules over gloo) can
be exercised in this GPU-less container.  It is installed by monkeypatching inside tests and never shipped: the
product (`compactfusion_amd.codecs`) has no CPU path and refuses CPU tensors."""
import numpy as np
import torch

from oracle import ref_np as R

NAMES = {1: "binary", 2: "int2", 3: "int4", 4: "int8", 5: "topk"}


def _np16(t):
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16)


def compress_batch(codec, xs, bases, new_bases, packets, N, C, param=0, update_cache=True, ef=True, stream=None, ws=None):
    name = NAMES[int(codec)]
    for x, b, nb, p in zip(xs, bases, new_bases, packets):
        xb = _np16(x).reshape(N, C)
        bb = None if b is None else _np16(b).reshape(N, C).copy()
        pkt, newb = R.residual_compress(name, xb, bb, param, ef)
        pv = p.view(torch.int16).numpy().view(np.uint16).reshape(-1)
        pv[:pkt.size] = pkt
        if update_cache and nb is not None:
            nb.view(torch.int16).numpy().view(np.uint16).reshape(N, C)[:] = R.bits(newb)


def decompress_batch(codec, packets, bases, recons, N, C, param=0, stream=None):
    name = NAMES[int(codec)]
    n_half = R.packet_halves(name, N, C, param)
    for p, b, r in zip(packets, bases, recons):
        pw = _np16(p).reshape(-1)[:n_half].copy()
        bb = None if b is None else _np16(b).reshape(N, C).copy()
        rec = R.residual_decompress(name, pw, bb, N, C, param)
        r.view(torch.int16).numpy().view(np.uint16).reshape(N, C)[:] = R.bits(rec)


def install(monkeypatch):
    from compactfusion_amd import codecs
    monkeypatch.setattr(codecs, "compress_batch", compress_batch)
    monkeypatch.setattr(codecs, "decompress_batch", decompress_batch)


def install_plain():
    """For spawned worker processes (no pytest monkeypatch there)."""
    from compactfusion_amd import codecs
    codecs.compress_batch = compress_batch
    codecs.decompress_batch = decompress_batch
