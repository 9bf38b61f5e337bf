"""TEST-ONLY stand-in for the HIP kernels: implements `codecs.compress_batch` / `codecs.decompress_batch` (and their prepared forms) on CPU
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


# ---- low-rank family: the reference algorithm (compress_lowrank.py:14-61, slowpath.py:54-75,151-164) in torch on CPU ----
def _lr_sections(quantized, N, C, r):
    if not quantized:
        return (0, N * r), (N * r, C * r)
    nu = N * r // 4 + 2 * r
    return (0, nu), (nu, C * r // 4 + 2 * r)


def _q4(m):
    rows, r = m.shape
    pkt, _ = R.compress("int4", _np16(m).reshape(rows, r), None)
    return torch.from_numpy(pkt.view(np.int16).copy()).view(torch.float16)


def _dq4(sec, rows, r):
    rec = R.decompress("int4", _np16(sec).reshape(-1).copy(), rows, r)
    return torch.from_numpy(R.bits(rec).view(np.int16).copy()).view(torch.float16)


def _lr_decode(quantized, pkt, N, C, r):
    (ou, nu), (ov, nv) = _lr_sections(quantized, N, C, r)
    if not quantized:
        return torch.matmul(pkt[ou:ou + nu].view(N, r).float(), pkt[ov:ov + nv].view(r, C).float()).half()
    return torch.matmul(_dq4(pkt[ou:ou + nu], N, r).float(), _dq4(pkt[ov:ov + nv], C, r).float().t()).half()


def lr_compress_batch(quantized, xs, bases, new_bases, packets, init_qs, N, C, rank, update_cache=True, ef=True, stream=None):
    for x, b, nb, p, q0 in zip(xs, bases, new_bases, packets, init_qs):
        d = x if b is None else x - b
        Af = d.float()
        Q = q0[:, :rank].float()
        for _ in range(2):
            Q, _ = torch.linalg.qr(Af.t() @ (Af @ Q))
        U, _ = torch.linalg.qr(Af @ Q)
        V = U.t() @ Af
        U, V = U.half(), V.half()
        (ou, nu), (ov, nv) = _lr_sections(quantized, N, C, rank)
        if not quantized:
            p[ou:ou + nu] = U.reshape(-1)
            p[ov:ov + nv] = V.reshape(-1)
        else:
            p[ou:ou + nu] = _q4(U)
            p[ov:ov + nv] = _q4(V.t().contiguous())
        if update_cache and nb is not None:
            recv = _lr_decode(quantized, p, N, C, rank)
            nb.copy_(x if not ef else (recv if b is None else b + recv))


def lr_decompress_batch(quantized, packets, bases, recons, N, C, rank, stream=None):
    for p, b, r in zip(packets, bases, recons):
        recv = _lr_decode(quantized, p, N, C, rank)
        r.copy_(recv if b is None else b + recv)


def prepare_compress(codec, bases, new_bases, packets, N, C, param=0, update_cache=True, ef=True):
    def run(xs, stream_handle=None):
        compress_batch(codec, xs, bases, new_bases, packets, N, C, param, update_cache, ef)
    return run


def prepare_decompress(codec, packets, bases, recons, N, C, param=0):
    def run(stream_handle=None):
        decompress_batch(codec, packets, bases, recons, N, C, param)
    return run


def residual2_delta(x, base, delta_base, out, stream=None):
    out.view(torch.int16).numpy().view(np.uint16).reshape(-1)[:] = R.bits(R.residual2_delta(
        _np16(x).view(np.float16).reshape(-1), _np16(base).view(np.float16).reshape(-1), _np16(delta_base).view(np.float16).reshape(-1)))
    return out


def residual2_update(base, delta_base, recv, new_base, new_delta_base, decay, stream=None):
    nb, nd = R.residual2_update(_np16(base).view(np.float16).reshape(-1).copy(), _np16(delta_base).view(np.float16).reshape(-1).copy(),
                                _np16(recv).view(np.float16).reshape(-1).copy(), decay)
    new_base.view(torch.int16).numpy().view(np.uint16).reshape(-1)[:] = R.bits(nb)
    new_delta_base.view(torch.int16).numpy().view(np.uint16).reshape(-1)[:] = R.bits(nd)


def install(monkeypatch):
    from compactfusion_amd import codecs
    monkeypatch.setattr(codecs, "compress_batch", compress_batch)
    monkeypatch.setattr(codecs, "decompress_batch", decompress_batch)
    monkeypatch.setattr(codecs, "residual2_delta", residual2_delta)
    monkeypatch.setattr(codecs, "residual2_update", residual2_update)
    monkeypatch.setattr(codecs, "prepare_compress", prepare_compress)
    monkeypatch.setattr(codecs, "prepare_decompress", prepare_decompress)
    monkeypatch.setattr(codecs, "lr_compress_batch", lr_compress_batch)
    monkeypatch.setattr(codecs, "lr_decompress_batch", lr_decompress_batch)


def install_plain():
    """For spawned worker processes (no pytest monkeypatch there)."""
    from compactfusion_amd import codecs
    codecs.compress_batch = compress_batch
    codecs.decompress_batch = decompress_batch
    codecs.residual2_delta = residual2_delta
    codecs.residual2_update = residual2_update
    codecs.prepare_compress = prepare_compress
    codecs.prepare_decompress = prepare_decompress
    codecs.lr_compress_batch = lr_compress_batch
    codecs.lr_decompress_batch = lr_decompress_batch
