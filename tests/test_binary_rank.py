"""The 1-bit codec with rank-K scales (a4 / a7, rank >= 1: xfuser/compact/fastpath.py:88-120, 186-200; compress_quantize.py:37-49 -
deprecated in the reference, main.py:188-189, but its own test parametrises rank in {-1, 1, 4}, tests/compact/compress_fastpath_test.py:45-101).
Golden group G1b (tests/golden/make_golden_rank.py) holds the reference's packed bits, U, V, new_base for ranks 1 and 4 and the start
matrix its subspace iteration drew.  Parity: sign bits bit-exact; the scale matrix U V^T within 3e-3 (the low-rank tolerance: the result
depends on GEMM / QR rounding order); the new state within 1e-3 of the reference's; decoding the REFERENCE's packet with the receiver
kernel reproduces the reference's state (an ulp on few entries: the reference adds the K fp16 products in fp16, tree order unspecified).
Group G1c (`make_golden_rank.py wide`): ranks 16 and 32 - the reference's Triton kernels take any power of two (fastpath.py:91), this repo
up to the factor chain's 32 (until round 6: 8)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_np as R

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "g1b_binary_rank.npz")
GOLD_WIDE = os.path.join(HERE, "golden", "g1c_binary_rank_wide.npz")
WIDE = [(64, 256, 42, 16), (64, 256, 42, 32), (128, 1152, 43, 16)]
CASES = [(N, C, seed, r) for (N, C) in ((64, 256), (256, 1152)) for seed in (42, 43) for r in (1, 4)] + WIDE
F16 = np.float16


def _inputs(seed, N, C):
    torch.manual_seed(seed)
    x = torch.randn((N, C), dtype=torch.half).contiguous()
    base = (torch.randn_like(x) * 0.1).contiguous()
    return x, base


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _gold(tag, name):
    g = np.load(GOLD_WIDE if int(tag.rsplit("/r", 1)[1]) > 8 else GOLD)
    a = g[f"{tag}/{name}"]
    return a.view(F16) if a.dtype == np.uint16 else a


@pytest.mark.parametrize("N,C,seed,r", CASES)
def test_oracle_vs_reference(N, C, seed, r):
    tag = f"{N}x{C}_s{seed}/r{r}"
    x, base = _inputs(seed, N, C)
    packed, U, VT, nb = R.binary_rank_quant_fastpath(x.numpy(), base.numpy(), r, _gold(tag, "q0"))
    assert np.array_equal(packed, _gold(tag, "packed")), "sign bits"
    gu, gv = _gold(tag, "u").reshape(N, r), _gold(tag, "v").reshape(C, r)
    assert _rel(U.astype(np.float32) @ VT.astype(np.float32).T, gu.astype(np.float32) @ gv.astype(np.float32).T) < 3e-3
    gnb = _gold(tag, "new_base").reshape(N, C)
    assert _rel(nb, gnb) < 1e-3
    # the receiver arithmetic on the reference's own factors: the reference's state up to the order its fp16 sum takes
    mine = R.binary_rank_apply(base.numpy(), R.unpack_bits_1(_gold(tag, "packed")), gu, gv)
    d = np.abs(mine.astype(np.float32) - gnb.astype(np.float32))
    assert d.max() <= 2e-3 and (d > 0).mean() <= (0.0 if r == 1 else (0.05 if r <= 4 else 0.15)), (d.max(), (d > 0).mean())


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,seed,r", CASES)
def test_hip_vs_reference_and_oracle(N, C, seed, r):
    from compactfusion_amd.compact import fastpath as FP, lowrank as LR
    tag = f"{N}x{C}_s{seed}/r{r}"
    x, base = _inputs(seed, N, C)
    q0 = torch.from_numpy(_gold(tag, "q0"))
    LR.set_init_q(q0)
    try:
        packed, u, v, nb = FP.binary_quant_fastpath(x.cuda(), base.cuda(), r, True)
    finally:
        LR.set_init_q(None)
    assert tuple(packed.shape) == (N, C // 8) and tuple(u.shape) == (N, r) and tuple(v.shape) == (C, r)
    assert np.array_equal(packed.cpu().numpy(), _gold(tag, "packed")), "sign bits"
    gu, gv = _gold(tag, "u").reshape(N, r), _gold(tag, "v").reshape(C, r)
    sc = u.float().cpu().numpy() @ v.float().cpu().numpy().T
    assert _rel(sc, gu.astype(np.float32) @ gv.astype(np.float32).T) < 3e-3
    gnb = _gold(tag, "new_base").reshape(N, C)
    assert _rel(nb.float().cpu().numpy(), gnb) < 1e-3
    # HIP sender state == the oracle's arithmetic on the HIP factors, bit for bit; receiver == sender, bit for bit
    want = R.binary_rank_apply(base.numpy(), R.unpack_bits_1(packed.cpu().numpy()), u.cpu().numpy(), v.cpu().numpy())
    assert np.array_equal(nb.cpu().numpy().view(np.uint16), want.view(np.uint16))
    rec = FP.binary_dequant_fastpath(packed, u, v, base.cuda())
    assert torch.equal(rec.view(torch.int16), nb.view(torch.int16))
    # the receiver kernel on the REFERENCE's packet
    rec_ref = FP.binary_dequant_fastpath(torch.from_numpy(_gold(tag, "packed")).cuda(), torch.from_numpy(gu.copy()).cuda(),
                                         torch.from_numpy(gv.copy()).cuda(), base.cuda())
    d = (rec_ref.float().cpu() - torch.from_numpy(gnb.astype(np.float32))).abs()
    assert float(d.max()) <= 2e-3 and float((d > 0).float().mean()) <= (0.0 if r == 1 else (0.05 if r <= 4 else 0.15))


@pytest.mark.gpu
@pytest.mark.parametrize("rank", [4, 16])
def test_state_machine_with_rank_scales(monkeypatch, tmp_path, rank):
    """compact_compress / compact_decompress with BINARY, comp_rank = 4 / 16 (COMPACT_ALLOW_DEPRECATED): sender and receiver states stay
    bit-identical over error-feedback steps, and the slow-path wire [q | U (N,K) | V (K,C)] round-trips."""
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.collector import collector
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    from compactfusion_amd.compact.slowpath import slowpath_compress, slowpath_decompress
    collector.init(collector.Collector(str(tmp_path), enabled=False))
    monkeypatch.setattr(cm, "ALLOW_DEPRECATED", True)
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: None, residual=1, ef=True, fastpath=True, comp_rank=rank))
    g = torch.Generator().manual_seed(3)
    cur = torch.randn(1, 64, 8, 64, generator=g).half()
    for t in range(4):
        typ = T.WARMUP if t == 0 else T.BINARY
        pkt = cm.compact_compress("0-0-k", cur.cuda(), typ, update_cache=True)
        if t:
            assert pkt.numel() == 64 * 512 // 16 + (64 + 512) * rank
        rec = cm.compact_decompress("0-1-k", pkt.clone(), typ, (1, 64, 8, 64), update_cache=True)
        assert torch.equal(cm.compact_cache().get_base("0-0-k"), cm.compact_cache().get_base("0-1-k")), f"step {t}: sender / receiver diverged"
        err = float((rec.float().cpu() - cur.float()).norm() / cur.float().norm())
        assert err < (1e-6 if t == 0 else 0.12)
        cur = (cur.float() + 0.1 * torch.randn(1, 64, 8, 64, generator=g)).half()
    x = torch.randn(128, 256, generator=g).half().cuda()
    p = slowpath_compress(x, T.BINARY, rank=rank)
    assert p.numel() == 128 * 256 // 16 + (128 + 256) * rank
    d = slowpath_decompress(p, (128, 256), T.BINARY, rank=rank)
    assert float((d.float() - x.float()).norm() / x.float().norm()) < 0.75 and torch.isfinite(d.float()).all()
