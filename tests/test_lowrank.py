"""Low-rank family: subspace iteration and the LOW_RANK / LOW_RANK_Q wire codecs.
CPU part: torch ops on CPU + the oracle stand-in for the int4 factor kernel; GPU part (-m gpu): everything on device."""
import numpy as np
import pytest
import torch

import _golden as G
import _oracle_backend as OB

FN = "g8_lowrank_eager.npz"


def t16(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.float16)


def rel(a, b):
    return float(torch.norm(a.float() - b.float()) / torch.norm(b.float()))


@pytest.fixture(autouse=True)
def _collector(tmp_path):
    from compactfusion_amd.collector import collector
    collector.init(collector.Collector(str(tmp_path), enabled=False))
    yield


def _delta(tag, N, C, dev="cpu"):
    x, b = G.inputs(FN, tag, 42, N, C)
    return (t16(x) - t16(b)).contiguous().to(dev)


def _check_subspace(dev):
    from compactfusion_amd.compact.compress_lowrank import subspace_iter
    for (N, C) in [(64, 256), (256, 1152)]:
        tag = f"{N}x{C}_s42"
        d = _delta(tag, N, C, dev)
        for r in (8, 32):
            q0 = torch.from_numpy(G.get(FN, f"{tag}/r{r}/q0")).to(dev)
            U, V, Q = subspace_iter(d, r, 2, init_q=q0)
            assert U.shape == (N, r) and V.shape == (r, C) and U.dtype == torch.float16
            gU, gV = t16(G.get(FN, f"{tag}/r{r}/U")).to(dev), t16(G.get(FN, f"{tag}/r{r}/V")).to(dev)
            # U V is the projection of A on the iterated subspace: independent of QR sign conventions
            assert rel(U.float() @ V.float(), gU.float() @ gV.float()) < 2e-3, (N, C, r)
            # factors agree up to the sign of each column / row
            sign = torch.sign((U.float() * gU.float()).sum(0))
            assert rel(U.float() * sign, gU) < 5e-3 and rel(V.float() * sign[:, None], gV) < 5e-3
            ortho = U.float().t() @ U.float()
            assert torch.allclose(ortho, torch.eye(r, device=dev), atol=5e-3)


def test_subspace_iter_matches_reference_given_init_q():
    _check_subspace("cpu")


def test_subspace_iter_vs_svd():
    """compress_slowpath_test.py:185-216 (tol 0.1, 100 iterations, nearly low-rank input)."""
    from compactfusion_amd.compact.compress_lowrank import subspace_iter, svd
    torch.manual_seed(42)
    for (n, h, r) in [(128, 128, 1), (32, 256, 2), (512, 32, 2)]:
        left, right = torch.randn(n, r), torch.randn(r, h)
        lr = left @ right
        a = (lr + 0.01 * torch.norm(lr) * torch.randn(n, h) / (n * h) ** 0.5).half()
        pu, pv, _ = subspace_iter(a, r, num_iters=100)
        u, v = svd(a, r)
        assert rel(pu.float() @ pv.float(), u.float() @ v.float()) < 0.1


def _check_wire(dev):
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T
    from compactfusion_amd.compact import lowrank as LR
    from compactfusion_amd.compact.slowpath import sim_compress, slowpath_compress, slowpath_decompress
    N, C = 64, 256
    tag = f"{N}x{C}_s42"
    d = _delta(tag, N, C, dev)
    # LOW_RANK: packet = [U | V], decode = U @ V
    torch.manual_seed(42)
    pkt = slowpath_compress(d, T.LOW_RANK, rank=8)
    assert pkt.numel() == (N + C) * 8 == G.get(FN, f"{tag}/lr8/packet").size
    dec = slowpath_decompress(pkt, (N, C), T.LOW_RANK, rank=8)
    assert rel(dec, pkt[:N * 8].view(N, 8).float() @ pkt[N * 8:].view(8, C).float()) < 1e-3
    assert 0.3 < rel(dec, d) < 1.0                       # rank 8 of a noise matrix keeps little energy; sanity only
    # LOW_RANK_Q: section sizes of slowpath.py:120-131 and decode of the REFERENCE's own packet
    gp = t16(G.get(FN, f"{tag}/lrq32/packet")).to(dev)
    assert gp.numel() == LR.packet_halves(LR.LOW_RANK_Q_ID, 32, N, C) == N * 32 // 4 + 64 + C * 32 // 4 + 64
    mine = slowpath_decompress(gp, (N, C), T.LOW_RANK_Q, rank=32)
    assert rel(mine, t16(G.get(FN, f"{tag}/lrq32/dec")).to(dev)) < 2e-3
    # int4 factor quantiser on the reference's factors reproduces the reference's packet sections bit for bit
    gU, gV = t16(G.get(FN, f"{tag}/r32/U")).to(dev), t16(G.get(FN, f"{tag}/r32/V")).to(dev)
    sec = torch.empty(N * 32 // 4 + 64, dtype=torch.float16, device=dev)
    LR._q4(gU, sec)
    want = np.concatenate([G.get(FN, f"{tag}/r32/qU").reshape(-1).view(np.uint16), G.get(FN, f"{tag}/r32/sU").reshape(-1),
                           G.get(FN, f"{tag}/r32/mU").reshape(-1)])
    assert np.array_equal(sec.cpu().view(torch.int16).numpy().view(np.uint16), want)
    sec = torch.empty(C * 32 // 4 + 64, dtype=torch.float16, device=dev)
    LR._q4(gV.t(), sec)
    want = np.concatenate([G.get(FN, f"{tag}/r32/qV").reshape(-1).view(np.uint16), G.get(FN, f"{tag}/r32/sV").reshape(-1),
                           G.get(FN, f"{tag}/r32/mV").reshape(-1)])
    assert np.array_equal(sec.cpu().view(torch.int16).numpy().view(np.uint16), want)
    # simulate == decode(encode) under the same seed (compress_slowpath_test.py:128-183, INT4_TOL = 0.05)
    torch.manual_seed(7)
    sim = sim_compress(d, T.LOW_RANK_Q, rank=32)
    torch.manual_seed(7)
    dec = slowpath_decompress(slowpath_compress(d, T.LOW_RANK_Q, rank=32), (N, C), T.LOW_RANK_Q, rank=32)
    assert rel(dec, sim) < 0.05


def test_lowrank_wire_cpu(monkeypatch):
    OB.install(monkeypatch)
    _check_wire("cpu")


def _check_state_machine(dev):
    """LOW_RANK / LOW_RANK_Q presets of examples/configs.py:63-97 through compact_compress / compact_decompress: sender
    and receiver states stay bit-identical (the packet is decoded by the same code on both sides) and error feedback
    keeps the reconstruction error bounded."""
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T, CompactConfig
    N, C = 128, 512
    g = torch.Generator().manual_seed(3)
    left, right = torch.randn(N, 6, generator=g), torch.randn(6, C, generator=g)
    for typ, rank in ((T.LOW_RANK, 8), (T.LOW_RANK_Q, 32)):
        cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: None, comp_rank=rank, residual=1, ef=True))
        cur = torch.randn(N, C, generator=g).half()
        errs = []
        for t in range(5):
            x = cur.to(dev).view(1, N, C)
            ty = T.WARMUP if t == 0 else typ
            pkt = cm.compact_compress("0-0-k", x, ty, update_cache=True)
            rec = cm.compact_decompress("0-1-k", pkt.clone(), ty, x.shape, update_cache=True)
            assert torch.equal(cm.compact_cache().get_base("0-0-k"), cm.compact_cache().get_base("0-1-k")), (typ, t)
            errs.append(rel(rec.view(N, C), x.view(N, C)))
            # drift = low-rank change + small noise: what the codec is built for
            cur = (cur.float() + 0.1 * (left * torch.randn(1, 6, generator=g)) @ right / 6 ** 0.5 + 0.005 * torch.randn(N, C, generator=g)).half()
        assert errs[0] == 0.0 and max(errs[1:]) < 0.05, (typ, errs)
        if typ == T.LOW_RANK:
            assert pkt.numel() == (N + C) * rank


def test_lowrank_state_machine_cpu(monkeypatch):
    OB.install(monkeypatch)
    _check_state_machine("cpu")


@pytest.mark.gpu
def test_subspace_iter_gpu():
    _check_subspace("cuda")


@pytest.mark.gpu
def test_lowrank_wire_gpu():
    _check_wire("cuda")


@pytest.mark.gpu
def test_lowrank_state_machine_gpu():
    _check_state_machine("cuda")
