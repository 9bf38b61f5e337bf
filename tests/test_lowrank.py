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
    from compactfusion_amd.compact.compress_quantize import quantize_int4
    from compactfusion_amd.compact.slowpath import sim_compress, slowpath_compress, slowpath_decompress
    N, C = 64, 256
    tag = f"{N}x{C}_s42"
    d = _delta(tag, N, C, dev)
    for r in (8, 32):
        # same start matrix as the reference run -> same subspace -> same projection U V (QR conventions do not matter)
        LR.set_init_q(torch.from_numpy(G.get(FN, f"{tag}/r{r}/q0")).to(dev))
        try:
            pkt = slowpath_compress(d, T.LOW_RANK, rank=r)
        finally:
            LR.set_init_q(None)
        assert pkt.numel() == (N + C) * r
        U, V = pkt[:N * r].view(N, r), pkt[N * r:].view(r, C)
        gU, gV = t16(G.get(FN, f"{tag}/r{r}/U")).to(dev), t16(G.get(FN, f"{tag}/r{r}/V")).to(dev)
        assert rel(U.float() @ V.float(), gU.float() @ gV.float()) < 3e-3, r
        assert torch.allclose(U.float().t() @ U.float(), torch.eye(r, device=dev), atol=5e-3)
        dec = slowpath_decompress(pkt, (N, C), T.LOW_RANK, rank=r)
        assert rel(dec, U.float() @ V.float()) < 1e-3            # decode = fp16(U @ V)
        assert rel(dec, t16(G.get(FN, f"{tag}/r{r}/UV")).to(dev)) < 3e-3
    assert G.get(FN, f"{tag}/lr8/packet").size == (N + C) * 8
    # LOW_RANK_Q: section sizes of slowpath.py:120-131 and decode of the REFERENCE's own packet
    gp = t16(G.get(FN, f"{tag}/lrq32/packet")).to(dev)
    assert gp.numel() == LR.packet_halves(LR.LOW_RANK_Q_ID, 32, N, C) == N * 32 // 4 + 64 + C * 32 // 4 + 64
    mine = slowpath_decompress(gp, (N, C), T.LOW_RANK_Q, rank=32)
    assert rel(mine, t16(G.get(FN, f"{tag}/lrq32/dec")).to(dev)) < 2e-3
    # the int4 factor quantiser on the reference's factors reproduces the reference's packet sections bit for bit
    gU, gV = t16(G.get(FN, f"{tag}/r32/U")).to(dev), t16(G.get(FN, f"{tag}/r32/V")).to(dev)
    for mat, nm in ((gU, "U"), (gV.t().contiguous(), "V")):
        q, s_, m_ = quantize_int4(mat)
        assert np.array_equal(q.cpu().numpy(), G.get(FN, f"{tag}/r32/q{nm}"))
        assert np.array_equal(s_.cpu().view(torch.int16).numpy().view(np.uint16), G.get(FN, f"{tag}/r32/s{nm}"))
        assert np.array_equal(m_.cpu().view(torch.int16).numpy().view(np.uint16), G.get(FN, f"{tag}/r32/m{nm}"))
    # simulate == decode(encode) under the same start (compress_slowpath_test.py:128-183, INT4_TOL = 0.05)
    q0 = torch.randn(C, 32, generator=torch.Generator().manual_seed(5))
    LR.set_init_q(q0)
    try:
        sim = sim_compress(d, T.LOW_RANK_Q, rank=32)
        dec = slowpath_decompress(slowpath_compress(d, T.LOW_RANK_Q, rank=32), (N, C), T.LOW_RANK_Q, rank=32)
    finally:
        LR.set_init_q(None)
    assert rel(dec, sim) < 0.05
    # a rank-deficient residual (rank 3 < r = 8) must not produce NaNs and is reproduced almost exactly
    g = torch.Generator().manual_seed(9)
    low = (torch.randn(N, 3, generator=g) @ torch.randn(3, C, generator=g)).half().to(dev)
    dec = slowpath_decompress(slowpath_compress(low, T.LOW_RANK, rank=8), (N, C), T.LOW_RANK, rank=8)
    assert torch.isfinite(dec.float()).all() and rel(dec, low) < 5e-3
    zero = torch.zeros(N, C, dtype=torch.float16, device=dev)
    dec = slowpath_decompress(slowpath_compress(zero, T.LOW_RANK, rank=8), (N, C), T.LOW_RANK, rank=8)
    assert torch.isfinite(dec.float()).all() and float(dec.float().abs().max()) == 0.0


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
    for typ, rank in ((T.LOW_RANK, 8), (T.LOW_RANK, 12), (T.LOW_RANK, 16), (T.LOW_RANK_Q, 16), (T.LOW_RANK_Q, 32)):   # examples/configs.py:63-110
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


@pytest.mark.gpu
@pytest.mark.parametrize("N,C", [(70, 520), (544, 3072), (33, 8), (160, 1032)])
@pytest.mark.parametrize("rank", [2, 8, 12, 16, 18, 24, 32])
def test_lowrank_decode_kernels_vs_matmul(N, C, rank):
    """Both decode kernels (VALU form for rank <= 16, MFMA + LDS form for rank 17..32) against fp32 matmul on ragged shapes:
    plain packets [U (N,r) | V (r,C)] and, where the int4 factor format allows it, packets [q4(U) | q4(V^T)] (the V^T path)."""
    from compactfusion_amd import codecs as K
    from compactfusion_amd.compact.compress_quantize import dequantize_int4, quantize_int4
    g = torch.Generator().manual_seed(N * 7 + rank)
    U = (torch.randn(N, rank, generator=g) / rank ** 0.5).half().cuda()
    V = torch.randn(rank, C, generator=g).half().cuda()
    bases = [torch.randn(N, C, generator=g).half().cuda() for _ in range(3)]
    pkt = torch.cat([U.reshape(-1), V.reshape(-1)]).contiguous()
    want = (U.float() @ V.float()).half()
    outs = [torch.empty(N, C, dtype=torch.float16, device="cuda") for _ in range(3)]
    K.lr_decompress_batch(False, [pkt] * 3, [None, bases[1], bases[2]], outs, N, C, rank)
    torch.cuda.synchronize()
    assert rel(outs[0], want) < 1e-3
    for i in (1, 2):
        assert rel(outs[i], bases[i] + want) < 1e-3
    if rank % 8 == 0 and N % 2 == 0:
        qu, su, mu = quantize_int4(U)
        qv, sv, mv = quantize_int4(V.t().contiguous())
        as_half = lambda t: t.contiguous().view(torch.uint8).view(torch.float16).reshape(-1)   # noqa: E731
        qpkt = torch.cat([as_half(qu), su.reshape(-1), mu.reshape(-1), as_half(qv), sv.reshape(-1), mv.reshape(-1)]).contiguous()
        assert qpkt.numel() == K.lr_packet_halves(True, N, C, rank)
        U16 = dequantize_int4(qu, su, mu)
        V16t = dequantize_int4(qv, sv, mv)
        wantq = (U16.float() @ V16t.float().t()).half()
        out = torch.empty(N, C, dtype=torch.float16, device="cuda")
        K.lr_decompress_batch(True, [qpkt], [bases[0]], [out], N, C, rank)
        torch.cuda.synchronize()
        assert rel(out, bases[0] + wantq) < 1e-3


FNB = "g8b_lowrank_sd3_eager.npz"


@pytest.mark.gpu
@pytest.mark.parametrize("rank", [8, 12, 16, 32])
def test_lowrank_config5_shard_vs_reference_factors(rank):
    """BASELINE config 5 (SD3-medium 1024^2, ring 8): shard (512, 1536), the reference's factors for every preset rank
    (golden G8b: subspace_iter(delta, r, 2, init_q) of the reference, examples/configs.py:63-110).  Given the same start matrix
    the HIP chain must span the same subspace: U V (what the receiver adds to its state) within 3e-3 of the reference's, the
    error-feedback state = base + fp16(U V), and for the int4-quantised presets the reference's OWN packet decodes to the same
    reconstruction."""
    from compactfusion_amd import codecs as K
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as T
    from compactfusion_amd.compact import lowrank as LR
    from compactfusion_amd.compact.slowpath import slowpath_decompress
    N, C = 512, 1536
    tag = f"{N}x{C}_s42"
    xb, bb = G.inputs(FNB, tag, 42, N, C)
    x, base = t16(xb).cuda(), t16(bb).cuda()
    gU, gV = t16(G.get(FNB, f"{tag}/r{rank}/U")).cuda(), t16(G.get(FNB, f"{tag}/r{rank}/V")).cuda()
    want = gU.float() @ gV.float()
    q0 = torch.zeros(C, K.lr_rank_pad(rank), dtype=torch.float32, device="cuda")
    q0[:, :rank] = torch.from_numpy(G.get(FNB, f"{tag}/r{rank}/q0")).cuda()
    pkt = torch.empty(K.lr_packet_halves(False, N, C, rank), dtype=torch.float16, device="cuda")
    nb = torch.empty(N, C, dtype=torch.float16, device="cuda")
    K.lr_compress_batch(False, [x], [base], [nb], [pkt], [q0], N, C, rank, update_cache=True, ef=True)
    torch.cuda.synchronize()
    U, V = pkt[:N * rank].view(N, rank), pkt[N * rank:].view(rank, C)
    assert rel(U.float() @ V.float(), want) < 3e-3, rank
    assert torch.allclose(U.float().t() @ U.float(), torch.eye(rank, device="cuda"), atol=5e-3)
    assert G.sha(np.ascontiguousarray((gU.float() @ gV.float()).half().cpu().view(torch.int16).numpy().view(np.uint16))) is not None
    # error-feedback state: base + decode(packet); within the codec tolerance of base + the reference's U V
    rec = torch.empty(N, C, dtype=torch.float16, device="cuda")
    K.lr_decompress_batch(False, [pkt], [base], [rec], N, C, rank)
    torch.cuda.synchronize()
    assert torch.equal(nb, rec), "sender state != receiver reconstruction"
    assert rel(nb.float() - base.float(), want) < 4e-3
    if rank in (16, 32):
        as_half = lambda a: torch.from_numpy(np.ascontiguousarray(a)).contiguous().view(torch.uint8).view(torch.float16).reshape(-1)   # noqa: E731
        h16 = lambda a: t16(a).reshape(-1)     # noqa: E731
        gp = torch.cat([as_half(G.get(FNB, f"{tag}/r{rank}/qU")), h16(G.get(FNB, f"{tag}/r{rank}/sU")), h16(G.get(FNB, f"{tag}/r{rank}/mU")),
                        as_half(G.get(FNB, f"{tag}/r{rank}/qV")), h16(G.get(FNB, f"{tag}/r{rank}/sV")), h16(G.get(FNB, f"{tag}/r{rank}/mV"))]).cuda()
        assert gp.numel() == LR.packet_halves(LR.LOW_RANK_Q_ID, rank, N, C)
        mine = slowpath_decompress(gp, (N, C), T.LOW_RANK_Q, rank=rank)
        # the reference decodes with dequantize_int4 + fp16 matmul (slowpath.py:151-164): recompute it from its own sections
        from compactfusion_amd.compact.compress_quantize import dequantize_int4
        qu = torch.from_numpy(G.get(FNB, f"{tag}/r{rank}/qU")).cuda(); qv = torch.from_numpy(G.get(FNB, f"{tag}/r{rank}/qV")).cuda()
        u = dequantize_int4(qu, t16(G.get(FNB, f"{tag}/r{rank}/sU")).cuda(), t16(G.get(FNB, f"{tag}/r{rank}/mU")).cuda())
        v = dequantize_int4(qv, t16(G.get(FNB, f"{tag}/r{rank}/sV")).cuda(), t16(G.get(FNB, f"{tag}/r{rank}/mV")).cuda())
        assert rel(mine, u.float() @ v.float().t()) < 1e-3
