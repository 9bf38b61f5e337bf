"""Parity of the HIP kernels (through the C-ABI, libcfx.so) with the oracle - GPU box only (-m gpu).

Bar: bit-exact for everything.  The HIP kernels and the oracle both use the exact (order-independent)
sum for the scale reductions, so even the fp16 scale vectors and the error-feedback state are bit-identical;
tolerances only enter where the *reference's* own fp32 accumulation order is involved
(tests/test_oracle_golden.py)."""
import os

import numpy as np
import pytest
import torch

import _golden as G
from oracle import ref_np as R

pytestmark = pytest.mark.gpu

F16 = np.float16
CODECS = [("binary", 1, 0), ("int2", 2, 0), ("int4", 3, 0), ("int8", 4, 0), ("topk", 5, 8), ("topk", 5, 2), ("topk", 5, 16), ("topk", 5, 1), ("topk", 5, 4)]


def dev(a16):
    return torch.from_numpy(np.ascontiguousarray(a16).view(np.int16)).view(torch.float16).cuda()


def host_bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def make_inputs(seed, N, C, drift=0.1):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((N, C)).astype(F16)
    x = (base.astype(np.float32) + drift * rng.standard_normal((N, C)).astype(np.float32)).astype(F16)
    return x, base


def same_bits(a, b, what):
    a = np.asarray(a).view(np.uint16).reshape(-1)
    b = np.asarray(b).view(np.uint16).reshape(-1)
    nan_a = (a & 0x7fff) > 0x7c00
    nan_b = (b & 0x7fff) > 0x7c00
    ok = (a == b) | (nan_a & nan_b)
    assert ok.all(), f"{what}: {int((~ok).sum())}/{a.size} differ (first at {int(np.argmax(~ok))})"


def run_case(name, cid, param, x, base, N, C):
    from compactfusion_amd import codecs as K
    pkt_ref, nb_ref = R.residual_compress(name, x, base, param)
    xd, bd = dev(x), (None if base is None else dev(base))
    pkt, nb = K.compress(cid, xd, bd, N, C, param, update_cache=True)
    torch.cuda.synchronize()
    assert pkt.numel() == pkt_ref.size
    same_bits(host_bits(pkt), pkt_ref, f"{name} packet")
    same_bits(host_bits(nb), R.bits(nb_ref), f"{name} new_base")
    rec = K.decompress(cid, pkt, bd, N, C, param)
    torch.cuda.synchronize()
    same_bits(host_bits(rec), R.bits(nb_ref), f"{name} recon == sender state")
    # update_cache=False writes the same packet and no state
    pkt2, nb2 = K.compress(cid, xd, bd, N, C, param, update_cache=False)
    torch.cuda.synchronize()
    assert nb2 is None
    same_bits(host_bits(pkt2), pkt_ref, f"{name} packet (no update)")


@pytest.mark.parametrize("name,cid,param", CODECS)
@pytest.mark.parametrize("shape", [(64, 256), (256, 1152), (544, 3072), (1024, 1152), (130, 1024), (2, 512), (34, 8192)])
def test_codec_vs_oracle(name, cid, param, shape):
    N, C = shape
    if name == "topk" and (N * C) % 1024:
        pytest.skip("SPARSE needs N*C % 1024 == 0")
    x, base = make_inputs(1234 + N + C, N, C)
    run_case(name, cid, param, x, base, N, C)


@pytest.mark.parametrize("name,cid,param", CODECS[:5])
def test_config1_shape_4096x1152(name, cid, param):
    """BASELINE.json configs[0]: synthetic [1,4096,1152]."""
    torch.manual_seed(1234)
    base = torch.randn(4096, 1152).half()
    x = (base.float() + 0.1 * torch.randn(4096, 1152)).half()
    run_case(name, cid, param, host_bits(x).view(F16), host_bits(base).view(F16), 4096, 1152)


@pytest.mark.parametrize("name,cid,param", CODECS[:5])
def test_residual0_no_base(name, cid, param):
    N, C = 64, 1024
    x, _ = make_inputs(5, N, C)
    run_case(name, cid, param, x, None, N, C)


@pytest.mark.parametrize("name,cid,param", CODECS[:5])
def test_batched_equals_single(name, cid, param):
    from compactfusion_amd import codecs as K
    N, C, B = 96, 1536, 7
    xs, bs = zip(*[make_inputs(100 + i, N, C) for i in range(B)])
    xd, bd = [dev(a) for a in xs], [dev(a) for a in bs]
    pk = [torch.empty(K.packet_halves(cid, N, C, param), dtype=torch.float16, device="cuda") for _ in range(B)]
    K.compress_batch(cid, xd, bd, bd, pk, N, C, param, update_cache=True)   # in place: new_base aliases base
    recv_base = [dev(a) for a in bs]
    K.decompress_batch(cid, pk, recv_base, recv_base, N, C, param)          # in place on the receiver
    torch.cuda.synchronize()
    for i in range(B):
        pkt_ref, nb_ref = R.residual_compress(name, xs[i], bs[i], param)
        same_bits(host_bits(pk[i]), pkt_ref, f"{name}[{i}] packet")
        same_bits(host_bits(bd[i]), R.bits(nb_ref), f"{name}[{i}] in-place sender state")
        same_bits(host_bits(recv_base[i]), R.bits(nb_ref), f"{name}[{i}] in-place receiver state")


@pytest.mark.parametrize("name,cid,param", CODECS[:5])
def test_no_ef_flag(name, cid, param):
    from compactfusion_amd import codecs as K
    N, C = 64, 1024
    x, base = make_inputs(9, N, C)
    pkt, nb = K.compress(cid, dev(x), dev(base), N, C, param, update_cache=True, ef=False)
    torch.cuda.synchronize()
    same_bits(host_bits(nb), R.bits(x), "ef=False stores x (main.py:233)")
    same_bits(host_bits(pkt), R.residual_compress(name, x, base, param)[0], "packet")


@pytest.mark.parametrize("rows", [16, 32, 64, 128])
def test_tiling_independent(rows):
    """Scales come from exact integer sums: any tiling gives the same bits."""
    from compactfusion_amd import codecs as K
    N, C = 300, 1152
    x, base = make_inputs(77, N, C)
    K.set_rows_per_tile(rows)
    try:
        for name, cid, param in CODECS[:4]:
            run_case(name, cid, param, x, base, N, C)
    finally:
        K.set_rows_per_tile(0)


def test_edge_zero_delta_and_extremes():
    from compactfusion_amd import codecs as K
    N, C = 32, 512
    x, base = make_inputs(3, N, C)
    # (a) x == base: 1-bit scales are 0/0 = NaN in the reference (no epsilon, fastpath.py:165); 2-bit has the epsilon
    for name, cid, param in CODECS[:5]:
        run_case(name, cid, param, base.copy(), base, N, C)
    # (b) a constant column (min == max) and large magnitudes, signed zeros, subnormals
    x2 = x.copy()
    x2[:, 7] = base[:, 7] + F16(0.5)
    x2[3, :] = F16(60000.0)
    x2[4, :] = F16(-60000.0)
    x2[5, ::2] = F16(-0.0)
    base2 = base.copy()
    base2[5, :] = F16(0.0)
    x2[6, :] = np.float16(6e-8)
    base2[6, :] = F16(0.0)
    for name, cid, param in CODECS[:5]:
        run_case(name, cid, param, x2, base2, N, C)


GOLD_FAST = [(64, 256, 42), (64, 256, 43), (256, 1152, 42), (128, 3072, 44)]


@pytest.mark.parametrize("N,C,seed", GOLD_FAST)
def test_golden_binary_through_abi(N, C, seed):
    """Committed reference vectors (G1): packed bits exact; with the reference's own U/V in the packet,
    the dequant+add kernel reproduces the reference's new_base / recon bit for bit."""
    from compactfusion_amd import codecs as K
    fn = "g1_binary_fastpath_eager.npz"
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    pkt, nb = K.compress(1, dev(x.view(F16)), dev(base.view(F16)), N, C)
    torch.cuda.synchronize()
    words = host_bits(pkt)
    packed, u, v = R.fastpath_unpacket(words, N, C, 8)
    G.check(fn, f"{tag}/packed", packed, "packed")
    gu, gv = G.get(fn, f"{tag}/u"), G.get(fn, f"{tag}/v")
    n1, mx1 = G.ulp_diff_count(R.bits(u), gu)
    n2, mx2 = G.ulp_diff_count(R.bits(v), gv)
    assert mx1 <= 1 and mx2 <= 1 and G.rel_err(R.bits(u), gu.reshape(-1)) < 1e-3 and G.rel_err(R.bits(v), gv.reshape(-1)) < 1e-3
    gpkt = R.fastpath_packet(G.get(fn, f"{tag}/packed"), gu, gv)
    rec = K.decompress(1, dev(gpkt.view(F16)), dev(base.view(F16)), N, C)
    torch.cuda.synchronize()
    G.check(fn, f"{tag}/recon", host_bits(rec).reshape(N, C), "recon from golden packet")
    G.check(fn, f"{tag}/new_base", host_bits(rec).reshape(N, C), "== reference new_base")


@pytest.mark.parametrize("N,C,seed", GOLD_FAST)
def test_golden_int2_through_abi(N, C, seed):
    from compactfusion_amd import codecs as K
    fn = "g2_int2_fastpath_eager.npz"
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    pkt, nb = K.compress(2, dev(x.view(F16)), dev(base.view(F16)), N, C)
    torch.cuda.synchronize()
    packed, u, v = R.fastpath_unpacket(host_bits(pkt), N, C, 4)
    gp, gu, gv = G.get(fn, f"{tag}/packed"), G.get(fn, f"{tag}/u"), G.get(fn, f"{tag}/v")
    assert float((packed != gp).mean()) <= 1e-3
    assert G.ulp_diff_count(R.bits(u), gu)[1] <= 1 and G.ulp_diff_count(R.bits(v), gv)[1] <= 1
    gpkt = R.fastpath_packet(gp, gu, gv)
    rec = K.decompress(2, dev(gpkt.view(F16)), dev(base.view(F16)), N, C)
    torch.cuda.synchronize()
    G.check(fn, f"{tag}/recon", host_bits(rec).reshape(N, C), "recon from golden packet")
    G.check(fn, f"{tag}/new_base", host_bits(rec).reshape(N, C), "== reference new_base")


@pytest.mark.parametrize("N,C,seed", GOLD_FAST)
def test_golden_int2_codes_exact_given_the_reference_scales(N, C, seed):
    """k_int2_quant ALONE with the reference's own tok / chan vectors planted in the packet (cfx_int2_quantize = the Triton kernel
    _int2_quant_fastpath behind the reference's eager prologue, fastpath.py:486-580): codes == G2 `packed` and the error-feedback state
    == G2 `new_base`, bit for bit - the 2-bit codes only ever differ from the reference through <= 1-ulp flips of the scale vectors
    (reference test tests/compact/compress_fastpath_test.py:105-162)."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    fn = "g2_int2_fastpath_eager.npz"
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    gp, gu, gv = G.get(fn, f"{tag}/packed"), G.get(fn, f"{tag}/u"), G.get(fn, f"{tag}/v")
    pkt_host = R.fastpath_packet(np.zeros_like(gp), gu, gv)                    # [codes = 0 | tok | chan]
    pkt = dev(pkt_host.view(F16))
    xd, bd = dev(x.view(F16)), dev(base.view(F16))
    nb = torch.empty_like(bd)
    it = (_lib.CompItem * 1)(_lib.CompItem(xd.data_ptr(), bd.data_ptr(), nb.data_ptr(), pkt.data_ptr()))
    assert lib.cfx_int2_quantize(K.context(0), N, C, _lib.FLAG_UPDATE_CACHE, 1, it, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    packed, u, v = R.fastpath_unpacket(host_bits(pkt), N, C, 4)
    assert np.array_equal(R.bits(u).reshape(-1), np.asarray(gu).view(np.uint16).reshape(-1)) and np.array_equal(R.bits(v).reshape(-1), np.asarray(gv).view(np.uint16).reshape(-1))
    G.check(fn, f"{tag}/packed", packed, "k_int2_quant codes | the reference's scales")
    G.check(fn, f"{tag}/new_base", host_bits(nb).reshape(N, C), "k_int2_quant error-feedback state | the reference's scales")


@pytest.mark.parametrize("N,C,seed", [(64, 256, 42), (256, 1152, 43)])
def test_golden_int8_int4_through_abi(N, C, seed):
    """G4/G5: the affine codecs have no order-dependent reduction: q, scale, zp/min all bit-exact vs the reference."""
    from compactfusion_amd import codecs as K
    fn = "g3_g6_slowpath_codecs_eager.npz"
    tag = f"{N}x{C}_s{seed}"
    x, base = G.inputs(fn, tag, seed, N, C)
    pkt, _ = K.compress(4, dev(x.view(F16)), dev(base.view(F16)), N, C, update_cache=False)
    torch.cuda.synchronize()
    w = host_bits(pkt)
    qn = N * C // 2
    G.check(fn, f"{tag}/i8/q", w[:qn].view(np.int8).reshape(N, C), "int8 q")
    G.check(fn, f"{tag}/i8/scale", w[qn:qn + C].reshape(1, C), "int8 scale")
    G.check(fn, f"{tag}/i8/zp", w[qn + C:].view(np.int16).reshape(1, C), "int8 zp")
    rec = K.decompress(4, pkt, None, N, C)
    G.check(fn, f"{tag}/i8/deq", host_bits(rec).reshape(N, C), "int8 dequant")
    pkt, _ = K.compress(3, dev(x.view(F16)), dev(base.view(F16)), N, C, update_cache=False)
    torch.cuda.synchronize()
    w = host_bits(pkt)
    qn = N * C // 4
    G.check(fn, f"{tag}/i4/q", w[:qn].view(np.uint8).reshape(N // 2, C), "int4 q")
    G.check(fn, f"{tag}/i4/scale", w[qn:qn + C].reshape(1, C), "int4 scale")
    G.check(fn, f"{tag}/i4/min", w[qn + C:].reshape(1, C), "int4 min")
    rec = K.decompress(3, pkt, None, N, C)
    G.check(fn, f"{tag}/i4/deq", host_bits(rec).reshape(N, C), "int4 dequant")


@pytest.mark.parametrize("tag,N,C", [("64x256_s42", 64, 256), ("32x1024_s42", 32, 1024)])
@pytest.mark.parametrize("m", [1, 2, 4, 8, 16])
def test_golden_topk_through_abi(tag, N, C, m):
    from compactfusion_amd import codecs as K
    fn = "g7_topk_eager.npz"
    x, base = G.get(fn, f"{tag}/x"), G.get(fn, f"{tag}/base")
    pkt, _ = K.compress(5, dev(x.view(F16)), dev(base.view(F16)), N, C, m, update_cache=False)
    torch.cuda.synchronize()
    w = host_bits(pkt)
    vn = N * C // m
    G.check(fn, f"{tag}/m{m}/val", w[:vn].reshape(-1, 1024 // m), "topk val")
    G.check(fn, f"{tag}/m{m}/idx", w[vn:].view(np.uint8).reshape(-1, 512 // m), "topk idx")
    rec = K.decompress(5, pkt, None, N, C, m)
    G.check(fn, f"{tag}/m{m}/dec", host_bits(rec).reshape(N, C), "topk decompress")


@pytest.mark.parametrize("name,cid,param,shape", [("binary", 1, 0, (544, 3072)), ("int2", 2, 0, (544, 3072)),
                                                  ("int4", 3, 0, (4448, 3072)), ("int8", 4, 0, (4096, 1152)),
                                                  ("topk", 5, 8, (512, 1536))])
def test_full_size_ef_round_trip(name, cid, param, shape):
    """BASELINE.json full sizes, size-independent properties: over T drifting steps the sender's error-feedback
    state and the receiver's reconstruction stay bit-identical, and the reconstruction error stays bounded
    (error feedback does not accumulate)."""
    from compactfusion_amd import codecs as K
    N, C = shape
    g = torch.Generator(device="cpu").manual_seed(42)
    cur = torch.randn(N, C, generator=g).half()
    send_base = cur.clone().cuda()       # WARMUP step: both sides hold x_0
    recv_base = cur.clone().cuda()
    pkt = torch.empty(K.packet_halves(cid, N, C, param), dtype=torch.float16, device="cuda")
    errs = []
    for t in range(6):
        cur = (cur.float() + 0.1 * torch.randn(N, C, generator=g)).half()
        xd = cur.cuda()
        K.compress_batch(cid, [xd], [send_base], [send_base], [pkt], N, C, param, update_cache=True)
        K.decompress_batch(cid, [pkt], [recv_base], [recv_base], N, C, param)
        torch.cuda.synchronize()
        assert torch.equal(send_base.view(torch.int16), recv_base.view(torch.int16)), f"state diverged at step {t}"
        errs.append(float((recv_base.float() - xd.float()).norm() / xd.float().norm()))
    assert all(np.isfinite(errs))
    assert errs[-1] < 0.2 and errs[-1] < 2.5 * errs[0] + 1e-3, errs


@pytest.mark.parametrize("name,cid,param,shape", [("binary", 1, 0, (544, 3072)), ("int8", 4, 0, (4096, 1152)), ("int4", 3, 0, (1024, 1152)),
                                                  ("int4", 3, 0, (4448, 3072)), ("topk", 5, 8, (512, 1536)), ("int2", 2, 0, (4448, 3072))])
def test_baseline_config_sizes_vs_c_oracle(name, cid, param, shape):
    """Every BASELINE.json configuration at its FULL size (SURVEY.md section 8d shapes S1-S5), two drifting steps, packet and
    error-feedback state bit for bit against the C oracle (the numpy oracle's twin, fast enough for 13.7 M elements)."""
    from compactfusion_amd import codecs as K
    from oracle import c_oracle as CO
    N, C = shape
    x, base = make_inputs(7, N, C)
    state_c = base.copy().view(np.uint16)
    state_g = dev(base)
    pkt_g = torch.empty(K.packet_halves(cid, N, C, param), dtype=torch.float16, device="cuda")
    rng = np.random.default_rng(8)
    for step in range(2):
        pkt_c, state_c = CO.compress(name, x, state_c, N, C, param)
        K.compress_batch(cid, [dev(x)], [state_g], [state_g], [pkt_g], N, C, param, update_cache=True)
        torch.cuda.synchronize()
        same_bits(host_bits(pkt_g), pkt_c, f"{name} {shape} step {step}: packet")
        same_bits(host_bits(state_g), state_c, f"{name} {shape} step {step}: state")
        x = (x.astype(np.float32) + 0.1 * rng.standard_normal((N, C)).astype(np.float32)).astype(F16)
    # and the receiver: reconstructs the sender's state from the last packet and the previous state
    rec_c = CO.decompress(name, pkt_c, base.view(np.uint16), N, C, param)
    rec_g = K.decompress(cid, pkt_g, dev(base), N, C, param)
    torch.cuda.synchronize()
    same_bits(host_bits(rec_g), rec_c, f"{name} {shape}: receiver")


@pytest.mark.parametrize("decay", [0.5, 0.3, 1.0])
@pytest.mark.parametrize("shape", [(64, 256), (130, 1024), (544, 3072)])
def test_residual2_kernels_vs_oracle(shape, decay):
    """cfx_residual2_delta / cfx_residual2_update (second-order residual, main.py:244-266) bit for bit against the oracle,
    out of place and in place; decay 0.3 is not an fp16 number (torch multiplies by the fp32 scalar)."""
    from compactfusion_amd import codecs as K
    N, C = shape
    rng = np.random.default_rng(N + int(decay * 10))
    x = rng.standard_normal((N, C)).astype(F16)
    base = (x.astype(np.float32) + 0.1 * rng.standard_normal((N, C)).astype(np.float32)).astype(F16)
    dbase = (0.05 * rng.standard_normal((N, C))).astype(F16)
    recv = (0.05 * rng.standard_normal((N, C))).astype(F16)
    dd = torch.empty(N, C, dtype=torch.float16, device="cuda")
    K.residual2_delta(dev(x), dev(base), dev(dbase), dd)
    same_bits(host_bits(dd), R.bits(R.residual2_delta(x, base, dbase)), "residual2 delta")
    want_nb, want_nd = R.residual2_update(base, dbase, recv, decay)
    nb, nd = torch.empty_like(dd), torch.empty_like(dd)
    K.residual2_update(dev(base), dev(dbase), dev(recv), nb, nd, decay)
    same_bits(host_bits(nb), R.bits(want_nb), "residual2 new_base")
    same_bits(host_bits(nd), R.bits(want_nd), "residual2 new_delta_base")
    b_, d_ = dev(base), dev(dbase)
    K.residual2_update(b_, d_, dev(recv), b_, d_, decay)                 # in place
    torch.cuda.synchronize()
    same_bits(host_bits(b_), R.bits(want_nb), "residual2 new_base (in place)")
    same_bits(host_bits(d_), R.bits(want_nd), "residual2 new_delta_base (in place)")
    with pytest.raises(ValueError):
        K.residual2_delta(dev(x)[:, :C // 2].contiguous(), dev(base), dev(dbase), dd)


def test_error_codes():
    from compactfusion_amd import codecs as K
    from compactfusion_amd._lib import CfxError
    x = torch.zeros(8, 20, dtype=torch.float16, device="cuda")
    with pytest.raises(ValueError):
        K.packet_bytes(1, 8, 20)                      # C % 8 != 0
    with pytest.raises(ValueError):
        K.packet_bytes(3, 7, 64)                      # int4 needs even N
    with pytest.raises(ValueError):
        K.packet_bytes(5, 8, 64, 8)                   # topk needs N*C % 1024 == 0
    with pytest.raises(ValueError):
        K.packet_bytes(9, 8, 64)                      # unknown codec
    with pytest.raises(CfxError):
        K.compress(1, torch.zeros(8, 64, dtype=torch.float16), None, 8, 64)   # CPU tensor: no fallback
    big = torch.zeros(8 * 64 + 8, dtype=torch.float16, device="cuda")
    mis = big[1:1 + 8 * 64].view(8, 64)               # 2-byte aligned only
    with pytest.raises(CfxError):
        K.compress(1, mis, None, 8, 64)


# ---- in-launch finalize (last-arriver tickets, write-through partial sums) -------------------------------------------------
@pytest.mark.parametrize("name,cid", [("binary", 1), ("int2", 2), ("int4", 3), ("int8", 4)])
@pytest.mark.parametrize("shape,B", [((544, 3072), 2), ((64, 256), 16), ((130, 1024), 5), ((4448, 3072), 2), ((1024, 1152), 3), ((2, 512), 1)])
def test_fused_finalize_equals_two_kernel_sequence_under_load(name, cid, shape, B):
    """The compress launch reduces its partial sums inside the launch (tickets + write-through partials): every word of the
    packets must equal the two-kernel sequence's and the oracle's, launch after launch (ticket blocks are recycled), with
    another stream hammering HBM (uneven load) and the consumer's caches warm (the same workspace is re-read every launch)."""
    from compactfusion_amd import codecs as K
    N, C = shape
    ins = [make_inputs(900 + 7 * i + N, N, C) for i in range(B)]
    xd, bd = [dev(a) for a, _ in ins], [dev(b) for _, b in ins]
    ph = K.packet_halves(cid, N, C, 0)

    def packets():
        return [torch.full((ph,), float("nan"), dtype=torch.float16, device="cuda") for _ in range(B)]

    K.set_fused_finalize(False)
    try:
        want = packets()
        K.compress_batch(cid, xd, bd, [None] * B, want, N, C, 0, update_cache=False)
        torch.cuda.synchronize()
    finally:
        K.set_fused_finalize(True)
    for i in (0, B - 1):
        same_bits(host_bits(want[i]), R.residual_compress(name, ins[i][0], ins[i][1], 0)[0], f"{name} two-kernel packet vs oracle")
    # background load on another stream: a big copy loop
    side = torch.cuda.Stream()
    junk_a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    junk_b = torch.empty_like(junk_a)
    reps = 40
    got = [packets() for _ in range(reps)]
    with torch.cuda.stream(side):
        for _ in range(60):
            junk_b.copy_(junk_a)
    for r in range(reps):
        K.compress_batch(cid, xd, bd, [None] * B, got[r], N, C, 0, update_cache=False)
    torch.cuda.synchronize()
    for r in range(reps):
        for i in range(B):
            assert torch.equal(got[r][i].view(torch.int16), want[i].view(torch.int16)), f"{name} launch {r} tensor {i}: fused finalize differs"


def test_ride_along_reconstruction_in_the_compress_launch():
    """cfx_compress_batch_ex: the previous layer's deferred error-feedback update rides in the statistics launch."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    N, C = 544, 3072
    ctx = K.context(0)
    (x0, b0), (x1, b1) = make_inputs(1, N, C), make_inputs(2, N, C)
    xd = [dev(x0), dev(x1)]
    state = [dev(b0), dev(b1)]
    pk = [torch.zeros(K.packet_halves(1, N, C), dtype=torch.float16, device="cuda") for _ in range(2)]
    ws = K.workspace(1, N, C, 0, 1, 0)
    sh = torch.cuda.current_stream().cuda_stream
    # layer 0: compress only (state untouched); layer 1: compress + layer 0's error-feedback update riding along
    c0 = (_lib.CompItem * 1)(_lib.CompItem(xd[0].data_ptr(), state[0].data_ptr(), None, pk[0].data_ptr()))
    assert lib.cfx_compress_batch_ex(ctx, 1, N, C, 0, 0, 1, c0, 0, None, ws.data_ptr(), ws.numel(), sh) == 0
    c1 = (_lib.CompItem * 1)(_lib.CompItem(xd[1].data_ptr(), state[1].data_ptr(), None, pk[1].data_ptr()))
    ride = (_lib.DecompItem * 1)(_lib.DecompItem(pk[0].data_ptr(), state[0].data_ptr(), state[0].data_ptr()))
    assert lib.cfx_compress_batch_ex(ctx, 1, N, C, 0, 0, 1, c1, 1, ride, ws.data_ptr(), ws.numel(), sh) == 0
    torch.cuda.synchronize()
    p0, nb0 = R.residual_compress("binary", x0, b0, 0)
    p1, _ = R.residual_compress("binary", x1, b1, 0)
    same_bits(host_bits(pk[0]), p0, "layer 0 packet")
    same_bits(host_bits(pk[1]), p1, "layer 1 packet")
    same_bits(host_bits(state[0]), R.bits(nb0), "layer 0 state after the ride-along update")
    same_bits(host_bits(state[1]), R.bits(b1), "layer 1 state untouched")
    # a ride item with a codec other than 1-bit is refused
    assert lib.cfx_compress_batch_ex(ctx, 2, N, C, 0, 0, 1, c1, 1, ride, ws.data_ptr(), ws.numel(), sh) == -4


@pytest.mark.parametrize("name,cid", [("binary", 1), ("int2", 2)])
def test_statistics_exact_sums_on_outliers_subnormals_and_signed_zeros(name, cid):
    """The exact (order-independent) sums of |x - base| over a tensor that mixes ordinary rows with outliers up to the fp16
    maximum, subnormals, exact zeros and -0.0 (whose sign bit must pack as >= 0): everything matches the oracle bit for bit."""
    N, C = 544, 3072
    rng = np.random.default_rng(77)
    base = rng.standard_normal((N, C)).astype(F16)
    x = (base.astype(np.float32) + 0.1 * rng.standard_normal((N, C)).astype(np.float32)).astype(F16)
    x[5, 17] = F16(60000.0); x[5, 18] = F16(-65504.0)              # |d| near the fp16 maximum
    x[100, :64] = F16(9.0) + base[100, :64]                        # just above the threshold of the fast lane
    x[101, :64] = (base[101, :64].astype(np.float32) + 7.99).astype(F16)   # just below it
    x[200, :] = base[200, :]                                       # exact zeros
    x[201, ::3] = F16(-0.0); base[201, ::3] = F16(0.0)             # -0.0 - 0.0 = -0.0 >= 0 is True
    x[300, :128] = (base[300, :128].astype(np.float32) + 6e-8).astype(F16)
    base[301, :128] = 0; x[301, :128] = np.frombuffer(np.arange(1, 129, dtype=np.uint16).tobytes(), dtype=F16)   # subnormals
    run_case(name, cid, 0, x, base, N, C)


@pytest.mark.parametrize("own_ef", ["items", "flag"])
@pytest.mark.parametrize("shape,B,NG", [((544, 3072), 2, 16), ((512, 1536), 2, 16), ((256, 1152), 1, 3), ((130, 1024), 2, 5),
                                       ((64, 256), 1, 2), ((1100, 3072), 1, 4), ((64, 264), 1, 2), ((2, 512), 1, 2), ((34, 8192), 2, 6),
                                       ((1024, 1152), 2, 4), ((4096, 1152), 1, 2)])
def test_gated_reconstruction_in_the_compress_launch(shape, B, NG, own_ef):
    """cfx_compress_batch_gated: the reconstruction of tensors whose packets THIS launch produces (own error feedback, looped-back
    peers) runs inside the compress launch behind an arrival gate.  Packets and states equal the oracle's bit for bit on every
    launch of a long back-to-back sequence (ticket / gate ring reuse, workgroups of consecutive launches in flight together),
    with a bandwidth hog on a second stream for uneven load; (64, 264) does not qualify and takes the two-launch sequence.
    own_ef = items: the rank's own error feedback is given as gated items too; flag: CFX_FLAG_UPDATE_CACHE - the compress
    workgroups do it from the registers they loaded for the statistics."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    N, C = shape
    ctx = K.context(0)
    L = 3                                                    # distinct layers cycled through
    rng = np.random.default_rng(5)
    xs, bs = [], []
    for l in range(L):
        for i in range(B):
            x, b = make_inputs(100 + 10 * l + i, N, C)
            xs.append(x); bs.append(b)
    xd = [dev(x) for x in xs]
    own = [dev(b) for b in bs]                               # the rank's own states (updated in place by gated items)
    # gated items: the B own states (EF) first, then G - B "peers" whose states start as copies of an own state
    src = [i % B for i in range(NG)]
    peer = [[dev(bs[l * B + src[g]]) for g in range(B, NG)] for l in range(L)]
    pk = [torch.zeros(K.packet_halves(1, N, C), dtype=torch.float16, device="cuda") for _ in range(L * B)]
    ws = K.workspace(1, N, C, 0, B, 0)
    sh = torch.cuda.current_stream().cuda_stream
    hog_s = torch.cuda.Stream()
    hog = torch.empty(64 << 20, dtype=torch.float16, device="cuda")
    comp, gated = [], []
    flag = own_ef == "flag"
    FL = _lib.FLAG_UPDATE_CACHE if flag else 0
    NGI = NG - B if flag else NG                              # gated items actually passed
    for l in range(L):
        comp.append((_lib.CompItem * B)(*[_lib.CompItem(xd[l * B + i].data_ptr(), own[l * B + i].data_ptr(),
                                                        own[l * B + i].data_ptr() if flag else None, pk[l * B + i].data_ptr())
                                          for i in range(B)]))
        items = []
        for g in range(B if flag else 0, NG):
            st = own[l * B + g] if g < B else peer[l][g - B]
            items.append(_lib.DecompItem(pk[l * B + src[g]].data_ptr(), st.data_ptr(), st.data_ptr()))
        gated.append((_lib.DecompItem * NGI)(*items))
    # oracle: T rounds over the L layers (x fixed, state evolving by error feedback)
    T = 4
    ostate = [R.bits(b).copy() for b in bs]
    opk = [None] * (L * B)
    reps = 90 if N * C >= 544 * 3072 else 30                 # > 256 launches at the big shapes: the ring wraps
    for t in range(T):
        for l in range(L):
            with torch.cuda.stream(hog_s):
                hog.add_(1.0)
            assert lib.cfx_compress_batch_gated(ctx, 1, N, C, 0, FL, B, comp[l], 0, None, NGI, gated[l], ws.data_ptr(), ws.numel(), sh) == 0
            for i in range(B):
                k = l * B + i
                p, nb = R.residual_compress("binary", xs[k], ostate[k].view(F16), 0)
                opk[k] = p; ostate[k] = R.bits(nb).copy()
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0
    for l in range(L):
        for i in range(B):
            same_bits(host_bits(pk[l * B + i]), opk[l * B + i], f"packet layer {l} item {i}")
            same_bits(host_bits(own[l * B + i]), ostate[l * B + i], f"own state layer {l} item {i}")
        for g in range(B, NG):
            same_bits(host_bits(peer[l][g - B]), ostate[l * B + src[g]], f"looped-back peer state layer {l} item {g}")
    # long back-to-back sequence: the states keep evolving, compare the end state with an ungated replay of the same launches
    ref_own = [o.clone() for o in own]
    ref_pk = [torch.zeros_like(p) for p in pk]
    for r in range(reps):
        l = r % L
        assert lib.cfx_compress_batch_gated(ctx, 1, N, C, 0, FL, B, comp[l], 0, None, NGI, gated[l], ws.data_ptr(), ws.numel(), sh) == 0
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0
    for r in range(reps):
        l = r % L
        c = (_lib.CompItem * B)(*[_lib.CompItem(xd[l * B + i].data_ptr(), ref_own[l * B + i].data_ptr(), None, ref_pk[l * B + i].data_ptr())
                                  for i in range(B)])
        assert lib.cfx_compress_batch_ex(ctx, 1, N, C, 0, 0, B, c, 0, None, ws.data_ptr(), ws.numel(), sh) == 0
        d = (_lib.DecompItem * B)(*[_lib.DecompItem(ref_pk[l * B + i].data_ptr(), ref_own[l * B + i].data_ptr(), ref_own[l * B + i].data_ptr())
                                    for i in range(B)])
        assert lib.cfx_decompress_batch(ctx, 1, N, C, 0, B, d, sh) == 0
    torch.cuda.synchronize()
    for k in range(L * B):
        assert torch.equal(own[k].view(torch.int16), ref_own[k].view(torch.int16)), f"own state {k} after {reps} gated launches"
        assert torch.equal(pk[k].view(torch.int16), ref_pk[k].view(torch.int16)), f"packet {k} after {reps} gated launches"
    for l in range(L):
        for g in range(B, NG):
            assert torch.equal(peer[l][g - B].view(torch.int16), own[l * B + src[g]].view(torch.int16)), f"peer {g} of layer {l} diverged from its owner"


@pytest.mark.parametrize("name,cid", [("int2", 2), ("int4", 3), ("int8", 4), ("topk", 5)])
@pytest.mark.parametrize("shape,B,NP", [((544, 3072), 2, 14), ((512, 1536), 2, 14), ((256, 1152), 1, 3), ((130, 1024), 2, 5), ((64, 264), 1, 2),
                                        ((2, 512), 1, 2), ((34, 8192), 2, 6), ((1024, 1152), 2, 4), ((4096, 1152), 1, 1), ((4448, 3072), 2, 6)])
def test_gated_int2_layer_in_one_launch(shape, B, NP, name, cid):
    """cfx_compress_batch_gated, 2-bit / int4 / int8 codec: statistics + finalize, quantise + error feedback of the own tensors (from the
    registers the statistics pass loaded) and the reconstruction of NP looped-back peers in ONE launch (two gates).  Packets and states
    equal the oracle's bit for bit over several rounds, then a long back-to-back sequence equals the multi-launch sequence; (64, 264)
    takes the fallback.  The min/max codecs must issue the layer launch (kernel id 31) and nothing else wherever their statistics tiles
    are co-resident - or, tall tensors ((4448, 3072): BASELINE config 4's shard), a column block's tiles at a time."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    N, C = shape
    ctx = K.context(0)
    CID = cid
    PRM = 8 if cid == 5 else 0                      # top-k: 1:8
    if cid == 5 and (N * C) % 1024:
        pytest.skip("the top-k codec works on the flat (-1, 1024) view")
    L = 3
    xs, bs = [], []
    for l in range(L):
        for i in range(B):
            x, b = make_inputs(300 + 10 * l + i, N, C)
            xs.append(x); bs.append(b)
    xd = [dev(x) for x in xs]
    own = [dev(b) for b in bs]
    src = [i % B for i in range(NP)]
    peer = [[dev(bs[l * B + src[g]]) for g in range(NP)] for l in range(L)]
    pk = [torch.zeros(K.packet_halves(CID, N, C, PRM), dtype=torch.float16, device="cuda") for _ in range(L * B)]
    ws = K.workspace(CID, N, C, PRM, B, 0)
    wsp, wsn = (None, 0) if ws is None else (ws.data_ptr(), ws.numel())
    sh = torch.cuda.current_stream().cuda_stream
    hog_s = torch.cuda.Stream()
    hog = torch.empty(64 << 20, dtype=torch.float16, device="cuda")
    comp, gated = [], []
    for l in range(L):
        comp.append((_lib.CompItem * B)(*[_lib.CompItem(xd[l * B + i].data_ptr(), own[l * B + i].data_ptr(), own[l * B + i].data_ptr(),
                                                        pk[l * B + i].data_ptr()) for i in range(B)]))
        gated.append((_lib.DecompItem * NP)(*[_lib.DecompItem(pk[l * B + src[g]].data_ptr(), peer[l][g].data_ptr(), peer[l][g].data_ptr())
                                              for g in range(NP)]))

    def go(l):
        assert lib.cfx_compress_batch_gated(ctx, CID, N, C, PRM, _lib.FLAG_UPDATE_CACHE, B, comp[l], 0, None, NP, gated[l],
                                            wsp, wsn, sh) == 0
    RL = 32 if (N + 31) // 32 <= 32 else 64
    PL, CBk = (N + RL - 1) // RL, (C + 511) // 512
    if cid == 5 or (cid in (3, 4) and C % 16 == 0 and PL <= 128 and (PL * CBk * B <= 500 or (RL == 64 and PL * 4 <= 512))):
        # the layer launch itself: one kernel (id 31), no statistics / quantise / reconstruction launch beside it
        import ctypes as _ct
        torch.cuda.synchronize()
        assert lib.cfx_profile_enable(ctx, 64, 0xffffffff, 1) == 0
        keep_own, keep_peer = [o.clone() for o in own], [[p.clone() for p in pl] for pl in peer]
        go(0)
        torch.cuda.synchronize()
        ids, ms = (_ct.c_int * 64)(), (_ct.c_float * 64)()
        n_ids = lib.cfx_profile_read(ctx, ids, ms, 64)
        lib.cfx_profile_enable(ctx, 0, 0, 1)
        assert [ids[i] for i in range(n_ids)] == [31], [ids[i] for i in range(n_ids)]
        for o, k_ in zip(own, keep_own):
            o.copy_(k_)
        for pl, kl in zip(peer, keep_peer):
            for p_, k_ in zip(pl, kl):
                p_.copy_(k_)
    ostate = [R.bits(b).copy() for b in bs]
    opk = [None] * (L * B)
    for t in range(3):
        for l in range(L):
            with torch.cuda.stream(hog_s):
                hog.add_(1.0)
            go(l)
            for i in range(B):
                k = l * B + i
                p, nb = R.residual_compress(name, xs[k], ostate[k].view(F16), PRM)
                opk[k] = p; ostate[k] = R.bits(nb).copy()
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0
    for l in range(L):
        for i in range(B):
            same_bits(host_bits(pk[l * B + i]), opk[l * B + i], f"packet layer {l} item {i}")
            same_bits(host_bits(own[l * B + i]), ostate[l * B + i], f"own state layer {l} item {i}")
        for g in range(NP):
            same_bits(host_bits(peer[l][g]), ostate[l * B + src[g]], f"looped-back peer state layer {l} item {g}")
    reps = 90 if N * C >= 544 * 3072 else 30
    ref_own = [o.clone() for o in own]
    ref_pk = [torch.zeros_like(p) for p in pk]
    for r in range(reps):
        go(r % L)
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0
    for r in range(reps):
        l = r % L
        c = (_lib.CompItem * B)(*[_lib.CompItem(xd[l * B + i].data_ptr(), ref_own[l * B + i].data_ptr(), ref_own[l * B + i].data_ptr(),
                                                ref_pk[l * B + i].data_ptr()) for i in range(B)])
        assert lib.cfx_compress_batch_ex(ctx, CID, N, C, PRM, _lib.FLAG_UPDATE_CACHE, B, c, 0, None, wsp, wsn, sh) == 0
    torch.cuda.synchronize()
    for k in range(L * B):
        assert torch.equal(own[k].view(torch.int16), ref_own[k].view(torch.int16)), f"own state {k} after {reps} gated launches"
        assert torch.equal(pk[k].view(torch.int16), ref_pk[k].view(torch.int16)), f"packet {k} after {reps} gated launches"
    for l in range(L):
        for g in range(NP):
            assert torch.equal(peer[l][g].view(torch.int16), own[l * B + src[g]].view(torch.int16)), f"peer {g} of layer {l} diverged from its owner"


def test_gated_launch_argument_errors():
    """cfx_compress_batch_gated / cfx_plan_add_compress_gated refuse what they cannot run: an unknown codec, too many gated items, a null
    item list, misaligned pointers."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    N, C = 64, 256
    ctx = K.context(0)
    x = torch.randn(N, C, device="cuda").half(); b = torch.zeros_like(x)
    pk = torch.zeros(K.packet_halves(1, N, C) + 8, dtype=torch.float16, device="cuda")
    ws = K.workspace(3, N, C, 0, 1, 0)
    sh = torch.cuda.current_stream().cuda_stream
    c = (_lib.CompItem * 1)(_lib.CompItem(x.data_ptr(), b.data_ptr(), None, pk.data_ptr()))
    g = (_lib.DecompItem * 1)(_lib.DecompItem(pk.data_ptr(), b.data_ptr(), b.data_ptr()))
    assert lib.cfx_compress_batch_gated(ctx, 9, 64, 256, 8, 0, 1, c, 0, None, 1, g, ws.data_ptr(), ws.numel(), sh) == -4      # no such codec
    assert lib.cfx_compress_batch_gated(ctx, 1, N, C, 0, 0, 1, c, 0, None, 17, g, ws.data_ptr(), ws.numel(), sh) == -5     # > CFX_MAX_BATCH
    assert lib.cfx_compress_batch_gated(ctx, 1, N, C, 0, 0, 1, c, 0, None, 1, None, ws.data_ptr(), ws.numel(), sh) == -5
    bad = (_lib.DecompItem * 1)(_lib.DecompItem(pk.data_ptr() + 2, b.data_ptr(), b.data_ptr()))
    assert lib.cfx_compress_batch_gated(ctx, 1, N, C, 0, 0, 1, c, 0, None, 1, bad, ws.data_ptr(), ws.numel(), sh) == -3    # alignment
    plan = lib.cfx_plan_create(ctx)
    assert lib.cfx_plan_add_compress_gated(plan, 9, N, C, 8, 0, 1, c, 0, None, 1, g, ws.data_ptr(), ws.numel()) == -2     # (a plan op checks codec and shape together)
    assert lib.cfx_plan_add_compress_gated(plan, 1, N, C, 0, 0, 1, c, 0, None, 1, g, ws.data_ptr(), ws.numel()) == 0
    lib.cfx_plan_destroy(plan)
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0


@pytest.mark.parametrize("name,cid", [("int4", 3), ("int8", 4)])
def test_minmax_layer_launches_from_more_streams_than_rings(name, cid):
    """The int4 / int8 layer launch hands its partials over as tagged words in an arena the context keeps PER STREAM (per ticket ring: 8).
    Twelve streams taking turns make the rings - and their arenas - change hands again and again while launches are in flight: every
    launch must still equal the oracle (a change of owner waits for the previous owner's launches), and no wait may time out."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    ctx = K.context(0)
    N, C, S, ROUNDS = 256, 1024, 12, 3
    streams = [torch.cuda.Stream() for _ in range(S)]
    xs = [[make_inputs(700 + 10 * s + r, N, C)[0] for r in range(ROUNDS)] for s in range(S)]
    b0 = [make_inputs(900 + s, N, C)[1] for s in range(S)]
    xd = [[dev(x) for x in row] for row in xs]
    own = [dev(b) for b in b0]
    peer = [dev(b) for b in b0]
    pk = [torch.zeros(K.packet_halves(cid, N, C), dtype=torch.float16, device="cuda") for _ in range(S)]
    ws = [K.workspace(cid, N, C, 0, 1, 0, stream_handle=st.cuda_stream) for st in streams]
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for s, st in enumerate(streams):
            c = (_lib.CompItem * 1)(_lib.CompItem(xd[s][r].data_ptr(), own[s].data_ptr(), own[s].data_ptr(), pk[s].data_ptr()))
            g = (_lib.DecompItem * 1)(_lib.DecompItem(pk[s].data_ptr(), peer[s].data_ptr(), peer[s].data_ptr()))
            assert lib.cfx_compress_batch_gated(ctx, cid, N, C, 0, _lib.FLAG_UPDATE_CACHE, 1, c, 0, None, 1, g, ws[s].data_ptr(), ws[s].numel(),
                                                st.cuda_stream) == 0, lib.cfx_last_error_string(ctx)
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0
    for s in range(S):
        st = R.bits(b0[s]).copy()
        for r in range(ROUNDS):
            p, nb = R.residual_compress(name, xs[s][r], st.view(F16), 0)
            st = R.bits(nb).copy()
        same_bits(host_bits(own[s]), st, f"stream {s}: own state")
        same_bits(host_bits(peer[s]), st, f"stream {s}: looped-back peer state")
        same_bits(host_bits(pk[s]), p, f"stream {s}: last packet")


def test_gated_item_with_a_packet_from_an_earlier_launch():
    """A gated item whose packet was NOT produced by this launch (an older packet of the same codec and shape) is legal: it waits
    for the gate like the others and reconstructs from the packet as it stands."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    N, C = 544, 3072
    ctx = K.context(0)
    (x0, b0), (x1, b1) = make_inputs(7, N, C), make_inputs(8, N, C)
    xd0, xd1 = dev(x0), dev(x1)
    s0, s1 = dev(b0), dev(b1)
    other = dev(b1)                                       # a state that receives the OLD packet (layer 0's) during layer 1's launch
    pk0 = torch.zeros(K.packet_halves(1, N, C), dtype=torch.float16, device="cuda"); pk1 = torch.zeros_like(pk0)
    ws = K.workspace(1, N, C, 0, 1, 0)
    sh = torch.cuda.current_stream().cuda_stream
    c0 = (_lib.CompItem * 1)(_lib.CompItem(xd0.data_ptr(), s0.data_ptr(), None, pk0.data_ptr()))
    assert lib.cfx_compress_batch_ex(ctx, 1, N, C, 0, 0, 1, c0, 0, None, ws.data_ptr(), ws.numel(), sh) == 0
    c1 = (_lib.CompItem * 1)(_lib.CompItem(xd1.data_ptr(), s1.data_ptr(), s1.data_ptr(), pk1.data_ptr()))
    g = (_lib.DecompItem * 1)(_lib.DecompItem(pk0.data_ptr(), other.data_ptr(), other.data_ptr()))
    assert lib.cfx_compress_batch_gated(ctx, 1, N, C, 0, _lib.FLAG_UPDATE_CACHE, 1, c1, 0, None, 1, g, ws.data_ptr(), ws.numel(), sh) == 0
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0
    p0, _ = R.residual_compress("binary", x0, b0, 0)
    p1, nb1 = R.residual_compress("binary", x1, b1, 0)
    same_bits(host_bits(pk1), p1, "layer 1 packet")
    same_bits(host_bits(s1), R.bits(nb1), "layer 1 own state (CFX_FLAG_UPDATE_CACHE inside the gated launch)")
    same_bits(host_bits(other), R.bits(R.residual_decompress("binary", p0, b1, N, C)), "state reconstructed from the older packet")


@pytest.mark.parametrize("name,cid", [("binary", 1), ("int2", 2)])
@pytest.mark.parametrize("shape,drift", [((544, 3072), 50.0), ((130, 1024), 400.0), ((544, 3072), 1.5), ((544, 3072), 3000.0)])
def test_statistics_partials_beyond_32_bits(name, cid, shape, drift):
    """The fused paths hand their partial sums from workgroup to workgroup in words narrower than the sums can get, with a sentinel + a
    64-bit side channel for sums that do not fit: 32-bit words in the stand-alone launches (every row and column partial at drift 50
    and 400, only the row partials at drift 1.5: 512 channels x 1.2 > 256), 40-bit tagged words in the layer launches (a tile's sum of
    |d| beyond 65536: the row partials at drift 400, the column partials too at drift 3000).  All still match the oracle bit for bit."""
    N, C = shape
    x, base = make_inputs(900 + int(drift), N, C, drift=drift)
    run_case(name, cid, 0, x, base, N, C)


@pytest.mark.parametrize("name,cid", [("binary", 1), ("int2", 2)])
@pytest.mark.parametrize("drift", [1.5, 400.0, 3000.0])
def test_layer_launch_partials_beyond_the_tagged_words(name, cid, drift):
    """cfx_compress_batch_gated at the FLUX shard with residuals of order 1 (round 4: the 32-bit row partials overflowed at an average |d|
    of 0.5 and the launch took a second round trip) and far beyond (the 40-bit tagged words' own side channel: row partials at drift 400,
    column partials at drift 3000): packets, sender states and looped-back peer states == the oracle, bit for bit, two launches in a row
    (the second meets the first one's tags in the arena)."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    N, C, B, NP = 544, 3072, 2, 6
    ctx = K.context(0)
    sh = torch.cuda.current_stream().cuda_stream
    ws = K.workspace(cid, N, C, 0, B, 0)
    for rep in range(2):
        xs = [make_inputs(4000 + 10 * rep + i + int(drift), N, C, drift=drift) for i in range(B)]
        xd = [dev(x) for x, _ in xs]
        own = [dev(b) for _, b in xs]
        peer = [dev(xs[g % B][1]) for g in range(NP)]
        pk = [torch.zeros(K.packet_halves(cid, N, C), dtype=torch.float16, device="cuda") for _ in range(B)]
        comp = (_lib.CompItem * B)(*[_lib.CompItem(xd[i].data_ptr(), own[i].data_ptr(), own[i].data_ptr(), pk[i].data_ptr()) for i in range(B)])
        gated = (_lib.DecompItem * NP)(*[_lib.DecompItem(pk[g % B].data_ptr(), peer[g].data_ptr(), peer[g].data_ptr()) for g in range(NP)])
        assert lib.cfx_compress_batch_gated(ctx, cid, N, C, 0, _lib.FLAG_UPDATE_CACHE, B, comp, 0, None, NP, gated, ws.data_ptr(), ws.numel(), sh) == 0
        torch.cuda.synchronize()
        assert lib.cfx_gate_errors(ctx) == 0
        for i in range(B):
            pkt_ref, nb_ref = R.residual_compress(name, xs[i][0], xs[i][1], 0)
            same_bits(host_bits(pk[i]), pkt_ref, f"{name} packet {i} (drift {drift})")
            same_bits(host_bits(own[i]), R.bits(nb_ref), f"{name} sender state {i}")
            for g in range(i, NP, B):
                same_bits(host_bits(peer[g]), R.bits(nb_ref), f"{name} peer state {g}")


def tagwrap_case(name, cid, N, C):
    """Body of test_layer_launches_across_the_tag_wrap: runs in a process of its own that loaded the DEVELOPER library (the hook that sets a
    context's launch tags - cfx_dev_set_launch_tags, include/cfx_dev.h - is not part of the product library)."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    B, NP = 2, 4
    ctx = K.context(0)
    sh = torch.cuda.current_stream().cuda_stream
    ws = K.workspace(cid, N, C, 0, B, 0)
    assert lib.cfx_dev_set_launch_tags(ctx, (1 << 24) - 4, (1 << 31) - 5) == 0
    for rep in range(8):
        xs = [make_inputs(9100 + 10 * rep + i, N, C) for i in range(B)]
        xd = [dev(x) for x, _ in xs]
        own = [dev(b) for _, b in xs]
        peer = [dev(xs[g % B][1]) for g in range(NP)]
        pk = [torch.zeros(K.packet_halves(cid, N, C), dtype=torch.float16, device="cuda") for _ in range(B)]
        comp = (_lib.CompItem * B)(*[_lib.CompItem(xd[i].data_ptr(), own[i].data_ptr(), own[i].data_ptr(), pk[i].data_ptr()) for i in range(B)])
        gated = (_lib.DecompItem * NP)(*[_lib.DecompItem(pk[g % B].data_ptr(), peer[g].data_ptr(), peer[g].data_ptr()) for g in range(NP)])
        assert lib.cfx_compress_batch_gated(ctx, cid, N, C, 0, _lib.FLAG_UPDATE_CACHE, B, comp, 0, None, NP, gated, ws.data_ptr(), ws.numel(), sh) == 0
        torch.cuda.synchronize()
        assert lib.cfx_gate_errors(ctx) == 0, f"launch {rep}"
        for i in range(B):
            pkt_ref, nb_ref = R.residual_compress(name, xs[i][0], xs[i][1], 0)
            same_bits(host_bits(pk[i]), pkt_ref, f"{name} packet {i}, launch {rep}")
            same_bits(host_bits(own[i]), R.bits(nb_ref), f"{name} sender state {i}, launch {rep}")
            for g in range(i, NP, B):
                same_bits(host_bits(peer[g]), R.bits(nb_ref), f"{name} peer state {g}, launch {rep}")


@pytest.mark.parametrize("name,cid,shape", [("binary", 1, (544, 3072)), ("int2", 2, (544, 3072)), ("int4", 3, (1024, 1152)), ("int8", 4, (1024, 1152))])
def test_layer_launches_across_the_tag_wrap(name, cid, shape):
    """The layer launches tag what they hand over with numbers the context gives out in sequence - 24 bits (1-bit / 2-bit), 31 bits (int4 /
    int8) - and where a sequence wraps the tagged arenas are zeroed and the numbers start over (never 0: that is what an untouched word
    carries).  A serving process gets there (16.7 million layer launches are a few thousand FLUX images); cfx_dev_set_launch_tags walks
    a context to three launches before both wraps, then eight launches in a row must equal the oracle bit for bit.  The hook lives in the
    developer library only (include/cfx_dev.h): the case runs in a child process that loads libcfx_dev.so - same sources, same kernels
    plus the probes - in place of libcfx.so (tests/tagwrap_child.py)."""
    import subprocess
    import sys
    from compactfusion_amd.build import build_lib
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, CFX_LIBCFX_PATH=build_lib(dev_probes=True))
    r = subprocess.run([sys.executable, os.path.join(here, "tagwrap_child.py"), name, str(cid), str(shape[0]), str(shape[1])],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])


def test_gated_launches_on_two_streams_at_once():
    """Two independent sequences of gated launches issued alternately on two streams (own workspaces, one context): their
    workgroups share the chip, each launch has its own ticket / gate slot, nobody waits on the other sequence - both end states
    equal an ungated replay of the same launches."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    N, C, B, NP = 544, 3072, 2, 6
    ctx = K.context(0)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    seqs = []
    for q in range(2):
        xs = [make_inputs(700 + 10 * q + i, N, C) for i in range(B)]
        xd = [dev(x) for x, _ in xs]
        own = [dev(b) for _, b in xs]
        peer = [dev(xs[g % B][1]) for g in range(NP)]
        pk = [torch.zeros(K.packet_halves(1, N, C), dtype=torch.float16, device="cuda") for _ in range(B)]
        ws = K.workspace(1, N, C, 0, B, 0, streams[q].cuda_stream)
        comp = (_lib.CompItem * B)(*[_lib.CompItem(xd[i].data_ptr(), own[i].data_ptr(), own[i].data_ptr(), pk[i].data_ptr()) for i in range(B)])
        gated = (_lib.DecompItem * NP)(*[_lib.DecompItem(pk[g % B].data_ptr(), peer[g].data_ptr(), peer[g].data_ptr()) for g in range(NP)])
        seqs.append(dict(xs=xs, xd=xd, own=own, peer=peer, pk=pk, ws=ws, comp=comp, gated=gated))
    torch.cuda.synchronize()
    reps = 40
    for r in range(reps):
        for q in range(2):
            sq = seqs[q]
            assert lib.cfx_compress_batch_gated(ctx, 1, N, C, 0, _lib.FLAG_UPDATE_CACHE, B, sq["comp"], 0, None, NP, sq["gated"],
                                                sq["ws"].data_ptr(), sq["ws"].numel(), streams[q].cuda_stream) == 0
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0
    sh = torch.cuda.current_stream().cuda_stream
    ws = K.workspace(1, N, C, 0, B, 0)
    for q in range(2):
        sq = seqs[q]
        ref_own = [dev(b) for _, b in sq["xs"]]
        ref_pk = [torch.zeros_like(p) for p in sq["pk"]]
        c = (_lib.CompItem * B)(*[_lib.CompItem(sq["xd"][i].data_ptr(), ref_own[i].data_ptr(), ref_own[i].data_ptr(), ref_pk[i].data_ptr()) for i in range(B)])
        for r in range(reps):
            assert lib.cfx_compress_batch_ex(ctx, 1, N, C, 0, _lib.FLAG_UPDATE_CACHE, B, c, 0, None, ws.data_ptr(), ws.numel(), sh) == 0
        torch.cuda.synchronize()
        for i in range(B):
            assert torch.equal(sq["own"][i].view(torch.int16), ref_own[i].view(torch.int16)), f"sequence {q}: own state {i}"
            assert torch.equal(sq["pk"][i].view(torch.int16), ref_pk[i].view(torch.int16)), f"sequence {q}: packet {i}"
        for g in range(NP):
            assert torch.equal(sq["peer"][g].view(torch.int16), sq["own"][g % B].view(torch.int16)), f"sequence {q}: peer {g} diverged from its owner"


def test_gated_and_ride_along_items_in_one_launch():
    """One launch carrying all three groups: compress (layer 1), gated reconstructions fed by layer 1's packet, and an ungated
    ride-along reconstruction fed by layer 0's (already complete) packet."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    N, C = 544, 3072
    ctx = K.context(0)
    (x0, b0), (x1, b1) = make_inputs(21, N, C), make_inputs(22, N, C)
    xd0, xd1, s0, s1 = dev(x0), dev(x1), dev(b0), dev(b1)
    peers = [dev(b1) for _ in range(3)]
    pk0 = torch.zeros(K.packet_halves(1, N, C), dtype=torch.float16, device="cuda"); pk1 = torch.zeros_like(pk0)
    ws = K.workspace(1, N, C, 0, 1, 0)
    sh = torch.cuda.current_stream().cuda_stream
    c0 = (_lib.CompItem * 1)(_lib.CompItem(xd0.data_ptr(), s0.data_ptr(), None, pk0.data_ptr()))
    assert lib.cfx_compress_batch_ex(ctx, 1, N, C, 0, 0, 1, c0, 0, None, ws.data_ptr(), ws.numel(), sh) == 0
    c1 = (_lib.CompItem * 1)(_lib.CompItem(xd1.data_ptr(), s1.data_ptr(), None, pk1.data_ptr()))
    ride = (_lib.DecompItem * 1)(_lib.DecompItem(pk0.data_ptr(), s0.data_ptr(), s0.data_ptr()))
    items = [_lib.DecompItem(pk1.data_ptr(), s1.data_ptr(), s1.data_ptr())] + [_lib.DecompItem(pk1.data_ptr(), p.data_ptr(), p.data_ptr()) for p in peers]
    gated = (_lib.DecompItem * len(items))(*items)
    for _ in range(1):
        assert lib.cfx_compress_batch_gated(ctx, 1, N, C, 0, 0, 1, c1, 1, ride, len(items), gated, ws.data_ptr(), ws.numel(), sh) == 0
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0
    p0, nb0 = R.residual_compress("binary", x0, b0, 0)
    p1, nb1 = R.residual_compress("binary", x1, b1, 0)
    same_bits(host_bits(pk1), p1, "layer 1 packet")
    same_bits(host_bits(s0), R.bits(nb0), "layer 0 state (ride-along)")
    same_bits(host_bits(s1), R.bits(nb1), "layer 1 state (gated)")
    for i, p in enumerate(peers):
        same_bits(host_bits(p), R.bits(nb1), f"peer {i} (gated)")


@pytest.mark.parametrize("codec", [1, 2])
def test_gated_layer_launches_beside_attention_kernels(codec):
    """The one-launch layer (1-bit and 2-bit) while a COMPUTE kernel - PyTorch's SDPA, long and CU-filling - runs on another stream:
    the gated workgroups must still get their slots behind the statistics group (forward progress rests on in-order dispatch), no
    gate may time out, and a long back-to-back sequence must equal the multi-launch replay bit for bit."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    N, C, B, NP = 544, 3072, 2, 14
    ctx = K.context(0)
    xs, bs = [], []
    for i in range(B):
        x, b = make_inputs(900 + i, N, C)
        xs.append(x); bs.append(b)
    xd = [dev(x) for x in xs]
    own = [dev(b) for b in bs]
    peer = [dev(bs[g % B]) for g in range(NP)]
    pk = [torch.zeros(K.packet_halves(codec, N, C), dtype=torch.float16, device="cuda") for _ in range(B)]
    ws = K.workspace(codec, N, C, 0, B, 0)
    sh = torch.cuda.current_stream().cuda_stream
    comp = (_lib.CompItem * B)(*[_lib.CompItem(xd[i].data_ptr(), own[i].data_ptr(), own[i].data_ptr(), pk[i].data_ptr()) for i in range(B)])
    gated = (_lib.DecompItem * NP)(*[_lib.DecompItem(pk[g % B].data_ptr(), peer[g].data_ptr(), peer[g].data_ptr()) for g in range(NP)])
    att_s = torch.cuda.Stream()
    g = torch.Generator(device="cuda").manual_seed(1)
    q, k, v = (torch.randn(1, 24, 4096, 128, device="cuda", dtype=torch.float16, generator=g) for _ in range(3))
    reps = 120
    torch.cuda.synchronize()
    with torch.cuda.stream(att_s):
        for _ in range(60):                                   # ~1 ms each: attention kernels cover the whole sequence of launches
            torch.ops.aten._scaled_dot_product_flash_attention(q, k, v, 0.0, False, False, scale=128 ** -0.5)
    for _ in range(reps):
        assert lib.cfx_compress_batch_gated(ctx, codec, N, C, 0, _lib.FLAG_UPDATE_CACHE, B, comp, 0, None, NP, gated, ws.data_ptr(), ws.numel(), sh) == 0
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0, "a gate timed out beside the attention kernels"
    # replay without the gate machinery: compress (+ EF) then reconstruction, on fresh copies of the initial states
    r_own = [dev(b) for b in bs]
    r_peer = [dev(bs[g % B]) for g in range(NP)]
    r_pk = [torch.zeros_like(p) for p in pk]
    c2 = (_lib.CompItem * B)(*[_lib.CompItem(xd[i].data_ptr(), r_own[i].data_ptr(), r_own[i].data_ptr(), r_pk[i].data_ptr()) for i in range(B)])
    d2 = (_lib.DecompItem * NP)(*[_lib.DecompItem(r_pk[g % B].data_ptr(), r_peer[g].data_ptr(), r_peer[g].data_ptr()) for g in range(NP)])
    for _ in range(reps):
        assert lib.cfx_compress_batch(ctx, codec, N, C, 0, _lib.FLAG_UPDATE_CACHE, B, c2, ws.data_ptr(), ws.numel(), sh) == 0
        assert lib.cfx_decompress_batch(ctx, codec, N, C, 0, NP, d2, sh) == 0
    torch.cuda.synchronize()
    for i in range(B):
        assert torch.equal(own[i].view(torch.int16), r_own[i].view(torch.int16)), f"own state {i}"
        assert torch.equal(pk[i].view(torch.int16), r_pk[i].view(torch.int16)), f"packet {i}"
    for gi in range(NP):
        assert torch.equal(peer[gi].view(torch.int16), r_peer[gi].view(torch.int16)), f"peer state {gi}"


def test_gated_int2_falls_back_when_its_statistics_group_cannot_be_co_resident():
    """The 2-bit one-launch layer needs every statistics workgroup resident at once (each waits at gate 1 for all the others).  A
    batch with more of them than the device holds - K,V of a (2304, 3072) shard: 864 - and ANY batch on a CU-masked stream must
    take the multi-launch form instead of spinning into the gate timeout; results equal the plain sequence bit for bit."""
    from compactfusion_amd import _lib, codecs as K, lanes
    lib = _lib.load()
    ctx = K.context(0)
    for (N, C, stream) in ((2304, 3072, torch.cuda.current_stream()), (544, 3072, lanes.exchange_stream(0))):
        B, NP = 2, 4
        xs, bs = [], []
        for i in range(B):
            x, b = make_inputs(950 + i, N, C)
            xs.append(x); bs.append(b)
        sh = stream.cuda_stream
        with torch.cuda.stream(stream):
            xd = [dev(x) for x in xs]
            outs = []
            for gated_call in (True, False):
                own = [dev(b) for b in bs]
                peer = [dev(bs[g % B]) for g in range(NP)]
                pk = [torch.zeros(K.packet_halves(2, N, C), dtype=torch.float16, device="cuda") for _ in range(B)]
                ws = K.workspace(2, N, C, 0, B, 0, stream_handle=sh)
                comp = (_lib.CompItem * B)(*[_lib.CompItem(xd[i].data_ptr(), own[i].data_ptr(), own[i].data_ptr(), pk[i].data_ptr()) for i in range(B)])
                dq = (_lib.DecompItem * NP)(*[_lib.DecompItem(pk[g % B].data_ptr(), peer[g].data_ptr(), peer[g].data_ptr()) for g in range(NP)])
                for _ in range(3):
                    if gated_call:
                        assert lib.cfx_compress_batch_gated(ctx, 2, N, C, 0, _lib.FLAG_UPDATE_CACHE, B, comp, 0, None, NP, dq, ws.data_ptr(), ws.numel(), sh) == 0
                    else:
                        assert lib.cfx_compress_batch(ctx, 2, N, C, 0, _lib.FLAG_UPDATE_CACHE, B, comp, ws.data_ptr(), ws.numel(), sh) == 0
                        assert lib.cfx_decompress_batch(ctx, 2, N, C, 0, NP, dq, sh) == 0
                torch.cuda.synchronize()
                outs.append((own, peer, pk))
        assert lib.cfx_gate_errors(ctx) == 0, (N, C)
        for a, b_ in zip(outs[0], outs[1]):
            for t0, t1 in zip(a, b_):
                assert torch.equal(t0.view(torch.int16), t1.view(torch.int16)), (N, C)
