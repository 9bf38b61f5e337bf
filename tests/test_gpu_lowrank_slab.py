"""The slab-resident low-rank chain (csrc/cfx_lrslab.hip, one persistent launch, rank <= 32, N <= 576): what the launch
structure could break - ragged shapes, rank-deficient residuals, sub-batching, reuse of the hand-over arena across shapes,
run-to-run reproducibility, hipGraph capture - checked against an fp64 replay of the reference's iteration
(xfuser/compact/compress_lowrank.py:14-61, Householder QR) with the same start matrix, and against the receiver's kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def subspace_iter_fp64(D, Q0, iters=2):
    """compress_lowrank.py:45-61 in fp64: Q <- qr(A^T (A Q)) twice, U = qr(A Q), V = U^T A."""
    A = D.double()
    Q = Q0.double()
    for _ in range(iters):
        Q, _ = torch.linalg.qr(A.t() @ (A @ Q))
    U, _ = torch.linalg.qr(A @ Q)
    return U, U.t() @ A


def make(N, C, rank, seed, decay=0.7, resid_rank=None):
    """x, base with a residual of decaying spectrum (sigma_k = decay^k) over `resid_rank` directions (default min(N, 48)) plus a
    little noise, and the start matrix."""
    from compactfusion_amd import codecs as K
    g = torch.Generator().manual_seed(seed)
    k = resid_rank or min(N, 48)
    L = torch.linalg.qr(torch.randn(N, k, generator=g))[0]
    R = torch.linalg.qr(torch.randn(C, k, generator=g))[0]
    s = decay ** torch.arange(k, dtype=torch.float32)
    D = (L * s) @ R.t() * (N * C) ** 0.5 * 0.05
    if resid_rank is None:
        D = D + 1e-3 * torch.randn(N, C, generator=g)
    base = torch.randn(N, C, generator=g).half()
    x = (base.float() + D).half()
    q0 = torch.zeros(C, K.lr_rank_pad(rank))
    q0[:, :rank] = torch.linalg.qr(torch.randn(C, rank, generator=g))[0]
    return x.cuda(), base.cuda(), q0.cuda()


def run(xs, bases, q0s, N, C, rank, ef=True):
    from compactfusion_amd import codecs as K
    B = len(xs)
    pk = [torch.empty(K.lr_packet_halves(False, N, C, rank), dtype=torch.float16, device="cuda") for _ in range(B)]
    nb = [torch.empty(N, C, dtype=torch.float16, device="cuda") for _ in range(B)]
    K.lr_compress_batch(False, xs, bases, nb, pk, q0s, N, C, rank, update_cache=True, ef=ef)
    torch.cuda.synchronize()
    return pk, nb


def check(x, base, q0, pkt, nb, N, C, rank, tol=3e-3):
    from compactfusion_amd import codecs as K
    U, V = pkt[:N * rank].view(N, rank).float(), pkt[N * rank:].view(rank, C).float()
    D = x - base                                                  # fp16, one rounding: what the kernel factorises
    Ur, Vr = subspace_iter_fp64(D.float(), q0[:, :rank])
    assert torch.isfinite(pkt.float()).all()
    assert rel(U @ V, (Ur @ Vr).float()) < tol
    assert torch.allclose(U.t() @ U, torch.eye(rank, device="cuda"), atol=5e-3)
    rec = torch.empty(N, C, dtype=torch.float16, device="cuda")
    K.lr_decompress_batch(False, [pkt], [base], [rec], N, C, rank)
    torch.cuda.synchronize()
    assert torch.equal(nb, rec), "sender state != receiver reconstruction"


def slab_chain_taken(N, C, rank):
    """the chain under test only runs when its workgroups fit the device (C / 32 per tensor, one per CU)"""
    return 32 <= N <= 576 and C % 128 == 0 and C >= 512 and rank <= 32 and C // 32 <= 250


@pytest.mark.parametrize("N,C", [(544, 3072), (512, 1536), (100, 512), (33, 640), (576, 1024), (256, 6144)])
@pytest.mark.parametrize("rank", [2, 8, 12, 16, 24, 32])
def test_projection_and_states(N, C, rank):
    assert slab_chain_taken(N, C, rank)
    # sigma_k = 0.7^k down to rank 16 (sigma_16 / sigma_1 = 5e-3); beyond that 0.85^k: 0.7^31 = 1.6e-5 is below what fp32 products
    # resolve at all (the reference's own fp32 iteration is off by more than the tolerance there)
    x, base, q0 = make(N, C, rank, seed=N + C + rank, decay=0.7 if rank <= 16 else 0.85)
    pk, nb = run([x], [base], [q0], N, C, rank)
    check(x, base, q0, pk[0], nb[0], N, C, rank)


def test_rank_deficient_residual_and_zero_residual():
    N, C, rank = 544, 3072, 8
    x, base, q0 = make(N, C, rank, seed=5, resid_rank=3, decay=0.5)
    pk, nb = run([x], [base], [q0], N, C, rank)
    U, V = pk[0][:N * rank].view(N, rank).float(), pk[0][N * rank:].view(rank, C).float()
    D = (x - base).float()
    assert torch.isfinite(pk[0].float()).all()
    # a rank-3 residual (up to its fp16 rounding): the projection keeps it
    assert rel(U @ V, D) < 2e-2
    # x == base: nothing to send, the state stays
    pk, nb = run([base.clone()], [base], [q0], N, C, rank)
    assert torch.isfinite(pk[0].float()).all()
    assert float(pk[0].float().abs().max()) == 0.0
    assert torch.equal(nb[0], base)


@pytest.mark.parametrize("N,C,rank", [(544, 3072, 8), (544, 3072, 16), (512, 1536, 8), (512, 1536, 32),
                                      (1024, 1152, 8), (1024, 1152, 32), (2048, 1152, 16)])     # (N > 576: the multi-launch chain, whose Y is scaled per product too)
@pytest.mark.parametrize("amp", [0.02, 40.0, 300.0])
def test_residual_magnitude(N, C, rank, amp):
    """The iteration is scale-free but its fp16 MFMA operands are not: Y = A Q grows with sigma, A^T Y with sigma^2.  The kernel
    rescales the N x r matrix by a power of two before every product; residual entries from 1e-3 up to ~100 must give the same
    projection (relative) and finite packets."""
    from compactfusion_amd import codecs as K
    g = torch.Generator().manual_seed(7 + rank)
    k = 48
    L = torch.linalg.qr(torch.randn(N, k, generator=g))[0]
    R = torch.linalg.qr(torch.randn(C, k, generator=g))[0]
    s = (0.7 if rank <= 16 else 0.85) ** torch.arange(k, dtype=torch.float32)
    D = ((L * s) @ R.t() * (N * C) ** 0.5 * 0.05 + 1e-3 * torch.randn(N, C, generator=g)) * amp
    assert float(D.abs().max()) < 6.0e4
    x = D.half().cuda()
    base = torch.zeros(N, C, dtype=torch.float16, device="cuda")
    q0 = torch.zeros(C, K.lr_rank_pad(rank))
    q0[:, :rank] = torch.linalg.qr(torch.randn(C, rank, generator=g))[0]
    q0 = q0.cuda()
    pk, nb = run([x], [base], [q0], N, C, rank)
    check(x, base, q0, pk[0], nb[0], N, C, rank)


def test_exactly_low_rank_large_entries_no_base():
    """a rank-3 matrix with entries of a few units and no state behind it (the first step of a layer), through the plugin-level
    entry: used to overflow the fp16 operands of the second product"""
    from compactfusion_amd.compact.slowpath import slowpath_compress, slowpath_decompress
    from compactfusion_amd.compact.utils import COMPACT_COMPRESS_TYPE as T
    g = torch.Generator().manual_seed(3)
    for (N, C) in ((544, 3072), (512, 1536)):
        low = (torch.randn(N, 3, generator=g) @ torch.randn(3, C, generator=g)).half().cuda()
        for rank in (8, 16):
            pkt = slowpath_compress(low, T.LOW_RANK, rank=rank)
            assert torch.isfinite(pkt.float()).all()
            out = slowpath_decompress(pkt, (N, C), T.LOW_RANK, rank=rank)
            assert rel(out, low) < 2e-3


def test_batches_bigger_than_one_launch_and_reproducible():
    """C = 3072: 96 workgroups per tensor, two tensors per launch - a batch of 5 is three launches on the same arena"""
    N, C, rank = 544, 3072, 8
    data = [make(N, C, rank, seed=100 + i) for i in range(5)]
    xs, bs, qs = [d[0] for d in data], [d[1] for d in data], [d[2] for d in data]
    pk, nb = run(xs, bs, qs, N, C, rank)
    for i in range(5):
        check(xs[i], bs[i], qs[i], pk[i], nb[i], N, C, rank)
    pk2, nb2 = run(xs, bs, qs, N, C, rank)
    for i in range(5):
        assert torch.equal(pk[i], pk2[i]) and torch.equal(nb[i], nb2[i]), "not reproducible run to run"
    # one at a time: the same bits as in the batch (the sums have a fixed order that does not depend on the launch's batch)
    for i in (0, 4):
        p1, n1 = run([xs[i]], [bs[i]], [qs[i]], N, C, rank)
        assert torch.equal(p1[0], pk[i]) and torch.equal(n1[0], nb[i])


@pytest.mark.parametrize("N,C,rank", [(128, 512, 8), (96, 640, 16), (160, 1152, 8), (160, 1024, 32), (544, 3072, 16)])
def test_sums_do_not_depend_on_the_batch(N, C, rank):
    """How the workgroups of a tensor are dealt to the XCDs depends on the launch's batch (tensors interleaved when the batch divides 8,
    tensor after tensor otherwise), and so does who sums which share of a sum; the order of additions must not: the same bits for every
    batch size, and again when the launch finds the hand-over arena as an earlier launch left it."""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    ctx = K.context(0)
    fit = 254 // (C // 32)
    data = [make(N, C, rank, seed=400 + i, decay={8: 0.7, 16: 0.8, 32: 0.9}[rank]) for i in range(8)]
    xs, bs, qs = [d[0] for d in data], [d[1] for d in data], [d[2] for d in data]
    ref_pk, ref_nb = [], []
    for i in range(8):
        p1, n1 = run([xs[i]], [bs[i]], [qs[i]], N, C, rank)
        check(xs[i], bs[i], qs[i], p1[0], n1[0], N, C, rank)
        ref_pk.append(p1[0]); ref_nb.append(n1[0])
    for B in (1, 2, 3, 4, 5, 8):
        if B > max(fit, 1) and B not in (2, 8):
            continue                                                  # (a batch above what fits is split into launches of `fit`: covered by 2 and 8)
        for rep in range(2):
            pk, nb = run(xs[:B], bs[:B], qs[:B], N, C, rank)
            for i in range(B):
                assert torch.equal(pk[i], ref_pk[i]) and torch.equal(nb[i], ref_nb[i]), f"batch {B} tensor {i} run {rep}: bits differ"
    assert lib.cfx_gate_errors(ctx) == 0


@pytest.mark.parametrize("N,C,rank", [(4096, 1152, 8), (1024, 1152, 16), (2048, 1536, 32)])
def test_multi_launch_chain_above_576_rows_reproducible_and_batch_independent(N, C, rank):
    """Shards with more than 576 rows take the multi-launch chain (csrc/cfx_lowrank.hip): Z = D^T Y is summed by whichever workgroup of a
    column tile arrives last, in split order - the bits must not depend on who that is, nor on what else is in the batch."""
    data = [make(N, C, rank, seed=300 + i, decay=0.85 if rank > 16 else 0.7) for i in range(3)]
    xs, bs, qs = [d[0] for d in data], [d[1] for d in data], [d[2] for d in data]
    pk, nb = run(xs, bs, qs, N, C, rank)
    for i in range(3):
        check(xs[i], bs[i], qs[i], pk[i], nb[i], N, C, rank)
    hog_a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    hog_b = torch.empty_like(hog_a)
    side = torch.cuda.Stream()
    for rep in range(6):
        if rep >= 3:                                              # beside a copy stream: other arrival orders
            with torch.cuda.stream(side):
                for _ in range(8):
                    hog_b.copy_(hog_a)
        pk2, nb2 = run(xs, bs, qs, N, C, rank)
        for i in range(3):
            assert torch.equal(pk[i], pk2[i]) and torch.equal(nb[i], nb2[i]), "not reproducible run to run"
    torch.cuda.synchronize()
    for i in (0, 2):
        p1, n1 = run([xs[i]], [bs[i]], [qs[i]], N, C, rank)
        assert torch.equal(p1[0], pk[i]) and torch.equal(n1[0], nb[i]), "the bits depend on the batch"


def test_arena_survives_changing_shapes():
    """the hand-over arena is re-laid-out (and zeroed) when the shape changes; stale tagged words of another layout must never
    be taken for this launch's"""
    a = (544, 3072, 8)
    b = (512, 1536, 16)
    da, db = make(*a, seed=1), make(*b, seed=2)
    ref_a = run([da[0]], [da[1]], [da[2]], *a)
    ref_b = run([db[0]], [db[1]], [db[2]], *b)
    for _ in range(3):
        ra = run([da[0]], [da[1]], [da[2]], *a)
        rb = run([db[0]], [db[1]], [db[2]], *b)
        assert torch.equal(ra[0][0], ref_a[0][0]) and torch.equal(ra[1][0], ref_a[1][0])
        assert torch.equal(rb[0][0], ref_b[0][0]) and torch.equal(rb[1][0], ref_b[1][0])
    # many launches on one layout: the 2-bit sequence tags wrap around every four sums
    for _ in range(9):
        ra = run([da[0]], [da[1]], [da[2]], *a)
    assert torch.equal(ra[0][0], ref_a[0][0]) and torch.equal(ra[1][0], ref_a[1][0])


def test_no_error_feedback_and_no_base():
    from compactfusion_amd import codecs as K
    N, C, rank = 512, 1536, 8
    x, base, q0 = make(N, C, rank, seed=9)
    pk, nb = run([x], [base], [q0], N, C, rank, ef=False)
    assert torch.equal(nb[0], x), "error feedback off: the state is the activation itself"
    pk2, _ = run([x], [base], [q0], N, C, rank, ef=True)
    assert torch.equal(pk[0], pk2[0])
    # no base (first step): the residual is x
    pkt = torch.empty(K.lr_packet_halves(False, N, C, rank), dtype=torch.float16, device="cuda")
    nbn = torch.empty(N, C, dtype=torch.float16, device="cuda")
    K.lr_compress_batch(False, [x], [None], [nbn], [pkt], [q0], N, C, rank, update_cache=True, ef=True)
    torch.cuda.synchronize()
    U, V = pkt[:N * rank].view(N, rank).float(), pkt[N * rank:].view(rank, C).float()
    Ur, Vr = subspace_iter_fp64(x.float(), q0[:, :rank])
    assert rel(U @ V, (Ur @ Vr).float()) < 3e-3
    rec = torch.empty(N, C, dtype=torch.float16, device="cuda")
    K.lr_decompress_batch(False, [pkt], [None], [rec], N, C, rank)
    torch.cuda.synchronize()
    assert torch.equal(nbn, rec)


def test_capturable_in_a_hip_graph():
    """nothing in the launch depends on a host-side counter: captured once, replayed, it keeps producing the eager result"""
    from compactfusion_amd import codecs as K
    N, C, rank = 544, 3072, 8
    data = [make(N, C, rank, seed=40 + i) for i in range(2)]
    xs, bs, qs = [d[0] for d in data], [d[1] for d in data], [d[2] for d in data]
    s = torch.cuda.Stream()
    pk = [torch.empty(K.lr_packet_halves(False, N, C, rank), dtype=torch.float16, device="cuda") for _ in range(2)]
    nb = [torch.empty(N, C, dtype=torch.float16, device="cuda") for _ in range(2)]
    with torch.cuda.stream(s):
        K.lr_compress_batch(False, xs, bs, nb, pk, qs, N, C, rank, update_cache=True, ef=True)     # warm-up: arena, attributes
        s.synchronize()
        want_p, want_n = [p.clone() for p in pk], [n.clone() for n in nb]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            K.lr_compress_batch(False, xs, bs, nb, pk, qs, N, C, rank, update_cache=True, ef=True)
    for _ in range(5):
        for t in pk + nb:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        for i in range(2):
            assert torch.equal(pk[i], want_p[i]) and torch.equal(nb[i], want_n[i])


def test_plan_ops_equal_direct_calls():
    """cfx_plan_add_lr_compress / cfx_plan_add_lr_decompress: a LOW_RANK layer replayed from a native plan gives the bits of the
    direct calls (sender state, packets, receiver states)"""
    import ctypes
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    ctx = K.context(0)
    N, C, rank, B, R = 512, 1536, 8, 2, 6
    data = [make(N, C, rank, seed=70 + i) for i in range(B)]
    xs, bs, qs = [d[0] for d in data], [d[1] for d in data], [d[2] for d in data]
    peers = [torch.randn(N, C, device="cuda").half() for _ in range(R)]
    # direct
    pk, nb = run(xs, bs, qs, N, C, rank)
    rec = [torch.empty(N, C, dtype=torch.float16, device="cuda") for _ in range(R)]
    K.lr_decompress_batch(False, [pk[j % B] for j in range(R)], peers, rec, N, C, rank)
    torch.cuda.synchronize()
    # plan
    pk2 = [torch.zeros_like(p) for p in pk]
    nb2 = [torch.zeros_like(n) for n in nb]
    rec2 = [torch.zeros_like(r_) for r_ in rec]
    wsb = lib.cfx_lr_workspace_bytes(0, N, C, rank, 16)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    plan = lib.cfx_plan_create(ctx)
    c = (_lib.CompItem * B)(*[_lib.CompItem(xs[i].data_ptr(), bs[i].data_ptr(), nb2[i].data_ptr(), pk2[i].data_ptr()) for i in range(B)])
    qp = (ctypes.c_void_p * B)(*[q.data_ptr() for q in qs])
    assert lib.cfx_plan_add_lr_compress(plan, 0, N, C, rank, _lib.FLAG_UPDATE_CACHE, B, c, qp, ws.data_ptr(), wsb) == 0
    d = (_lib.DecompItem * R)(*[_lib.DecompItem(pk2[j % B].data_ptr(), peers[j].data_ptr(), rec2[j].data_ptr()) for j in range(R)])
    assert lib.cfx_plan_add_lr_decompress(plan, 0, N, C, rank, R, d, ws.data_ptr(), wsb) == 1
    assert lib.cfx_plan_add_lr_compress(plan, 0, N, C, 7, 0, B, c, qp, ws.data_ptr(), wsb) < 0, "odd rank must be refused"
    for _ in range(2):
        assert lib.cfx_plan_run(plan, 0, 2, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    lib.cfx_plan_destroy(plan)
    for i in range(B):
        assert torch.equal(pk[i], pk2[i]) and torch.equal(nb[i], nb2[i])
    for j in range(R):
        assert torch.equal(rec[j], rec2[j])


def test_cu_masked_lane_runs_the_multi_launch_chain():
    """a 32-CU exchange lane cannot hold C / 32 co-resident workgroups: the call must take the six-launch chain there (no wait
    that never ends) and give the same projection"""
    from compactfusion_amd import codecs as K, lanes
    N, C, rank = 544, 3072, 8
    x, base, q0 = make(N, C, rank, seed=21)
    want_p, want_n = run([x], [base], [q0], N, C, rank)
    s = lanes.exchange_stream(0)
    pkt = torch.empty(K.lr_packet_halves(False, N, C, rank), dtype=torch.float16, device="cuda")
    nb = torch.empty(N, C, dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        K.lr_compress_batch(False, [x], [base], [nb], [pkt], [q0], N, C, rank, update_cache=True, ef=True, stream=s)
    s.synchronize()
    assert K.gate_errors(0) == 0 if hasattr(K, "gate_errors") else True
    check(x, base, q0, pkt, nb, N, C, rank)
    U0, V0 = want_p[0][:N * rank].view(N, rank).float(), want_p[0][N * rank:].view(rank, C).float()
    U1, V1 = pkt[:N * rank].view(N, rank).float(), pkt[N * rank:].view(rank, C).float()
    assert rel(U1 @ V1, U0 @ V0) < 3e-3


def test_hand_over_under_load_and_soak():
    """the tagged hand-over must not depend on timing: 600 back-to-back launches (rank 8 and 32 alternating shapes every 100) while
    another stream saturates the memory system with copies, every result equal to the quiet run's bits; no wait may time out"""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    ctx = K.context(0)
    cfgs = [(544, 3072, 8, False), (512, 1536, 32, True)]
    data, ref = [], []
    for (N, C, rank, q) in cfgs:
        d = [make(N, C, rank, seed=300 + i) for i in range(2)]
        xs, bs, qs = [t[0] for t in d], [t[1] for t in d], [t[2] for t in d]
        pk = [torch.empty(K.lr_packet_halves(q, N, C, rank), dtype=torch.float16, device="cuda") for _ in range(2)]
        nb = [torch.empty(N, C, dtype=torch.float16, device="cuda") for _ in range(2)]
        K.lr_compress_batch(q, xs, bs, nb, pk, qs, N, C, rank, update_cache=True, ef=True)
        torch.cuda.synchronize()
        data.append((xs, bs, qs, pk, nb))
        ref.append(([p.clone() for p in pk], [n.clone() for n in nb]))
    hog = torch.cuda.Stream()
    src = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    stop = False
    for rnd in range(6):
        (N, C, rank, q), (xs, bs, qs, pk, nb), (rp, rn) = cfgs[rnd & 1], data[rnd & 1], ref[rnd & 1]
        with torch.cuda.stream(hog):
            for _ in range(40):
                dst.copy_(src, non_blocking=True)
        for it in range(100):
            K.lr_compress_batch(q, xs, bs, nb, pk, qs, N, C, rank, update_cache=True, ef=True)
            if it % 25 == 24:
                torch.cuda.synchronize()
                for i in range(2):
                    # bit patterns (a LOW_RANK_Q packet's codes, seen as fp16, contain NaNs)
                    assert torch.equal(pk[i].view(torch.int16), rp[i].view(torch.int16)) and torch.equal(nb[i], rn[i]), (rnd, it, i)
                with torch.cuda.stream(hog):
                    for _ in range(40):
                        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0


def test_two_streams_do_not_starve_each_other():
    """two streams issuing the persistent launch back to back: a launch of one must never share the CUs with a launch of the other
    (each would wait for workgroups that find no room) - the library orders them; results equal the single-stream bits"""
    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    N, C, rank = 544, 3072, 8
    sets = []
    for si in range(2):
        d = [make(N, C, rank, seed=500 + 2 * si + i) for i in range(2)]
        xs, bs, qs = [t[0] for t in d], [t[1] for t in d], [t[2] for t in d]
        pk, nb = run(xs, bs, qs, N, C, rank)
        sets.append((xs, bs, qs, [p.clone() for p in pk], [n.clone() for n in nb]))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for si in range(2):
        outs.append(([torch.empty_like(p) for p in sets[si][3]], [torch.empty_like(n) for n in sets[si][4]]))
    torch.cuda.synchronize()
    for it in range(60):
        for si in range(2):
            xs, bs, qs, _, _ = sets[si]
            with torch.cuda.stream(streams[si]):
                K.lr_compress_batch(False, xs, bs, outs[si][1], outs[si][0], qs, N, C, rank, update_cache=True, ef=True, stream=streams[si])
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(K.context(0)) == 0
    for si in range(2):
        for i in range(2):
            assert torch.equal(outs[si][0][i], sets[si][3][i]) and torch.equal(outs[si][1][i], sets[si][4][i])


@pytest.mark.parametrize("B", [3, 4, 5])
def test_several_tensors_in_one_launch(B):
    """C = 1536: 48 workgroups per tensor, up to five tensors co-resident in ONE launch (tensor-major and XCD-interleaved block
    orders); every tensor's bits equal its single-tensor run"""
    N, C, rank = 512, 1536, 8
    data = [make(N, C, rank, seed=700 + i) for i in range(B)]
    xs, bs, qs = [d[0] for d in data], [d[1] for d in data], [d[2] for d in data]
    pk, nb = run(xs, bs, qs, N, C, rank)
    for i in range(B):
        check(xs[i], bs[i], qs[i], pk[i], nb[i], N, C, rank)
        p1, n1 = run([xs[i]], [bs[i]], [qs[i]], N, C, rank)
        assert torch.equal(p1[0], pk[i]) and torch.equal(n1[0], nb[i])


@pytest.mark.parametrize("N,C", [(544, 3072), (512, 1536), (100, 512)])
@pytest.mark.parametrize("rank", [8, 16, 24, 32])
def test_quantised_factors_sender_equals_receiver(N, C, rank):
    """LOW_RANK_Q through the slab-resident launch + the one-launch int4 factor quantiser (k_lr_q4) and dequantiser (k_lr_dq4): the
    sender's state is the receiver's reconstruction bit for bit, and what they add is the rank-r projection up to the int4 noise"""
    from compactfusion_amd import codecs as K
    x, base, q0 = make(N, C, rank, seed=N + rank, decay=0.85)
    pkt = torch.empty(K.lr_packet_halves(True, N, C, rank), dtype=torch.float16, device="cuda")
    nb = torch.empty(N, C, dtype=torch.float16, device="cuda")
    K.lr_compress_batch(True, [x], [base], [nb], [pkt], [q0], N, C, rank, update_cache=True, ef=True)
    rec = torch.empty(N, C, dtype=torch.float16, device="cuda")
    K.lr_decompress_batch(True, [pkt], [base], [rec], N, C, rank)
    torch.cuda.synchronize()
    assert torch.equal(nb, rec), "sender state != receiver reconstruction"
    D = (x - base).float()
    Ur, Vr = subspace_iter_fp64(D, q0[:, :rank])
    want = (Ur @ Vr).float()
    got = nb.float() - base.float()
    assert rel(got, want) < 0.25, "int4 factors: 4 bits a factor entry (15-18 % here; parity with the reference: G8b, G12, G13)"
    # the quantiser is deterministic: same packet bits on a second run
    pkt2 = torch.empty_like(pkt)
    nb2 = torch.empty_like(nb)
    K.lr_compress_batch(True, [x], [base], [nb2], [pkt2], [q0], N, C, rank, update_cache=True, ef=True)
    torch.cuda.synchronize()
    assert torch.equal(pkt.view(torch.int16), pkt2.view(torch.int16)) and torch.equal(nb, nb2)
