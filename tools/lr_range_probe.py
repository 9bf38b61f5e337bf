"""Developer probe: low-rank compress at residual magnitudes from 1e-2 to ~100 on every chain (slab-resident, six-launch, C-space);
prints the relative error of the projection against an fp64 replay of the reference's iteration."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compactfusion_amd import codecs as K

def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())

def replay(D, Q0):
    A, Q = D.double(), Q0.double()
    for _ in range(2): Q, _ = torch.linalg.qr(A.t() @ (A @ Q))
    U, _ = torch.linalg.qr(A @ Q)
    return U @ (U.t() @ A)

for (N, C) in ((544, 3072), (4096, 1152), (1024, 1152), (64, 256), (4448, 3072)):
    for rank in (8, 16, 32):
        for amp in (0.02, 1.0, 40.0, 300.0):
            g = torch.Generator().manual_seed(rank)
            k = 48
            L = torch.linalg.qr(torch.randn(N, k, generator=g))[0]
            R = torch.linalg.qr(torch.randn(C, k, generator=g))[0]
            s = (0.7 if rank <= 16 else 0.85) ** torch.arange(k, dtype=torch.float32)
            D = ((L * s) @ R.t() * (N * C) ** 0.5 * 0.05 + 1e-3 * torch.randn(N, C, generator=g)) * amp
            x = D.half().cuda()
            q0 = torch.zeros(C, K.lr_rank_pad(rank)); q0[:, :rank] = torch.linalg.qr(torch.randn(C, rank, generator=g))[0]; q0 = q0.cuda()
            pk = torch.empty(K.lr_packet_halves(False, N, C, rank), dtype=torch.float16, device="cuda")
            nb = torch.empty(N, C, dtype=torch.float16, device="cuda")
            K.lr_compress_batch(False, [x], [None], [nb], [pk], [q0], N, C, rank, update_cache=True, ef=True)
            torch.cuda.synchronize()
            U, V = pk[:N * rank].view(N, rank).float(), pk[N * rank:].view(rank, C).float()
            print(f"({N},{C}) r={rank} amp={amp:g} max|x|={float(x.abs().max()):.3g} finite={bool(torch.isfinite(pk.float()).all())} "
                  f"rel={rel(U @ V, replay(x.float(), q0[:, :rank])):.2e}", flush=True)
