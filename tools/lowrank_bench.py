#!/usr/bin/env python3
"""Developer timing of the low-rank codec path on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compactfusion_amd.compact import lowrank as LR

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for (N, C) in [(544, 3072), (512, 1536), (4096, 1152)]:
    x = torch.randn(N, C, device="cuda").half(); b = (x.float() + 0.1 * torch.randn(N, C, device="cuda")).half()
    d = (x - b)
    for cid, r in ((LR.LOW_RANK_ID, 8), (LR.LOW_RANK_ID, 16), (LR.LOW_RANK_Q_ID, 32)):
        pkt = torch.empty(LR.packet_halves(cid, r, N, C), dtype=torch.float16, device="cuda")
        out = torch.empty_like(x)
        Af = d.float(); Q = torch.linalg.qr(torch.randn(C, r, device="cuda"))[0]
        print(f"({N},{C}) cid {cid} r={r}: compress {t(lambda: LR.compress(cid, r, x, b, b.clone(), pkt, True)):8.1f} us | "
              f"decompress {t(lambda: LR.decompress(cid, r, pkt, b, out)):7.1f} us | subspace_iter {t(lambda: LR.subspace_iter(d, r, 2)):8.1f} us | "
              f"qr(C,r) {t(lambda: torch.linalg.qr(Af.t() @ (Af @ Q))):7.1f} us | A@Q {t(lambda: Af @ Q):6.1f} us | At@Y {t(lambda: Af.t() @ (Af @ Q)):6.1f} us")
