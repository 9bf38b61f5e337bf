import torch, sys
sys.path.insert(0,'/root/repo')
from compactfusion_amd.compact.lowrank import slowpath_compress, slowpath_decompress
from compactfusion_amd.compact.utils import COMPACT_COMPRESS_TYPE as T
import compactfusion_amd.compact.lowrank as LR
def rel(a,b): return float((a.float()-b.float()).norm()/b.float().norm())
dev='cuda'
for (N,C) in ((64,256),(128,3072),(544,3072),(512,1536)):
    g = torch.Generator().manual_seed(9)
    low = (torch.randn(N, 3, generator=g) @ torch.randn(3, C, generator=g)).half().to(dev)
    for r in (8,16):
        pkt = slowpath_compress(low, T.LOW_RANK, rank=r)
        dec = slowpath_decompress(pkt, (N, C), T.LOW_RANK, rank=r)
        U = pkt[:N*r].view(N,r).float()
        print(N,C,r,"finite",bool(torch.isfinite(dec.float()).all()),"rel",rel(dec,low),"orth err",float((U.t()@U-torch.eye(r,device=dev)).abs().max()))
    full = torch.randn(N, C, generator=g).half().to(dev)
    q0 = torch.randn(C, 8, generator=torch.Generator().manual_seed(5)); LR.set_init_q(q0)
    pkt = slowpath_compress(full, T.LOW_RANK, rank=8); LR.set_init_q(None)
    U = pkt[:N*8].view(N,8).float(); V = pkt[N*8:].view(8,C).float()
    # reference in torch
    A=full.float(); Q=q0.to(dev)
    for _ in range(2): Q,_=torch.linalg.qr(A.t()@(A@Q))
    Ur,_=torch.linalg.qr(A@Q); Vr=Ur.t()@A
    print("  full-rank: rel diff of projection vs torch subspace_iter", rel(U@V, Ur@Vr), "orth err", float((U.t()@U-torch.eye(8,device=dev)).abs().max()))
