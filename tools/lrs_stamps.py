"""Developer probe: stage timeline of the slab-resident low-rank chain (k_lrs, cfx_dev_stamps).  Per stamp: min / median / max over
the workgroups, microseconds after the earliest workgroup's start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from compactfusion_amd import _lib, codecs as K

_lib.use_dev_library()          # per-workgroup stamps exist in libcfx_dev.so only (include/cfx_dev.h)
lib = _lib.load(); ctx = K.context(0)
N, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (544, 3072)
names = ["start", "slab in registers + LDS", "Y0 partial written", "Y0 summed", "W1 + M1 partial written", "W1 + M1 summed", "factor of M1, Y1",
         "W2 partial written", "W2 summed", "P = W2^T W2, its factor", "U", "V + state done", "W1: share summed + published", "P = W2^T W2 formed", "factor of M1"]
for r in (8, 16, 32):
    B = 2
    xs = [torch.randn(N, C, device="cuda").half() for _ in range(B)]
    bs = [(x.float() + 0.1 * torch.randn(N, C, device="cuda")).half() for x in xs]
    pk = [torch.empty(K.lr_packet_halves(False, N, C, r), dtype=torch.float16, device="cuda") for _ in range(B)]
    q0 = [torch.randn(C, K.lr_rank_pad(r), device="cuda") for _ in range(B)]
    nb = [b.clone() for b in bs]
    f = lambda: K.lr_compress_batch(False, xs, bs, nb, pk, q0, N, C, r, True)
    for _ in range(5): f()
    st = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
    lib.cfx_dev_stamps(ctx, st.data_ptr())
    f(); torch.cuda.synchronize()
    lib.cfx_dev_stamps(ctx, None)
    a = st.cpu().numpy().reshape(-1, 16)
    t0 = a[a[:, 0] > 0][:, 0].min()
    a = a[0::2]                                  # tensor 0 (z = workgroup % batch)
    a = a[a[:, 0] > 0]
    a = np.where(a == 0, t0, a)
    print(f"({N},{C}) r={r}: {len(a)} workgroups of tensor 0")
    for k, nm in enumerate(names):
        v = (a[:, k] - t0) / 100.0
        print(f"  {nm:22s} min {v.min():7.2f}  med {np.median(v):7.2f}  max {v.max():7.2f}")
