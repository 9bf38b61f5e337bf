#!/usr/bin/env python3
"""Developer probe: phase timeline of ONE int4 / int8 layer launch (k_minmax_layer; loop-back form: B own tensors + NP reconstructions) from
the per-workgroup wall-clock stamps (cfx_dev_stamps).  N, C, B, NP, CODEC from the environment."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from compactfusion_amd import _lib, codecs as K
N, C, B, NP, CID, L = (int(os.environ.get(k, d)) for k, d in (("N", 1024), ("C", 1152), ("B", 2), ("NP", 2), ("CODEC", 3), ("L", 8)))
_lib.use_dev_library()          # per-workgroup stamps exist in libcfx_dev.so only (include/cfx_dev.h)
lib, ctx = _lib.load(), K.context(0)
g = torch.Generator(device="cuda").manual_seed(0)
own = torch.randn(L, B, N, C, device="cuda", generator=g).half()
x = (own.float() + 0.1 * torch.randn(L, B, N, C, device="cuda", generator=g)).half()
peer = own[:, [j % B for j in range(max(NP, 1))]].clone()
pk = torch.zeros(L, B, K.packet_halves(CID, N, C), dtype=torch.float16, device="cuda")
ws = K.workspace(CID, N, C, 0, B, 0)
main = torch.cuda.Stream()
sh = main.cuda_stream
comp = [(_lib.CompItem * B)(*[_lib.CompItem(x[l, i].data_ptr(), own[l, i].data_ptr(), own[l, i].data_ptr(), pk[l, i].data_ptr()) for i in range(B)]) for l in range(L)]
rec = [(_lib.DecompItem * max(NP, 1))(*[_lib.DecompItem(pk[l, j % B].data_ptr(), peer[l, j].data_ptr(), peer[l, j].data_ptr()) for j in range(max(NP, 1))]) for l in range(L)]
nwg = 8192
st = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
def go(l):
    assert lib.cfx_compress_batch_gated(ctx, CID, N, C, 0, 1, B, comp[l], 0, None, NP, rec[l], ws.data_ptr(), ws.numel(), sh) == 0, lib.cfx_last_error_string(ctx)
for l in range(L): go(l)
torch.cuda.synchronize()
for rep in range(3):
    st.zero_()
    lib.cfx_dev_stamps(ctx, st.data_ptr())
    go(rep % L)
    torch.cuda.synchronize()
    lib.cfx_dev_stamps(ctx, None)
a = st.cpu().numpy().reshape(nwg, 16)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
S, D = a[a[:, 7] == 1], a[a[:, 7] == 4]
print(f"codec {CID} ({N},{C}) B={B} NP={NP}: S tiles {len(S)}, D tiles {len(D)}, gate errors {lib.cfx_gate_errors(ctx)}")
def show(name, col):
    col = (col[col > 0] - t0) / 100.0
    if len(col): print(f"  {name:44s} min {col.min():6.2f}  p50 {np.median(col):6.2f}  p90 {np.percentile(col, 90):6.2f}  max {col.max():6.2f} us")
for k, nm in enumerate(["start", "tile loaded, partial issued", "scales known", "codes issued", "codes acknowledged, flag issued", "state stores acknowledged"]):
    show("[S] " + nm, S[:, k])
for k, nm in enumerate(["start", "state in registers + gate seen", "codes landed, stores issued", "stores acknowledged"]):
    show("[D] " + nm, D[:, k])
