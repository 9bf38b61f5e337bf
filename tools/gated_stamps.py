"""Developer probe: phase timeline of ONE gated launch (compress K,V + 16 gated reconstructions), FLUX shard."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from compactfusion_amd import _lib, codecs as K

N, C, L, P = 544, 3072, 8, 14
_lib.use_dev_library()          # per-workgroup stamps exist in libcfx_dev.so only (include/cfx_dev.h)
lib = _lib.load(); ctx = K.context(0)

torch.manual_seed(0)
own = torch.randn(L, 2, N, C, device="cuda").half()
x = (own.float() + 0.1 * torch.randn(L, 2, N, C, device="cuda")).half()
peer = own[:, None].expand(L, 7, 2, N, C).reshape(L, P, N, C).contiguous()
pk = torch.zeros(L, 2, K.packet_halves(1, N, C), dtype=torch.float16, device="cuda")
ws = K.workspace(1, N, C, 0, 2, 0)
sh = torch.cuda.current_stream().cuda_stream
comp, allg = [], []
for l in range(L):
    comp.append((_lib.CompItem * 2)(*[_lib.CompItem(x[l, i].data_ptr(), own[l, i].data_ptr(), None, pk[l, i].data_ptr()) for i in range(2)]))
    e = [_lib.DecompItem(pk[l, i].data_ptr(), own[l, i].data_ptr(), own[l, i].data_ptr()) for i in range(2)]
    pp = [_lib.DecompItem(pk[l, j % 2].data_ptr(), peer[l, j].data_ptr(), peer[l, j].data_ptr()) for j in range(P)]
    allg.append((_lib.DecompItem * (P + 2))(*(e + pp)))
nwg = 4096
st = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
def go(l):
    assert lib.cfx_compress_batch_gated(ctx, 1, N, C, 0, 0, 2, comp[l], 0, None, P + 2, allg[l], ws.data_ptr(), ws.numel(), sh) == 0
for l in range(L): go(l)
torch.cuda.synchronize()
for rep in range(4):
    st.zero_()
    lib.cfx_dev_stamps(ctx, st.data_ptr())
    go(rep % L)
    torch.cuda.synchronize()
    lib.cfx_dev_stamps(ctx, None)
a = st.cpu().numpy().reshape(nwg, 16)
# per own tensor (K = item 0, V = item 1): compress workgroup b belongs to tensor b // (CB * P)
CBP = ((C + 511) // 512) * ((N + 31) // 32)
t00 = a[a[:, 0] > 0][:, 0].min()
for z, nm in ((0, "K"), (1, "V")):
    blk = a[z * CBP:(z + 1) * CBP]
    uj = blk[(blk[:, 7] & 2) != 0]
    vj = blk[(blk[:, 7] & 1) != 0]
    def at(col):                                          # latest stamp of a column, "n/a" where this form of the launch never sets it
        col = col[col > 0]
        return f"{(col.max() - t00) / 100:.2f}" if len(col) else "n/a"
    print(f"tensor {nm}: tiles done p50 {(np.median(blk[:, 1]) - t00) / 100:.2f} max {at(blk[:, 1])}; V jobs end max {at(vj[:, 6])}; "
          f"U job end {at(uj[:, 6])} us (tail loads back {at(uj[:, 4])})")
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
A = a[a[:, 7] != 4]; G = a[a[:, 7] == 4]
print("compress workgroups", len(A), " gated workgroups", len(G))
def show(name, col):
    col = (col[col > 0] - t0) / 100.0
    if len(col): print(f"  {name:50s} min {col.min():6.2f}  p50 {np.median(col):6.2f}  p90 {np.percentile(col, 90):6.2f}  max {col.max():6.2f} us  (n={len(col)})")
for k, nm in enumerate(["start", "tile done", "partials drained + barrier", "tickets drawn", "tail loads back", "V written", "U written / end"]):
    show("[compress] " + nm, A[:, k])
for k, nm in enumerate(["start", "state tile in registers", "gate seen open", "bits + scales landed", "stores drained (end)"]):
    show("[gated] " + nm, G[:, k])
print("gate errors", lib.cfx_gate_errors(ctx))
gs = np.sort((G[:, 0] - t0) / 100.0)
print("gated starts (us), every 16th:", " ".join(f"{v:.1f}" for v in gs[::16]))
print("gated started before 3 us:", int((gs < 3).sum()), " before 12 us:", int((gs < 12).sum()))
ae = np.sort((A[:, 3] - t0) / 100.0)
print("compress tickets drawn, every 12th:", " ".join(f"{v:.1f}" for v in ae[::12]))
