"""Developer probe: phase timeline of the single-launch compress kernel (cfx_dev_stamps), FLUX shard, K and V.
Prints, over the workgroups of one launch, when each phase ends relative to the first workgroup's start (us)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from compactfusion_amd import _lib, codecs as K

N, C, B, L = 544, 3072, 2, 16
_lib.use_dev_library()          # per-workgroup stamps exist in libcfx_dev.so only (include/cfx_dev.h)
lib = _lib.load(); ctx = K.context(0)
torch.manual_seed(0)
base = torch.randn(L, B, N, C, device="cuda").half()
x = (base.float() + 0.1 * torch.randn(L, B, N, C, device="cuda")).half()
pk = torch.zeros(L, B, K.packet_halves(1, N, C), dtype=torch.float16, device="cuda")
ws = K.workspace(1, N, C, 0, B, 0)
sh = torch.cuda.current_stream().cuda_stream
nwg = 4096
st = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
items = [(_lib.CompItem * B)(*[_lib.CompItem(x[l, i].data_ptr(), base[l, i].data_ptr(), None, pk[l, i].data_ptr()) for i in range(B)]) for l in range(L)]
for l in range(L):   # warm
    lib.cfx_compress_batch_ex(ctx, 1, N, C, 0, 0, B, items[l], 0, None, ws.data_ptr(), ws.numel(), sh)
torch.cuda.synchronize()
names = ["start", "tile done (loads+math+partial stores issued)", "partials drained + barrier", "tickets drawn", "tail loads back", "V written", "U written"]
agg = []
for rep in range(8):
    st.zero_()
    lib.cfx_dev_stamps(ctx, st.data_ptr())
    lib.cfx_compress_batch_ex(ctx, 1, N, C, 0, 0, B, items[rep % L], 0, None, ws.data_ptr(), ws.numel(), sh)
    torch.cuda.synchronize()
    lib.cfx_dev_stamps(ctx, None)
    a = st.cpu().numpy().reshape(nwg, 16)
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    rel = (a[:, :7] - t0) / 100.0      # us
    roles = a[:, 7]
    row = [len(a)]
    for k in range(7):
        col = rel[:, k][a[:, k] > 0]
        row += [col.min(), np.median(col), col.max()]
    lc = rel[roles & 1 > 0]; la = rel[roles & 2 > 0]
    agg.append((row, lc, la))
row, lc, la = agg[-1]
print("workgroups", row[0])
for k in range(7):
    print(f"  {names[k]:55s} min {row[1+3*k]:6.2f}  median {row[2+3*k]:6.2f}  max {row[3+3*k]:6.2f} us")
inner = ["loads landed (wave 0)", "math done", "row butterfly done", "col sums in LDS", "barrier passed"]
for k in range(5):
    col = (a[:, 8 + k] - t0) / 100.0
    print(f"  [stats] {inner[k]:47s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")
print("last-of-column-block workgroups (per phase, us):")
for r in lc: print("   ", " ".join(f"{v:6.2f}" for v in r))
print("last-of-tensor workgroups:")
for r in la: print("   ", " ".join(f"{v:6.2f}" for v in r))
sel = a[roles & 2 > 0]
for r in sel: print("    U phase: acc ready %.2f  wave sum %.2f  barrier %.2f  U written %.2f" % tuple((r[k] - t0) / 100.0 for k in (13, 14, 15, 6)))
