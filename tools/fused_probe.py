"""Developer probe: time the compress launch variants (run under rocprofv3 --kernel-trace --stats).
usage: python tools/fused_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compactfusion_amd import _lib, codecs as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, C, B, L = 544, 3072, 2, 24
_lib.use_dev_library()          # per-workgroup stamps exist in libcfx_dev.so only (include/cfx_dev.h)
lib = _lib.load(); ctx = K.context(0)
torch.manual_seed(0)
base = torch.randn(L, B, N, C, device="cuda").half()
x = (base.float() + 0.1 * torch.randn(L, B, N, C, device="cuda")).half()
pk = torch.zeros(L, B, K.packet_halves(1, N, C), dtype=torch.float16, device="cuda")
ws = K.workspace(1, N, C, 0, B, 0)
sh = torch.cuda.current_stream().cuda_stream
items = []
for l in range(L):
    items.append((_lib.CompItem * B)(*[_lib.CompItem(x[l, i].data_ptr(), base[l, i].data_ptr(), None, pk[l, i].data_ptr()) for i in range(B)]))
dbg = int(os.environ.get("DBG", "0"))     # developer build (python -m compactfusion_amd.build --dev-probes): early exits 1..4 of the compress kernel
if dbg:
    assert lib.cfx_dev_set_probe(ctx, dbg) == 0, "DBG needs a --dev-probes build"
for fused in ([1, 0] if not dbg else [1]):
    lib.cfx_set_fused_finalize(ctx, fused)
    for r in range(reps):
        l = r % L
        assert lib.cfx_compress_batch_ex(ctx, 1, N, C, 0, 0, B, items[l], 0, None, ws.data_ptr(), ws.numel(), sh) == 0
    torch.cuda.synchronize()
print("done")
