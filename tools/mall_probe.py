"""Developer probe: does reading the peers' states ahead of the reconstruction launch (into the 256 MB Infinity Cache)
shorten it?  cold = rotate over 57 layers; hot = same layer; pref = a read pass over the layer's states right before."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compactfusion_amd import _lib, codecs as K

N, C, P, L = 544, 3072, 14, 57
lib = _lib.load(); ctx = K.context(0)
torch.manual_seed(0)
base = (torch.randn(L, P, N, C, device="cuda") * 0.5).half()
pk = torch.randn(L, P, K.packet_halves(1, N, C), device="cuda").half()
sh = torch.cuda.current_stream().cuda_stream
items = []
for l in range(L):
    items.append((_lib.DecompItem * P)(*[_lib.DecompItem(pk[l, i].data_ptr(), base[l, i].data_ptr(), base[l, i].data_ptr()) for i in range(P)]))


def B(l):
    assert lib.cfx_decompress_batch(ctx, 1, N, C, 0, P, items[l], sh) == 0


def timeit(pre, n=171):
    tot = 0.0
    evs = []
    for r in range(n):
        l = pre(r)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); B(l); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2], ts[len(ts) // 10]


iv = base.view(torch.int32)
print("cold  (median, p10) us:", timeit(lambda r: r % L))
print("hot   (median, p10) us:", timeit(lambda r: 0))
def pref(r):
    l = r % L
    iv[l].sum()
    return l
print("pref  (median, p10) us:", timeit(pref))
def pref_half(r):
    l = r % L
    iv[l, :7].sum()
    return l
print("pref7 (median, p10) us:", timeit(pref_half))
