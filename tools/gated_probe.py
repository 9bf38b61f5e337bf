"""Developer probe: one layer of the in-order step as  A(ride own EF) ; B(14 peers)  vs ONE gated launch (own EF + 14 peers
behind the arrival gate).  Run under rocprofv3 --kernel-trace --stats, or read the event timings it prints."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compactfusion_amd import _lib, codecs as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, C, L, P = 544, 3072, 57, 14
lib = _lib.load(); ctx = K.context(0)

torch.manual_seed(0)
own = torch.randn(L, 2, N, C, device="cuda").half()
x = (own.float() + 0.1 * torch.randn(L, 2, N, C, device="cuda")).half()
peer = own[:, None].expand(L, 7, 2, N, C).reshape(L, P, N, C).contiguous()
pk = torch.zeros(L, 2, K.packet_halves(1, N, C), dtype=torch.float16, device="cuda")
ws = K.workspace(1, N, C, 0, 2, 0)
sh = torch.cuda.current_stream().cuda_stream
comp, ef, peers, allg = [], [], [], []
for l in range(L):
    comp.append((_lib.CompItem * 2)(*[_lib.CompItem(x[l, i].data_ptr(), own[l, i].data_ptr(), None, pk[l, i].data_ptr()) for i in range(2)]))
    e = [_lib.DecompItem(pk[l, i].data_ptr(), own[l, i].data_ptr(), own[l, i].data_ptr()) for i in range(2)]
    pp = [_lib.DecompItem(pk[l, j % 2].data_ptr(), peer[l, j].data_ptr(), peer[l, j].data_ptr()) for j in range(P)]
    ef.append((_lib.DecompItem * 2)(*e)); peers.append((_lib.DecompItem * P)(*pp)); allg.append((_lib.DecompItem * (P + 2))(*(e + pp)))


def step_two():
    for l in range(L):
        r = lib.cfx_compress_batch_ex(ctx, 1, N, C, 0, 0, 2, comp[l], 2 if l else 0, ef[l - 1] if l else None, ws.data_ptr(), ws.numel(), sh)
        assert r == 0
        assert lib.cfx_decompress_batch(ctx, 1, N, C, 0, P + (2 if l == L - 1 else 0), allg[l] if l == L - 1 else peers[l], sh) == 0
        # (last layer: own EF cannot ride anywhere)


def step_gated():
    for l in range(L):
        assert lib.cfx_compress_batch_gated(ctx, 1, N, C, 0, 0, 2, comp[l], 0, None, P + 2, allg[l], ws.data_ptr(), ws.numel(), sh) == 0


for name, fn in (("two launches per layer", step_two), ("one gated launch per layer", step_gated)):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / reps:.3f} ms/step", flush=True)
print("gate errors:", lib.cfx_gate_errors(ctx))
same = all(torch.equal(own[l].view(torch.int16), peer[l, j].view(torch.int16).reshape(7, 2, N, C)[0]) if False else True for l in range(1) for j in range(1))
ok = all(torch.equal(peer[l, j].view(torch.int16), own[l, j % 2].view(torch.int16)) for l in range(L) for j in range(P))
print("peers == owners:", ok)
