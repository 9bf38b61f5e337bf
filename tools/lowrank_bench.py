#!/usr/bin/env python3
"""Developer timing of the native low-rank codec chain on the GPU box."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compactfusion_amd import _lib, codecs as K

lib = _lib.load()
ctx = K.context(0)
if os.environ.get("LR_CHAIN"):          # 1 = never the single launch (six-launch N-space chain), 2 = C-space chain only
    assert lib.cfx_set_lr_chain(ctx, int(os.environ["LR_CHAIN"])) == 0

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

def t_host(fn, n=30):
    """host time of one call (enqueue only): a chain whose launches take longer to issue than to run is host-bound"""
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize(); return dt

def prof(fn, n=10):
    lib.cfx_profile_enable(ctx, 4096, 0xffffffff, 1)
    for _ in range(n): fn()
    torch.cuda.synchronize()
    ids = (ctypes.c_int * 4096)(); ms = (ctypes.c_float * 4096)()
    k = lib.cfx_profile_read(ctx, ids, ms, 4096)
    lib.cfx_profile_enable(ctx, 0, 0, 1)
    if os.environ.get("LR_SEQ"):        # the launches of one call, in order (average over the n calls)
        per = k // n
        print("   seq:", " ".join(f"{lib.cfx_kernel_name(ids[j]).decode().split()[0]}={sum(ms[c * per + j] for c in range(n)) / n * 1e3:.1f}" for j in range(per)))
    agg = {}
    for i in range(k): agg.setdefault(lib.cfx_kernel_name(ids[i]).decode(), []).append(ms[i] * 1e3)
    return {a: (round(sum(v) / len(v), 2), len(v) // n) for a, v in agg.items()}

for (N, C) in [(544, 3072), (512, 1536), (4096, 1152), (1024, 1152)]:
    for B in (2,):
        xs = [torch.randn(N, C, device="cuda").half() for _ in range(B)]
        bs = [(x.float() + 0.1 * torch.randn(N, C, device="cuda")).half() for x in xs]
        for q, r in ((False, 8), (False, 16), (True, 32)):
            pk = [torch.empty(K.lr_packet_halves(q, N, C, r), dtype=torch.float16, device="cuda") for _ in range(B)]
            q0 = [torch.randn(C, K.lr_rank_pad(r), device="cuda") for _ in range(B)]
            nb = [b.clone() for b in bs]
            f = lambda: K.lr_compress_batch(q, xs, bs, nb, pk, q0, N, C, r, True)
            out = [torch.empty_like(x) for x in xs] * 7
            g = lambda: K.lr_decompress_batch(q, (pk * 7)[:14], (bs * 7)[:14], out[:14], N, C, r)
            print(f"({N},{C}) q={int(q)} r={r} batch {B}: compress {t(f):7.1f} us (host {t_host(f):5.1f}) | decompress(14) {t(g):7.1f} us | {prof(f)} | {prof(g)}", flush=True)
