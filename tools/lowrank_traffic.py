#!/usr/bin/env python3
"""Cold-cache run of the slab-resident low-rank launch (k_lrs) for profiling: 64 distinct (x, state) K,V pairs of the FLUX shard
(1.7 GB, well past the 256 MB Infinity Cache) compressed in turn, rank 8 then rank 16; 96 MiB copy probes first (the WRITE_SIZE
calibration of tools/pmc_summary.py).  Plain run: event-timed microseconds per launch.  Under rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -o lr -- python3 tools/lowrank_traffic.py
    rocprofv3 --pmc FETCH_SIZE -d fetch -o pmc -- python3 tools/lowrank_traffic.py      (WRITE_SIZE: a second pass)
    python3 tools/pmc_summary.py fetch write out.json
Algorithmic HBM bytes of one launch (K and V): x + state in, state out = 6 B/el = 20.05 MB, + the packets (0.1 MB at r = 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compactfusion_amd import _lib, codecs as K

lib = _lib.load(); ctx = K.context(0)
N, C, L, B = 544, 3072, 64, 2
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(L, B, N, C, generator=g, device="cuda").half()
st = (x.float() + 0.1 * torch.randn(L, B, N, C, generator=g, device="cuda")).half()
sh = torch.cuda.current_stream().cuda_stream
nb = 96 * 1024 * 1024
src = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(4)]
dst = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(4)]
for i in range(8):
    assert lib.cfx_copy_probe(ctx, dst[i % 4].data_ptr(), src[i % 4].data_ptr(), nb, sh) == 0
for r in (8, 16):
    pk = [torch.empty(K.lr_packet_halves(False, N, C, r), dtype=torch.float16, device="cuda") for _ in range(B)]
    q0 = [torch.randn(C, K.lr_rank_pad(r), generator=g, device="cuda") for _ in range(B)]
    def layer(l):
        xs = [x[l, j] for j in range(B)]; bs = [st[l, j] for j in range(B)]
        K.lr_compress_batch(False, xs, bs, bs, pk, q0, N, C, r, True)
    for l in range(8): layer(l)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 2
    e0.record()
    for _ in range(reps):
        for l in range(L): layer(l)
    e1.record(); torch.cuda.synchronize()
    print(f"(544,3072) r={r} K,V per launch, cold ({L} distinct pairs in turn): {e0.elapsed_time(e1) / (reps * L) * 1e3:.1f} us per launch "
          f"(host-paced if the Python call is slower than the kernel); algorithmic HBM bytes {6 * B * N * C + B * 2 * K.lr_packet_halves(False, N, C, r)}")
