"""Developer probe: host + device cost of ONE denoise step through the Python API mirror (`compact_fwd`, gather
schedule) at the FLUX shape, with the collective looped back in-process and attention stubbed out, next to the native
plan replay bench.py measures.  Run on the GPU box:  python tools/api_overhead.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from compactfusion_amd.compact import ring, main as cm
from compactfusion_amd.compact.utils import CompactConfig, COMPACT_COMPRESS_TYPE as T
from compactfusion_amd.collector import collector
from compactfusion_amd.prof import Profiler

W, L, N, H, D = 8, 57, 544, 24, 128
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)

# in-process loopback of the 8-rank group
ring.dist.get_rank = lambda g=None: 0
ring.dist.get_world_size = lambda g=None: W
def _ag(recv, send, group=None):
    recv.view(W, -1).copy_(send.view(1, -1).expand(W, -1))
ring.dist.all_gather_into_tensor = _ag
zero_o = torch.zeros(1, N, H, D, device=dev, dtype=torch.float32)
zero_l = torch.zeros(1, H, N, 1, device=dev, dtype=torch.float32)
ring.block_attention = lambda q, k, v, *a, **kw: (zero_o, zero_l)
ring.update_out_and_lse = lambda out, lse, bo, bl: (bo, bl)

Profiler.instance().disable()
collector.init(collector.Collector("/tmp/none", enabled=False))
cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T.BINARY, comp_rank=-1,
                              residual=1, ef=True, fastpath=True))
g = torch.Generator(device=dev).manual_seed(1)
ks = [torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)]
vs = [torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)]
q = ks[0]

def step(i):
    cm.compact_set_step(i)
    for l in range(L):
        ring.compact_fwd(q, ks[l], vs[l], causal=False, mod_idx=l, current_iter=i)

step(0); step(1); torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for i in range(2, 2 + K):
    step(i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"API path: host issue {t_host / K * 1e3:.2f} ms/step, wall {t_all / K * 1e3:.2f} ms/step "
      f"({t_all / K / L * 1e6:.1f} us/layer)")

if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(2 + K, 2 + K + 3):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
