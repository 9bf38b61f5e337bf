#!/usr/bin/env python3
"""Developer probe: the min/max layer launch at a tall shard (BASELINE config 4: (4448, 3072), 2 own + 6 peer tensors) - per-launch kernel ids and
durations of the loop-back gated form and of the exchange-layer op, with the layer launch on and off (cfx_set_gated_launch)."""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compactfusion_amd import _lib, codecs as K
lib, ctx = _lib.load(), K.context(0)
dev = torch.device("cuda:0")
N, C = int(os.environ.get("N", 4448)), int(os.environ.get("C", 3072))
CID, B, NP, L = int(os.environ.get("CODEC", 3)), int(os.environ.get("B", 2)), int(os.environ.get("NP", 6)), int(os.environ.get("L", 8))
PRM = int(os.environ.get("PARAM", 8 if CID == 5 else 0))
g = torch.Generator(device=dev).manual_seed(1)
x = [torch.randn(L, B, N, C, generator=g, device=dev).half() for _ in range(2)]
own = torch.randn(L, B, N, C, generator=g, device=dev).half()
peer = own[:, [j % B for j in range(NP)]].clone()
ph = K.packet_halves(CID, N, C, PRM)
pk = torch.zeros(L, B, ph, dtype=torch.float16, device=dev)
ws = K.workspace(CID, N, C, PRM, B, 0)
wsp, wsn = (None, 0) if ws is None else (ws.data_ptr(), ws.numel())
main = torch.cuda.Stream(dev)
sh = main.cuda_stream
def items(s, l):
    c = (_lib.CompItem * B)(*[_lib.CompItem(x[s][l, i].data_ptr(), own[l, i].data_ptr(), own[l, i].data_ptr(), pk[l, i].data_ptr()) for i in range(B)])
    d = (_lib.DecompItem * NP)(*[_lib.DecompItem(pk[l, j % B].data_ptr(), peer[l, j].data_ptr(), peer[l, j].data_ptr()) for j in range(NP)])
    return c, d
its = [[items(s, l) for l in range(L)] for s in range(2)]
def step(i):
    for l in range(L):
        c, d = its[i & 1][l]
        assert lib.cfx_compress_batch_gated(ctx, CID, N, C, PRM, 1, B, c, 0, None, NP, d, wsp, wsn, sh) == 0, lib.cfx_last_error_string(ctx)
for on in [1, 0, 1] + [int(v) + 100 for v in os.environ.get("STAGGER", "").split(",") if v]:
    if on >= 100:
        assert lib.cfx_set_tall_stagger(ctx, on - 100) == 0
        print(f"stagger {on - 100}: ", end="")
        on = 1
    assert lib.cfx_set_gated_launch(ctx, on) == 0
    for i in range(3): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 / L * 1e6
    lib.cfx_profile_enable(ctx, 4096, 0xffffffff, 1)
    step(0); torch.cuda.synchronize()
    ids = (ctypes.c_int * 4096)(); ms = (ctypes.c_float * 4096)()
    k = lib.cfx_profile_read(ctx, ids, ms, 4096)
    lib.cfx_profile_enable(ctx, 0, 0, 1)
    agg = {}
    for i in range(k): agg.setdefault(ids[i], []).append(ms[i] * 1e3)
    alg = (B * 3 + NP * 2) * N * C * 2           # (fp16 activation bytes; the packets come on top)
    print(f"layer launch {'on ' if on else 'off'}: {dt:.1f} us/layer  ({alg / dt / 1e6:.2f} TB/s algorithmic)  gate errors {lib.cfx_gate_errors(ctx)}  "
          f"{ {a: (round(sum(v) / len(v), 1), len(v)) for a, v in agg.items()} }", flush=True)
