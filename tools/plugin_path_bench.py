#!/usr/bin/env python3
"""SURVEY.md §8d protocol 1 THROUGH THE PLUGIN API: the pure exchange step (attention replaced by a no-op) of the bench workload - FLUX.1-dev
1024^2, logical ring of 8, 1-bit residual + error feedback, 57 layers, shard (544, 3072) - issued the way a model issues it:
  compact_all_gather_kv   what `patch_gather_fwd` calls per layer (reference xfuser/compact/main.py:390-420, patchpara/fwd.py:88-102)
  compact_fwd             the ring forward, gather schedule (reference ring.py:188-206 + 265-269), with `block_attention` /
                          `update_out_and_lse` replaced by no-ops
Both go through compact/xlayer.py: ONE native op per layer (cfx_plan_add_exchange_layer_p2p), which for the 1-bit codec is ONE codec launch.
One GPU: the 8 logical ranks are looped back (every logical peer reads this rank's packets from the uncached IPC arena) - the same op,
launches and kernels as any N minus the remote reads and the waiting.  Reported per leg: wall ms/step (back-to-back steps), host issue
us/layer (median of single steps issued into an idle queue; and of the back-to-back loop, where it contains the waiting for queue space),
and the kernel ids one step issued (cfx_profile_read: 31 = the gated layer launch).
Run on the GPU box:  python tools/plugin_path_bench.py [--steps K] [--json out.json]"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--layers", type=int, default=57)
ap.add_argument("--codec", default="BINARY")
ap.add_argument("--json", default=None)
ap.add_argument("--quiet", action="store_true")
args = ap.parse_args()

import compactfusion_amd
from compactfusion_amd import _lib, codecs as K
from compactfusion_amd.compact import ring, main as cm, xlayer
compactfusion_amd.configure(lane="off")       # this tool is about the ONE-op layer exchange on the caller's stream (round 5's default is the lane)
from compactfusion_amd.compact.utils import CompactConfig, COMPACT_COMPRESS_TYPE as T
from compactfusion_amd.collector import collector
from compactfusion_amd.prof import Profiler

W, L, N, H, D = 8, args.layers, 544, 24, 128
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
lib, ctx = _lib.load(), K.context(0)
CT = T[args.codec]

# ---- the 8-rank group, looped back in-process -----------------------------------------------------------------------------------
ring.dist.get_rank = lambda g=None: 0
ring.dist.get_world_size = lambda g=None: W
ring.dist.all_gather_into_tensor = lambda recv, send, group=None: recv.view(W, -1).copy_(send.view(1, -1).expand(W, -1))
xlayer.set_p2p_loopback(True)
Profiler.instance().disable()
collector.init(collector.Collector("/tmp/none", enabled=False))
g = torch.Generator(device=dev).manual_seed(1)
q0 = torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g)
k0 = [torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)]
v0 = [torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)]
drift = [[0.1 * torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)] for _ in range(2)]
ks = [[(k0[l] + drift[s][l]) for l in range(L)] for s in range(2)]
vs = [[(v0[l] - drift[s][l]) for l in range(L)] for s in range(2)]
torch.cuda.synchronize()

# ---- attention replaced by a no-op (protocol 1) ------------------------------------------------------------------------------------
_out = torch.zeros(1, N, H, D, device=dev, dtype=torch.float16)
_lse = torch.zeros(1, N, H, 1, device=dev, dtype=torch.float32)
ring.block_attention = lambda q, k, v, *a, **kw: (_out, _lse)
ring.update_out_and_lse = lambda out, lse, bo, bl, wait=None: (_out, _lse)
ring._SteadyLayer._fast_ok = lambda self, q: False            # (the lean path calls the fused SDPA op directly)


def fwd(i):
    cm.compact_set_step(i)
    ki, vi = ks[i & 1], vs[i & 1]
    for l in range(L):
        ring.compact_fwd(q0, ki[l], vi[l], causal=False, mod_idx=l, current_iter=i)


def gather(i):
    cm.compact_set_step(i)
    ct = T.WARMUP if i == 0 else CT
    ki, vi = ks[i & 1], vs[i & 1]
    for l in range(L):
        cm.compact_all_gather_kv(f"{l}-k", f"{l}-v", ki[l], vi[l], ct, group=None)


def init():
    cm._drop_kv_exchanges()
    for e in ring._xbuf.values():
        e.close()
    ring._xbuf.clear()
    ring._steady.clear()
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else CT, comp_rank=-1, residual=1, ef=True,
                                  fastpath=args.codec in ("BINARY", "INT2"), sparse_ratio=8))


def kernel_ids(fn, i):
    torch.cuda.synchronize()
    assert lib.cfx_profile_enable(ctx, 4096, 0xffffffff, 1) == 0
    fn(i)
    torch.cuda.synchronize()
    ids = (ctypes.c_int * 4096)()
    ms = (ctypes.c_float * 4096)()
    n = lib.cfx_profile_read(ctx, ids, ms, 4096)
    lib.cfx_profile_enable(ctx, 0, 0, 1)
    out = {}
    for j in range(n):
        name = lib.cfx_kernel_name(ids[j]).decode()
        out[f"{ids[j]} {name}"] = out.get(f"{ids[j]} {name}", 0) + 1
    return out


def timed(fn, first, steps):
    fn(first); fn(first + 1)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        h0 = time.perf_counter()
        fn(first + 2 + i)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # host cost proper: ONE step issued into an idle queue (in the back-to-back loop above the host runs ahead of the GPU - 25 us of kernel per
    # layer - until the hardware queue is full, and its issue time then contains the waiting)
    idle = []
    for i in range(9):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        fn(first + 2 + steps + i)
        idle.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    idle.sort()
    return {"ms_per_step": round(wall / steps * 1e3, 4), "host_us_per_layer": round(idle[len(idle) // 2] / L * 1e6, 2),
            "host_us_per_layer_back_to_back": round(host / steps / L * 1e6, 2)}


res = {"workload": f"FLUX.1-dev 1024^2, logical ring 8 looped back on one GPU, {args.codec} residual + error feedback, {L} layers, shard (544,3072), "
                   "attention replaced by a no-op", "steps": args.steps, "legs": {}}
# (the default-stream legs first: the legacy NULL stream serialises with every BLOCKING stream of the process, and the side-stream legs create
# one - the CU-masked exchange stream; a model that runs on the default stream never has it: its exchange stream is a non-blocking one)
for sname, stream in (("default_stream", torch.cuda.default_stream(dev)), ("side_stream", torch.cuda.Stream(dev))):
    with torch.cuda.stream(stream):
        for name, fn in (("compact_all_gather_kv", gather), ("compact_fwd_noop_attention", fwd)):
            init()
            for i in range(4):
                fn(i)
            torch.cuda.synchronize()
            ids = kernel_ids(fn, 4)
            leg = timed(fn, 5, args.steps)
            leg["kernels_of_one_step"] = ids
            ops = [e.xop for e in ring._xbuf.values() if e.xop is not None] + [e.xop for e in cm._kv_exchanges.values() if e.xop is not None]
            leg["transport"] = sorted({o.transport for o in ops})
            leg["one_native_op_per_layer"] = len(ops) == L
            res["legs"][f"{name}/{sname}"] = leg
    assert lib.cfx_gate_errors(ctx) == 0
main = res["legs"]["compact_all_gather_kv/side_stream"]
res["ms_per_step"] = main["ms_per_step"]
res["host_us_per_layer"] = main["host_us_per_layer"]
arena = next(iter(xlayer._arenas.values()), None)
res["ipc_memory_kind"] = None if arena is None else {2: "uncached", 1: "fine-grained", 0: "ordinary"}.get(arena.kind)
if args.json:
    json.dump(res, open(args.json, "w"), indent=1)
if not args.quiet:
    print(json.dumps(res, indent=1))
