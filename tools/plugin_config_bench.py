#!/usr/bin/env python3
"""SURVEY.md section 8d protocol 1 THROUGH THE PLUGIN API for EVERY BASELINE configuration and every shipped preset (task 4).

tools/config_table.py / bench.py's `configs` time each configuration through a native plan replay - one host call per step, which no model
issues.  This tool issues the same steps the way a model does: one Python call per layer,
  ring configurations     compact_fwd (reference ring.py:36-70, 188-206), attention replaced by a no-op, lane off (the one-op exchange)
  gather configurations   compact_all_gather_kv (what patch_gather_fwd calls per layer, reference patchpara/fwd.py:88-102)
  configuration 1         compact_compress + compact_decompress on one tensor (world size 1: no exchange)
on ONE GPU with the logical ranks looped back, and reports per row: ms per step back to back, the host's issue time per layer into an
IDLE queue (one step issued after a synchronise: the host cost proper - in the back-to-back loop the host runs ahead until the hardware
queue is full and its time then contains the waiting), and the kernels one step issued.
  rows      1, 2, 3, 4, 5-topk, 5-lr8, 5-lr16   (BASELINE.json configs, shapes of SURVEY 8d / tools/config_table.py)
  presets   binary, int2, lowrank8, lowrank16, lowrankq32 at the FLUX shard (reference examples/configs.py:39-98)
Run on the GPU box:  python tools/plugin_config_bench.py [--rows 1,2,...] [--json out.json]"""
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

import compactfusion_amd
from compactfusion_amd import _lib, codecs as K
from compactfusion_amd.compact import ring, main as cm, xlayer, lowrank
compactfusion_amd.configure(lane="off")
from compactfusion_amd.compact.utils import CompactConfig, COMPACT_COMPRESS_TYPE as T
from compactfusion_amd.collector import collector
from compactfusion_amd.prof import Profiler

# name: (api, world, layers, N, heads, head_dim, compress type, config kwargs)
ROWS = {
    "1": ("codec", 1, 1, 4096, 16, 72, "INT8", dict(comp_rank=-1, fastpath=False)),
    "2": ("gather", 2, 28, 1024, 16, 72, "INT4", dict(comp_rank=-1, fastpath=False)),
    "3": ("ring", 8, 57, 544, 24, 128, "BINARY", dict(comp_rank=-1, fastpath=True)),
    "4": ("ring", 4, 42, 4448, 24, 128, "INT4", dict(comp_rank=-1, fastpath=False)),
    "5 top-k 1:8": ("gather", 8, 24, 512, 24, 64, "SPARSE", dict(comp_rank=-1, fastpath=False, sparse_ratio=8)),
    "5 LOW_RANK r=8": ("gather", 8, 24, 512, 24, 64, "LOW_RANK", dict(comp_rank=8, fastpath=False)),
    "5 LOW_RANK r=16": ("gather", 8, 24, 512, 24, 64, "LOW_RANK", dict(comp_rank=16, fastpath=False)),
    # the shipped presets at the FLUX shard through the ring forward
    "preset binary": ("ring", 8, 57, 544, 24, 128, "BINARY", dict(comp_rank=-1, fastpath=True)),
    "preset int2": ("ring", 8, 57, 544, 24, 128, "INT2", dict(comp_rank=-1, fastpath=True)),
    "preset lowrank8": ("ring", 8, 57, 544, 24, 128, "LOW_RANK", dict(comp_rank=8, fastpath=False)),
    "preset lowrank16": ("ring", 8, 57, 544, 24, 128, "LOW_RANK", dict(comp_rank=16, fastpath=False)),
    "preset lowrankq32": ("ring", 8, 57, 544, 24, 128, "LOW_RANK_Q", dict(comp_rank=32, fastpath=False)),
}

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default=",".join(ROWS))
ap.add_argument("--budget", type=float, default=0.25, help="seconds of back-to-back steps per row (at least 6 steps)")
ap.add_argument("--json", default=None)
ap.add_argument("--quiet", action="store_true")
ap.add_argument("--pin-start", action="store_true", help="low-rank rows: one pinned start matrix instead of a fresh draw per execution (what a model runs)")
args = ap.parse_args()

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
lib, ctx = _lib.load(), K.context(0)
Profiler.instance().disable()
collector.init(collector.Collector("/tmp/none", enabled=False))
_WORLD = [1]
ring.dist.get_rank = lambda g=None: 0
ring.dist.get_world_size = lambda g=None: _WORLD[0]
ring.dist.all_gather_into_tensor = lambda recv, send, group=None: recv.view(_WORLD[0], -1).copy_(send.view(1, -1).expand(_WORLD[0], -1))
cm.dist.get_rank = ring.dist.get_rank
cm.dist.get_world_size = ring.dist.get_world_size
cm.dist.all_gather_into_tensor = ring.dist.all_gather_into_tensor
xlayer.set_p2p_loopback(True)


def drop_state():
    cm._drop_kv_exchanges()
    for e in ring._xbuf.values():
        e.close()
    ring._xbuf.clear()
    ring._steady.clear()
    xlayer.release()
    xlayer.set_p2p_loopback(True)
    torch.cuda.empty_cache()


def one_row(name):
    api, W, L, N, H, D, tname, kw = ROWS[name]
    _WORLD[0] = W
    CT = T[tname]
    drop_state()
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else CT, residual=1, ef=True, **kw))
    g = torch.Generator(device=dev).manual_seed(1)
    # distinct layers' worth of K,V past the Infinity Cache are the states the ops keep; the inputs alternate between two drifts
    k0 = [torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)]
    v0 = [torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)]
    ks = [[(k0[l] + 0.1 * torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g)) for l in range(L)] for _ in range(2)]
    vs = [[(v0[l] + 0.1 * torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g)) for l in range(L)] for _ in range(2)]
    q0 = torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g)
    out_ = torch.zeros(1, N, H, D, device=dev, dtype=torch.float16)
    lse_ = torch.zeros(1, N, H, 1, device=dev, dtype=torch.float32)
    ring.block_attention = lambda q, k, v, *a, **kw_: (out_, lse_)
    ring.update_out_and_lse = lambda out, lse, bo, bl, wait=None: (out_, lse_)
    ring._SteadyLayer._fast_ok = lambda self, q: False
    if kw.get("comp_rank", -1) > 0 and args.pin_start:
        lowrank.set_init_q(torch.randn(H * D, kw["comp_rank"], generator=torch.Generator().manual_seed(3)))
    torch.cuda.synchronize()

    def step(i):
        cm.compact_set_step(i)
        ct = T.WARMUP if i == 0 else CT
        ki, vi = ks[i & 1], vs[i & 1]
        if api == "ring":
            for l in range(L):
                ring.compact_fwd(q0, ki[l], vi[l], causal=False, mod_idx=l, current_iter=i)
        elif api == "gather":
            for l in range(L):
                cm.compact_all_gather_kv(f"{l}-k", f"{l}-v", ki[l], vi[l], ct, group=None)
        else:
            for l in range(L):
                pk = cm.compact_compress(f"{l}-0-k", ki[l], ct, update_cache=True)
                cm.compact_decompress(f"{l}-1-k", pk, ct, (1, N, H, D), update_cache=True)
    stream = torch.cuda.Stream(dev)
    try:
        with torch.cuda.stream(stream):
            for i in range(5):
                step(i)
            torch.cuda.synchronize()
            assert lib.cfx_profile_enable(ctx, 8192, 0xffffffff, 1) == 0
            step(5)
            torch.cuda.synchronize()
            ids, ms = (ctypes.c_int * 8192)(), (ctypes.c_float * 8192)()
            n = lib.cfx_profile_read(ctx, ids, ms, 8192)
            lib.cfx_profile_enable(ctx, 0, 0, 1)
            kern = {}
            for j in range(n):
                nm = lib.cfx_kernel_name(ids[j]).decode().split(" (")[0]
                kern[nm] = kern.get(nm, 0) + 1
            step(6); step(7)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = 0
            while steps < 6 or time.perf_counter() - t0 < args.budget:
                step(8 + steps)
                steps += 1
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps
            idle = []
            for i in range(7):
                torch.cuda.synchronize()
                h0 = time.perf_counter()
                step(8 + steps + i)
                idle.append(time.perf_counter() - h0)
            torch.cuda.synchronize()
            idle.sort()
    finally:
        lowrank.set_init_q(None)
    assert lib.cfx_gate_errors(ctx) == 0
    ops = [e.xop for e in ring._xbuf.values() if e.xop is not None] + [e.xop for e in cm._kv_exchanges.values() if e.xop is not None]
    row = {"api": {"ring": "compact_fwd (no-op attention, lane off)", "gather": "compact_all_gather_kv", "codec": "compact_compress + compact_decompress"}[api],
           "world": W, "layers": L, "shard": [N, H * D], "type": tname, "comp_rank": kw.get("comp_rank", -1),
           "plugin_ms_per_step": round(wall * 1e3, 4), "host_us_per_layer_idle_queue": round(idle[len(idle) // 2] / L * 1e6, 2),
           "steps": steps, "kernels_per_layer": {k_: round(v_ / L, 2) for k_, v_ in kern.items()},
           "one_native_op_per_layer": bool(ops) and len(ops) == L, "transport": sorted({o.transport for o in ops})}
    del k0, v0, ks, vs
    return row


res = {"what": "protocol 1 through the plugin API, one Python call per layer, logical ranks looped back on one GPU; host_us_per_layer_idle_queue = "
               "one step issued into an idle queue / layers (the host cost proper)", "rows": {}}
for name in [r.strip() for r in args.rows.split(",") if r.strip()]:
    try:
        res["rows"][name] = one_row(name)
    except Exception as e:  # noqa: BLE001  (a row that cannot run says so; the others still report)
        res["rows"][name] = {"error": f"{type(e).__name__}: {e}"}
    if not args.quiet:
        print(name, json.dumps(res["rows"][name]), flush=True)
drop_state()
if args.json:
    json.dump(res, open(args.json, "w"), indent=1)
if args.quiet:
    print(json.dumps(res))
