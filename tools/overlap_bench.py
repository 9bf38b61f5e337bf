"""SURVEY.md §8d protocol 2: the deployable path (`compact_fwd`, gather schedule) with REAL attention, one GPU.

FLUX.1-dev shape as one rank of a ring of 8 sees it: q/k/v (1, 544, 24, 128) fp16, 57 layers, 1-bit residual codec.  The 8
logical ranks are looped back: every peer's packet is this rank's own packet (tests/fake_rccl in loopback mode stands in for
RCCL - same stream-ordered all-gather, device copies instead of xGMI; with real peers the collective's wire time adds to what
must hide under the local attention block).  Legs, all on the same inputs:
  attention      the 8 attention blocks + merges of every layer on resident K,V (no exchange at all)
  default        what compact_fwd does with NO user opt-in (round 5): called on an ordinary stream, it puts itself on the exchange lane for the
                 duration of the call - flag-kernel fork to the lane's compute stream, the lane leg's schedule, flag-kernel join back
                 (compactfusion_amd.configure(lane="auto"), compactfusion_amd/lanes.py)
  sticky         configure(lane="sticky"): the first call makes the lane's compute stream the caller's current stream and leaves it there
  lane           compact_fwd on the EXCHANGE LANE, the caller already on it: the model on the lane's CU-masked compute stream (224 CUs), the layer's
                 whole chain - compress, all-gather, per-peer reconstruction - on the CU-masked exchange stream (32 CUs), ordered only
                 by flags in device memory (cfx_plan_run_lane + cfx_attn_merge_wait: one host call per layer for the exchange)
  layer_op       configure(lane="off") (round 4's default): the layer's exchange as ONE native op (compact/xlayer.py: one codec launch gated on the
                 packets' arrival) on the model's own stream, in front of the local attention block - nothing overlaps, one launch per layer
  lane_unmasked  the same flags, but the model on an ordinary stream and the chain on an unmasked exchange stream
  native         round 2's schedule: the chain on the exchange stream, forked and joined with EVENTS (cfx_plan_run_async + cfx_plan_join)
  native_gather_only_on_side   as round 1 scheduled it: compress and reconstruction on the compute stream, only the collective beside
                 the local block
  torchdist      compact_fwd with the collective issued from Python (what round 1 did; the loop-back copy stands in for
                 torch.distributed.all_gather_into_tensor, whose ~50 us of host time per call is NOT included here)
Reported: wall ms/step, host issue ms/step, exposed exchange = leg - attention, host us/layer (by difference of legs: noisy;
and the native exchange's own host calls measured directly).
Run on the GPU box:  python tools/overlap_bench.py [--steps K] [--json out.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--layers", type=int, default=57)
ap.add_argument("--json", default=None)
ap.add_argument("--quick", action="store_true", help="only the attention, lane and event-fork legs")
ap.add_argument("--quiet", action="store_true", help="do not print the JSON (bench.py runs this file as a child process and reads --json)")
ap.add_argument("--legs", default=None, help="comma-separated subset of the legs (for kernel traces); default all")
ap.add_argument("--preset", default="binary", choices=["binary", "int2", "lowrank8", "lowrank16", "lowrankq32"],
                help="the shipped preset the exchange runs (reference examples/configs.py:39-98); other than binary: the legs `attention`, "
                     "`attention_on_compute_lane`, `layer_op` (lane off: the one-call exchange on the caller's stream) and `default` "
                     "(every switch at its default: the lane for the streaming codecs, the layer op for the low-rank family)")
args = ap.parse_args()
PRESETS = {"binary": ("BINARY", dict(comp_rank=-1, fastpath=True)), "int2": ("INT2", dict(comp_rank=-1, fastpath=True)),
           "lowrank8": ("LOW_RANK", dict(comp_rank=8, fastpath=False)), "lowrank16": ("LOW_RANK", dict(comp_rank=16, fastpath=False)),
           "lowrankq32": ("LOW_RANK_Q", dict(comp_rank=32, fastpath=False))}
PTYPE, PKW = PRESETS[args.preset]

os.environ["CFX_FAKE_RCCL_MODE"] = "loopback"
from compactfusion_amd import _lib, codecs as K, exchange
from compactfusion_amd.compact import ring, main as cm
from compactfusion_amd.compact.attention import block_attention, update_out_and_lse
from compactfusion_amd.compact.utils import CompactConfig, COMPACT_COMPRESS_TYPE as T
from compactfusion_amd.collector import collector
from compactfusion_amd.prof import Profiler

W, L, N, H, D = 8, args.layers, 544, 24, 128
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
lib = _lib.load()

# ---- the 8-rank group, looped back in-process ---------------------------------------------------------------------------
ring.dist.get_rank = lambda g=None: 0
ring.dist.get_world_size = lambda g=None: W


def _ag(recv, send, group=None):
    recv.view(W, -1).copy_(send.view(1, -1).expand(W, -1))


ring.dist.all_gather_into_tensor = _ag
sys.path.insert(0, os.path.join(REPO, "tests", "fake_rccl"))
import build as fake_build
FAKE = fake_build.build()


class LoopComm:
    def __init__(self, group, device):
        ctx = K.context(device)
        assert lib.cfx_rccl_load(FAKE.encode()) == 0
        uid = ctypes.create_string_buffer(128)
        assert lib.cfx_comm_unique_id(ctx, uid) == 0
        self.handle = lib.cfx_comm_create(ctx, uid, W, 0)
        assert self.handle


Profiler.instance().disable()
collector.init(collector.Collector("/tmp/none", enabled=False))
g = torch.Generator(device=dev).manual_seed(1)
qs = [torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)]
k0 = [torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)]
v0 = [torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)]
drift = [[0.1 * torch.randn(1, N, H, D, device=dev, dtype=torch.float16, generator=g) for _ in range(L)] for _ in range(2)]
ks = [[(k0[l] + drift[s][l]) for l in range(L)] for s in range(2)]
vs = [[(v0[l] - drift[s][l]) for l in range(L)] for s in range(2)]
torch.cuda.synchronize()


def attention_only(i):
    """What the layer costs without any exchange: local block + 7 resident peer blocks, merged in ring order."""
    for l in range(L):
        q, k, v = qs[l], ks[i & 1][l], vs[i & 1][l]
        out = lse = None
        for step in range(W):
            bo, bl = block_attention(q, k, v, 0.0, None, causal=False)
            out, lse = update_out_and_lse(out, lse, bo, bl)
        out = out.to(q.dtype)                          # the epilogue every ring forward has (ring.py:271-272)
        lse = lse.squeeze(dim=-1).transpose(1, 2)


_peer_kv = None


def attention_distinct(i):
    """The same, but every peer block reads ITS OWN resident K,V (the 7 peers' cached reconstructions, as the exchange legs do):
    8 distinct K,V pairs per layer instead of one pair that stays hot in L2 for the whole layer."""
    for l in range(L):
        q, k, v = qs[l], ks[i & 1][l], vs[i & 1][l]
        bo, bl = block_attention(q, k, v, 0.0, None, causal=False)
        out, lse = update_out_and_lse(None, None, bo, bl)
        for kk, vv in _peer_kv[l]:
            bo, bl = block_attention(q, kk, vv, 0.0, None, causal=False)
            out, lse = update_out_and_lse(out, lse, bo, bl)
        out = out.to(q.dtype)
        lse = lse.squeeze(dim=-1).transpose(1, 2)


def fwd(i):
    cm.compact_set_step(i)
    for l in range(L):
        ring.compact_fwd(qs[l], ks[i & 1][l], vs[i & 1][l], causal=False, mod_idx=l, current_iter=i)


def init(mode, xstream="chain", lane_mode="off"):
    os.environ["CFX_RING_EXCHANGE"] = mode
    os.environ["CFX_RING_EXCHANGE_STREAM"] = xstream
    os.environ["CFX_LANE"] = lane_mode
    exchange.set_comm_factory(LoopComm if mode != "torch" else None)
    ring._xbuf.clear()
    ring._steady.clear()
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: T.WARMUP if s == 0 else T[PTYPE], residual=1, ef=True, **PKW))
    fwd(0); fwd(1); fwd(2)
    torch.cuda.synchronize()


def timed(fn, first):
    fn(first); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        fn(first + 1 + i)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # host issue time: one step at a time into an EMPTY queue (a full queue would make the host wait for the GPU)
    host = 0.0
    for i in range(args.steps):
        th = time.perf_counter()
        fn(first + 1 + args.steps + i)
        host += time.perf_counter() - th
        torch.cuda.synchronize()
    return wall / args.steps * 1e3, host / args.steps * 1e3


from compactfusion_amd import lanes
comp_stream = lanes.compute_stream(0)
# (`default` / `sticky` come LAST: they bring two more streams into the process, and with more streams than hardware queues
# (GPU_MAX_HW_QUEUES = 8) the flag-ordered legacy legs that follow share queues with something and are time-sliced - measured 36-43 ms per
# step for `lane_unmasked` / `native` behind them, 23.2 / 23.9 in a process of their own)
ALL = ["attention_on_compute_lane", "lane", "attention_distinct_kv_on_compute_lane", "attention", "layer_op", "lane_unmasked", "native", "native_gather_only_on_side", "torchdist", "default", "sticky"]
if args.preset != "binary" and not args.legs:
    ALL = ["attention_on_compute_lane", "attention", "layer_op", "default", "sticky"]
legs = [x for x in (args.legs.split(",") if args.legs else ALL) if x]
assert all(x in ALL for x in legs), f"legs must be among {ALL}"
if args.quick:
    legs = [x for x in legs if x not in ("native_gather_only_on_side", "torchdist")]
res = {}
default_path = None
lane_used = native_used = None
_side = None
for leg in legs:
    if leg == "attention_on_compute_lane":
        with torch.cuda.stream(comp_stream):
            attention_only(0); torch.cuda.synchronize()
            res[leg] = timed(attention_only, 0)
    elif leg in ("default", "sticky"):
        # the caller on an ORDINARY stream, every switch at its default
        if _side is None:
            _side = torch.cuda.Stream(dev)
        with torch.cuda.stream(_side):
            from compactfusion_amd.compact import xlayer as _xl
            if PTYPE.startswith("LOW_RANK"):
                _xl.set_p2p_loopback(True)             # (the low-rank family's default is the layer op: packets in the arena)
            init("native", "auto", "auto" if leg == "default" else "sticky")
            took_lane = all(ex.plan is not None and ex.lane for ex in ring._xbuf.values() if ex.sig is not None)
            took_xop = all(ex.xop is not None for ex in ring._xbuf.values() if ex.sig is not None)
            assert took_lane or (PTYPE.startswith("LOW_RANK") and took_xop), "the default path took neither the lane nor the layer op"
            lr_lane = PTYPE.startswith("LOW_RANK") and all(getattr(ex.xop, "_lane", None) is not None for ex in ring._xbuf.values() if ex.sig is not None)
            default_path = "exchange lane" if took_lane else ("layer op: factor chain on the compute lane, the peers' reconstructions on the exchange lane"
                                                              if lr_lane else "layer op on the caller's stream")
            res[leg] = timed(fwd, 3)
            if PTYPE.startswith("LOW_RANK"):
                for ex in ring._xbuf.values():
                    ex.close()
                ring._xbuf.clear(); ring._steady.clear()
                _xl.set_p2p_loopback(False)
    elif leg == "lane":
        with torch.cuda.stream(comp_stream):
            init("native", "lane")
            lane_used = all(ex.plan is not None and ex.lane for ex in ring._xbuf.values() if ex.sig is not None)
            res[leg] = timed(fwd, 3)
    elif leg == "attention_distinct_kv_on_compute_lane":
        with torch.cuda.stream(comp_stream):
            if not any(e.sig is not None for e in ring._xbuf.values()):
                init("native", "lane")
            by_layer = {e.kkeys[0].split("-")[0]: e for e in ring._xbuf.values() if e.sig is not None}
            _peer_kv = [[(kk.clone(), vv.clone()) for kk, vv in by_layer[str(l)].peer_views] for l in range(L)]
            attention_distinct(0); torch.cuda.synchronize()
            res[leg] = timed(attention_distinct, 0)
    elif leg == "attention":
        attention_only(0); torch.cuda.synchronize()
        res[leg] = timed(attention_only, 0)
    elif leg == "layer_op":
        # the default OFF the lane (round 4): ONE native op per layer (compact/xlayer.py) on the model's stream, in front of the local block;
        # packets in the uncached IPC arena, the 8 logical ranks looped back
        from compactfusion_amd.compact import xlayer
        xlayer.set_p2p_loopback(True)
        init("native", "xlayer")
        assert all(ex.xop is not None and ex.xop.transport == "p2p" for ex in ring._xbuf.values() if ex.sig is not None)
        res[leg] = timed(fwd, 3)
        for ex in ring._xbuf.values():
            ex.close()
        ring._xbuf.clear(); ring._steady.clear()
        xlayer.set_p2p_loopback(False)
    elif leg == "lane_unmasked":
        init("native", "lane")
        res[leg] = timed(fwd, 3)
    elif leg == "native":
        init("native", "chain")
        native_used = all(ex.plan is not None for ex in ring._xbuf.values() if ex.sig is not None)
        res[leg] = timed(fwd, 3)
    elif leg == "native_gather_only_on_side":
        init("native", "side")
        res[leg] = timed(fwd, 3)
    elif leg == "torchdist":
        init("torch")
        res[leg] = timed(fwd, 3)
    torch.cuda.synchronize()


# host cost of the exchange itself, measured directly (the difference of two ~13 ms host-issue legs is noise): the native
# call of a layer and the steady-state lane's checks, each into an empty queue
def _host_us(fn, n=200):
    tot = 0.0
    for _ in range(n):
        torch.cuda.synchronize()
        th = time.perf_counter(); fn(); tot += time.perf_counter() - th
    return tot / n * 1e6


direct = None
if not args.legs and args.preset == "binary":
    init("native", "lane")
    _ex = [e for e in ring._xbuf.values() if e.sig is not None and e.plan is not None][0]
    _st = next(iter(ring._steady.values()))
    _sh = torch.cuda.current_stream().cuda_stream
    _cfg = cm.compact_config()
    direct = {"run_lane (ready flag + wait + compress + all-gather + 7 x (reconstruct, flag) + own EF, one C call)": round(_host_us(lambda: _ex.run_lane(ks[0][0], vs[0][0], _sh)), 1)}
    direct["steady-lane checks (compress_func, matches, current_stream)"] = round(_host_us(
        lambda: (_cfg.compress_func(0, 5), _st.matches(qs[0], ks[0][0], vs[0][0], _st.ctype, _cfg, False, 0), torch.cuda.current_stream(dev).cuda_stream)), 1)
    direct["total"] = round(sum(direct.values()), 1)
    torch.cuda.synchronize()
att = res["attention"][0] if "attention" in res else None
att_lane = res["attention_on_compute_lane"][0] if "attention_on_compute_lane" in res else None
att_dist = res["attention_distinct_kv_on_compute_lane"][0] if "attention_distinct_kv_on_compute_lane" in res else None
base_of = lambda k: att_lane if k in ("lane", "default", "sticky") else att       # noqa: E731   each leg against attention on ITS compute stream
out = {
    "protocol": "SURVEY.md 8d(2): compact_fwd (gather schedule) with PyTorch-ROCm SDPA, one MI355X, 8 logical ranks looped back",
    "shape": {"q_k_v": [1, N, H, D], "layers": L, "ring": W, "codec": f"{PTYPE} residual + EF" + (f", rank {PKW['comp_rank']}" if PKW["comp_rank"] > 0 else ""),
              "preset": args.preset},
    "default_path": default_path,
    "steps": args.steps,
    "lane": {"exchange_cus": lanes.lane(0).exchange_cus, "compute_cus": lanes.lane(0).compute_cus, "lane_plan_used": lane_used},
    "native_plan_used": native_used,
    "legs_ms_per_step": {k: {"wall": round(v[0], 3), "host_issue": round(v[1], 3)} for k, v in res.items()},
    "exposed_exchange_ms_per_step": {k: round(res[k][0] - base_of(k), 3) for k in res if not k.startswith("attention") and base_of(k) is not None},
    "exposed_exchange_ms_per_step_vs_attention_over_distinct_kv": None if att_dist is None or "lane" not in res else round(res["lane"][0] - att_dist, 3),
    "host_us_per_layer": {k: round(res[k][1] * 1e3 / L, 1) for k in res},
    "native_exchange_host_us_per_layer_measured_directly": direct,
    "note": "exposed = wall(leg) - wall(attention on the same compute stream); the attention legs run all 8 blocks of a layer on ONE K,V pair "
            "(hot in L2 after the first block), attention_distinct_kv reads the 7 peers' own resident K,V like the exchange legs do - the fairer base; "
            " the collective is a loop-back device copy (no xGMI wire "
            "time); the torchdist leg excludes torch.distributed's own ~50 us/call host cost (its collective is a plain tensor copy here)",
}
if not args.quiet:
    print(json.dumps(out))
if args.json:
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1)
