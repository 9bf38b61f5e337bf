"""bench.py, part (a): the workload and the state of one bench process.

Constants of the judged workload, the command line, the `Run` object every other part works on (ranks, the synthetic inputs and states
resident in HBM, packet buffers, the native context), the item builders of the plans, the CPU baseline and the copy-bandwidth probe.
Nothing here decides a schedule (schedules.py), checks a result (safety.py) or formats a line (report.py)."""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

L_LAYERS, N_TOK, C_CH, W_LOGICAL = 57, 544, 3072, 8
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s float4-copy achievable)
ALG_BYTES = {"binary": {"compress": 6.125, "decompress": 4.125},   # SURVEY.md section 8d, bytes per element
             "int2": {"compress": 6.25, "decompress": 4.25}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 100 x ~1.5 ms: a timed region of ~150 ms (20 steps were a 30 ms sample, clocks still settling)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--layers", type=int, default=L_LAYERS, help="debug only; the judged workload uses 57")
    ap.add_argument("--rows", type=int, default=0, help="rows per tile override (0 = auto)")
    ap.add_argument("--ipc-memory", type=int, default=2, choices=[0, 1, 2],
                    help="debug: what cfx_ipc_alloc asks for first: 2 uncached (default), 1 fine-grained, 0 ordinary device memory")
    ap.add_argument("--stats-rows", type=int, default=0, help="debug: statistics tile height of the compress launches (cfx_set_stats_rows; 0 = auto)")
    ap.add_argument("--replay", choices=["inorder", "pipelined"], default="inorder",
                    help="inorder (default, the deployable schedule): cfx_plan_run, two launches per layer one after the other; "
                         "pipelined: cfx_plan_run_pipelined, reorders work ACROSS layers (resident synthetic inputs only)")
    ap.add_argument("--own-ef", choices=["gated", "ride", "inline", "xgate"], default="xgate",
                    help="inorder replay. xgate (default; 1-bit, all-gather pattern, native exchange - otherwise it behaves as ride): ONE launch per "
                         "layer with the collective IN the path - the reconstruction workgroups are launched with the compress group, pull their "
                         "state tiles into registers and wait for a gate the exchange stream sets after ncclAllGather "
                         "(cfx_plan_add_exchange_layer).  gated (1-bit, no collective between compress and reconstruction, i.e. N = 1): ONE launch per layer - "
                         "the reconstruction of everything whose packet the layer's compress produces (own error feedback + looped-back peers) "
                         "runs in the compress launch behind an arrival gate (cfx_compress_batch_gated); with a collective in between it "
                         "behaves as ride.  ride: the own error-feedback update rides in the NEXT layer's compress launch, two launches "
                         "per layer.  inline: it sits in the same layer's reconstruction launch (16 tensors per launch)")
    ap.add_argument("--exchange-stream", choices=["main", "side", "prio"], default="prio",
                    help="N > 1, pipelined replay only: 'main' issues every all-gather in order on the compute stream; 'side' / 'prio' "
                         "(prioritised stream) issue it on an exchange stream one unit ahead, underneath the next fused launch")
    ap.add_argument("--gather-group", type=int, default=0,
                    help="N > 1, native exchange: layers (1..7) whose packets travel in ONE all-gather; 0 = 1 for the in-order replay "
                         "(a model has one layer's packets at a time), 7 for the pipelined replay")
    ap.add_argument("--codec", choices=["binary", "int2"], default="binary",
                    help="binary (default, the judged workload: BASELINE.json configs[2]); int2 = the reference's other fused preset "
                         "(examples/configs.py:51-61), in-order replay only, reported as a secondary line")
    ap.add_argument("--no-collective", action="store_true",
                    help="N = 1 debug: build the step WITHOUT the collective between compress and reconstruction (the codec launches alone; "
                         "--own-ef gated needs it: one launch per layer only exists when nothing sits between the two)")
    ap.add_argument("--emulate-live", type=int, default=0,
                    help="N = 1 debug: lay the exchange out for this many live ranks (2..8) over a LOOP-BACK collective library (--rccl-lib: "
                         "tests/fake_rccl in loopback mode, every peer is this rank) - exercises the N > 1 plans, the raw baseline and the "
                         "xgmi object on one GPU; the figures are not link measurements")
    ap.add_argument("--rccl-lib", default=None, help="debug: collective library to load instead of the RCCL the process already uses")
    ap.add_argument("--no-raw-baseline", action="store_true", help="N > 1: skip the uncompressed all-gather legs")
    ap.add_argument("--overlap-steps", type=int, default=12,
                    help="N = 1: after the timed legs, also run SURVEY 8d protocol 2 for this many steps (tools/overlap_bench.py in-process: compact_fwd "
                         "on the exchange lane beside real SDPA attention, 8 logical ranks looped back) and carry its exposed-exchange figure; 0 = skip")
    ap.add_argument("--plugin-steps", type=int, default=40,
                    help="N = 1: steps of the plugin_path leg (tools/plugin_path_bench.py as a child process: the same step through compact_all_gather_kv / "
                         "compact_fwd with a no-op attention); 0 = skip")
    ap.add_argument("--no-config-table", action="store_true", help="N = 1: skip the per-BASELINE-configuration table (`configs`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the dominant kernel with events")
    ap.add_argument("--no-secondary", action="store_true", help="skip the long run and the pipelined upper-bound leg")
    ap.add_argument("--long-steps", type=int, default=200, help="steps of the long timed leg that follows the contract's K steps")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="debug: 'gloo' lets several ranks share one GPU to exercise the N>1 path")
    ap.add_argument("--same-gpu", action="store_true", help="debug: every rank uses cuda:0")
    ap.add_argument("--p2p", choices=["auto", "off"], default="auto",
                    help="--own-ef xgate, every N: auto (default) = no collective at all - every rank's packets stay in IPC-shared memory and the peers' "
                         "reconstruction workgroups read them in place (cfx_plan_add_exchange_layer_p2p; single node; no live peer at N = 1); "
                         "off = ncclAllGather between a flag-wait and a flag-set kernel on the exchange stream")
    ap.add_argument("--dist-path", action="store_true",
                    help="debug: take the N>1 code path (per-layer collectives) even with one rank, to measure its host overhead")
    ap.add_argument("--exchange", choices=["native", "torch"], default="native",
                    help="N>1: who issues the per-layer all-gather - libcfx's own RCCL communicator from the native plan "
                         "(one host call per step) or torch.distributed (one Python call per layer)")
    ap.add_argument("--exchange-pattern", choices=["allgather", "relay"], default="allgather",
                    help="N>1, in-order replay: one direct all-gather per layer (default; xGMI is a point-to-point mesh) or the "
                         "reference's ring relay (W-1 grouped send/recv hops per layer, xfuser/compact/ring.py:193-195)")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="(kept for old command lines; a multi-rank run now ALWAYS falls back in-process - p2p -> ncclAllGather in stream order -> "
                         "torch.distributed per layer - instead of exiting non-zero, and records it in `schedule_fallback`)")
    ap.add_argument("--poison-after-step", type=int, default=-1,
                    help="debug (N > 1, p2p): after this step rank 0 corrupts one reconstructed state - what a stale cache line would leave - to "
                         "exercise validate-then-fall-back")
    ap.add_argument("--copy-probe", type=int, default=0,
                    help="also launch the 96 MiB float4 copy probe this many times before the timed region "
                         "(known byte count: calibrates FETCH_SIZE / WRITE_SIZE in PMC profiles)")
    ap.add_argument("--no-copy-rate", action="store_true",
                    help="skip the copy-bandwidth measurement behind roofline.achievable_gbs (eight 96 MiB copy launches before the warm-up)")
    ap.add_argument("--print-config-key", action="store_true",
                    help="print the configuration key profile summaries are matched against (tools/collect_profiles.sh) and exit")
    ap.add_argument("--event-stride", type=int, default=29,
                    help="bracket every k-th launch of the profiled kernels with hipEvents (an event pair costs a few us of stream time)")
    return ap.parse_args()


def cpu_baseline(seconds: float, codec: str = "binary"):
    """C oracle on the host cores: one layer of the workload = 2 compress + 14 decompress at (544, 3072)."""
    import numpy as np
    from oracle import c_oracle as CO
    N, C = N_TOK, C_CH
    rng = np.random.default_rng(0)
    base = rng.standard_normal((N, C)).astype(np.float16)
    xs = [(base.astype(np.float32) + 0.1 * rng.standard_normal((N, C)).astype(np.float32)).astype(np.float16) for _ in range(2)]
    own = [base.copy().view(np.uint16) for _ in range(2)]
    peers = [base.copy().view(np.uint16) for _ in range(14)]
    pk = [np.zeros(CO.load().oracle_packet_bytes(1 if codec == "binary" else 2, N, C, 0) // 2, dtype=np.uint16) for _ in range(2)]
    def one_layer():
        for i in range(2):
            CO.compress(codec, xs[i], own[i], N, C, packet=pk[i], new_base=own[i])
        for j in range(14):
            CO.decompress(codec, pk[j % 2], peers[j], N, C, out=peers[j])

    one_layer()                                       # warm up (tables, threads, page faults)
    # The baseline is the CPU's BEST: the fastest thread count of a short sweep (a box may report more hardware threads than it schedules
    # for us - 128 reported threads measured 3.6x slower than 64 on the round-4 box); the all-threads figure is carried beside it.
    most = int(CO.num_threads())

    def timed_run(threads, budget):
        CO.set_num_threads(threads)
        one_layer()
        t0 = time.perf_counter()
        reps = 0
        while True:
            one_layer()
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= budget or reps >= 2000:
                break
        return reps, dt

    def rate(reps, dt):
        return round(reps * 16 * N * C * 2 / dt / 1e9, 4)
    reps_all, dt_all = timed_run(most, seconds * 0.4)
    best_t, best = most, dt_all / reps_all
    for t in sorted({c for c in (4, 8, 16, 32, 64, 128, 256) if c < most}):
        CO.set_num_threads(t)
        one_layer()
        t0 = time.perf_counter()
        one_layer()
        dt1 = time.perf_counter() - t0
        if dt1 < best:
            best, best_t = dt1, t
    reps, dt = (reps_all, dt_all) if best_t == most else timed_run(best_t, seconds * 0.6)
    if best_t != most and rate(reps, dt) < rate(reps_all, dt_all):      # (the sweep's single-shot pick did not hold up over the longer run)
        best_t, reps, dt = most, reps_all, dt_all
    CO.set_num_threads(most)
    return {"value": rate(reps, dt), "unit": "GB/s", "cores": best_t, "kind": "port",
            "all_threads": {"value": rate(reps_all, dt_all), "cores": most},
            "sample": f"{reps} x one layer of the workload (2 compress + 14 decompress, {'1-bit' if codec == 'binary' else '2-bit'}, (544,3072) fp16) in {dt:.1f} s, "
                      f"C oracle oracle/cfx_oracle.c with OpenMP on {best_t} threads = the fastest of a sweep over 4 .. {most} (the box reports {most} hardware threads; "
                      f"`all_threads` = the same on all of them) "
                      f"({'F16C conversions' if CO.load().oracle_uses_f16c() else 'software fp16 conversions'}); GB/s of fp16 activations through the codec"}


def measure_copy_rate(lib, ctx, dev, stream_handle, reps=6):
    """What this box's HBM sustains on a plain copy (the roofline fraction is quoted against the 8 TB/s spec AND against
    this): the 96 MiB float4 copy probe of libcfx (read 96 MiB + write 96 MiB per launch, four buffer pairs in turn: 768 MiB, past the
    Infinity Cache), hipEvents on the launch stream around groups of four launches after a warm-up group; the median group."""
    import torch
    nb = 96 * 1024 * 1024
    src = [torch.empty(nb, dtype=torch.uint8, device=dev).random_(0, 255) for _ in range(4)]
    dst = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(4)]
    st = torch.cuda.ExternalStream(stream_handle, device=dev)
    ev = []
    for g_ in range(reps + 1):                     # groups of four back-to-back launches (an event pair around ONE launch adds the launch gap)
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record(st)
        for i in range(4):
            if lib.cfx_copy_probe(ctx, dst[i].data_ptr(), src[i].data_ptr(), nb, stream_handle) != 0:
                return None
        b_.record(st)
        ev.append((a_, b_))
    torch.cuda.synchronize(dev)
    us = sorted(a_.elapsed_time(b_) * 1e3 / 4 for a_, b_ in ev[1:])
    del src, dst
    med = us[len(us) // 2]
    return {"achievable_gbs": round(2 * nb / (med * 1e-6) / 1e9, 1), "copy_probe_us": round(med, 2), "copy_probe_launches": 4 * reps,
            "copy_probe": "k_copy_probe: 96 MiB read + 96 MiB written per launch (16 B per lane), four buffer pairs back to back between two "
                          "hipEvents, median group / 4"}


def group_recv_offset(l: int, r: int, kv: int, G: int, L: int, live: int, slot: int) -> int:
    """Byte offset of rank r's packet (kv = 0: K, 1: V) of layer l in the grouped receive buffer.

    Layers travel G at a time: group g = layers [a, b) = [gG, min(L, gG + G)).  One all-gather per group sends
    send[a:b] = [layer][K|V][slot] (contiguous, (b-a)*2*slot bytes per rank) and receives [rank][layer in group][K|V][slot];
    the groups' receive regions follow each other, so the region of group g starts after a*live*2*slot bytes."""
    a = (l // G) * G
    b = min(L, a + G)
    return (a * live * 2 + (r * (b - a) + (l - a)) * 2 + kv) * slot


def config_key(args, n_gpus):
    """What a committed profile must have been taken with for its figures to be quoted beside this run's."""
    pipelined = args.replay == "pipelined"
    own_ef = args.own_ef
    if own_ef == "xgate" and ((args.codec != "binary" and args.p2p != "auto") or args.no_collective or args.exchange != "native" or args.exchange_pattern == "relay"):
        own_ef = "ride"
    return {"codec": args.codec, "replay": args.replay, "own_ef": own_ef if not pipelined else None, "layers": args.layers,
            "shard": [N_TOK, C_CH], "rows": args.rows, "n_gpus": n_gpus, "collective": not args.no_collective,
            "p2p": (args.p2p if (own_ef == "xgate" and not args.emulate_live) else None)}




class GateTripped(RuntimeError):
    """A native call refused to launch because an earlier launch's in-kernel wait gave up (include/cfx.h CFX_ERR_GATE).  Not fatal for a
    multi-rank run: the step loop stops issuing on this rank, every rank meets at the validation, the ladder takes the run down a rung."""


class Run:
    """Everything one bench process holds: set up once by `setup`, read by the schedule builders, the safety net and the report.

    ranks        world / rank / local_rank (processes), real_live (ranks that really exist), live (ranks the exchange is LAID OUT for:
                 real_live unless --emulate-live)
    workload     L layers x {K, V} of shape (N, C) fp16: xs[2] (the two input sets a step alternates between), own_base (sender error-feedback
                 states), peer_base (the 7 logical peers' states as this rank reconstructs them), all resident in HBM
    buffers      send / recv / grecv (packets: own, gathered per layer, gathered in groups), ws (the codec workspace)
    flags        what the command line selected and what the fall-back ladder turned it into (schedules.py / safety.py)"""

    def __init__(self, args):
        self.args = args
        self.gate_tripped = False          # a native call refused to launch (CFX_ERR_GATE) since the last validation: runner.guarded_step

    # ---- resident state and inputs -----------------------------------------------------------------------------------------
    def warm_state(self, src_rank):
        """x_0 of rank `src_rank` (what a WARMUP step leaves in every rank's cache for that rank's shard)."""
        torch = self.torch
        gg = torch.Generator(device=self.dev).manual_seed(1234 + src_rank)
        return gg, torch.randn(self.L, 2, self.N, self.C, generator=gg, device=self.dev, dtype=torch.float32).half()

    def reset_state(self):
        """State as a WARMUP step leaves it: every rank holds x_0 of every shard it tracks."""
        x0_ = self.warm_state(self.rank)[1]
        self.own_base.copy_(x0_)
        for p in range(W_LOGICAL - 1):
            if self.real_live > 1 and p < self.real_live - 1:
                self.peer_base[:, p] = self.warm_state((self.rank + 1 + p) % self.real_live)[1]      # a real peer: its own x_0
            else:
                self.peer_base[:, p] = x0_                                       # looped-back logical peer

    # ---- where packets live ----------------------------------------------------------------------------------------------------
    def own_pkt_ptr(self, l, kv, gathered):
        """Where the rank's own packet of layer l is written: with a collective, straight into ITS slot of the gather buffer - the
        all-gather is then in place (no local copy; with one live rank RCCL has nothing to move at all)."""
        if gathered:
            return self.grecv.data_ptr() + group_recv_offset(l, self.rank, kv, self.G, self.L, self.live, self.slot)
        return self.send[l, kv].data_ptr()

    def peer_packet_ptr(self, l, p, kv, gathered):
        """Packet of logical peer p for layer l: a real rank's slot of the gathered buffer, or (looped-back peer) our own packet
        - taken from OUR slot of the gathered buffer when there is one, so a collective's result is consumed even with one live rank."""
        if gathered:
            real = self.live > 1 and p < self.live - 1
            r = (self.rank + 1 + p) % self.live if real else self.rank          # a looped-back peer reads OUR slot (the compress launch wrote it there)
            return self.grecv.data_ptr() + group_recv_offset(l, r, kv, self.G, self.L, self.live, self.slot)
        if self.real_live > 1 and p < self.real_live - 1:
            return self.recv[l, (self.rank + 1 + p) % self.real_live, kv].data_ptr()
        return self.send[l, kv].data_ptr()

    # ---- plan items ------------------------------------------------------------------------------------------------------------
    def comp_items(self, s_, l, gathered=False):
        _lib = self._lib
        carr = (_lib.CompItem * 2)()
        for kv in range(2):
            carr[kv] = _lib.CompItem(self.xs[s_][l, kv].data_ptr(), self.own_base[l, kv].data_ptr(), None, self.own_pkt_ptr(l, kv, gathered))
        return carr

    def own_ef_items(self, l, gathered=False):
        _lib = self._lib
        return [_lib.DecompItem(self.own_pkt_ptr(l, kv, gathered), self.own_base[l, kv].data_ptr(), self.own_base[l, kv].data_ptr()) for kv in range(2)]

    def peer_items(self, l, gathered):
        _lib = self._lib
        return [_lib.DecompItem(self.peer_packet_ptr(l, p, kv, gathered), self.peer_base[l, p, kv].data_ptr(), self.peer_base[l, p, kv].data_ptr())
                for p in range(W_LOGICAL - 1) for kv in range(2)]

    def check(self, rc, what):
        if rc == -8:                              # CFX_ERR_GATE: an EARLIER launch's gate / flag wait timed out; nothing was launched by this call
            raise GateTripped(f"{what}: {self.lib.cfx_last_error_string(self.ctx)}")
        if rc != 0:
            raise RuntimeError(f"{what}: rc={rc} {self.lib.cfx_last_error_string(self.ctx)}")

    def sync_all(self):
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)


def setup(args) -> Run:
    """Ranks, the process group, the native context, the resident workload.  Returns the Run every later stage works on."""
    import torch
    import torch.distributed as dist
    S = Run(args)
    S.torch, S.dist = torch, dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    S.world, S.rank, S.local_rank, S.dev = world, rank, local_rank, dev
    S.real_live = world                  # ranks that really exist (processes / GPUs)
    if args.emulate_live:
        assert world == 1 and 2 <= args.emulate_live <= W_LOGICAL and args.rccl_lib, "--emulate-live needs one process and --rccl-lib (a loop-back library)"
        os.environ["CFX_FAKE_RCCL_MODE"] = "loopback"
    S.live = args.emulate_live or world  # ranks the exchange is LAID OUT for (= real_live unless --emulate-live)
    assert S.live <= W_LOGICAL
    S.pipelined = args.replay == "pipelined"
    S.int2 = args.codec == "int2"
    if S.int2 and S.pipelined:
        raise SystemExit("--codec int2 runs the in-order replay only (the cross-layer pipeline is 1-bit only)")
    S.alg = ALG_BYTES[args.codec]
    G = args.gather_group if args.gather_group > 0 else (7 if S.pipelined else 1)
    S.G = max(1, min(7, G))
    S.relay = args.exchange_pattern == "relay"
    if S.relay and (S.pipelined or S.G != 1):
        raise SystemExit("--exchange-pattern relay is an in-order, one-layer-per-exchange schedule")

    from compactfusion_amd import _lib, codecs as K
    S._lib, S.K = _lib, K
    S.lib = lib = _lib.load()
    S.ctx = ctx = K.context(local_rank)
    if args.rows:
        K.set_rows_per_tile(args.rows, local_rank)
    if args.stats_rows:
        assert lib.cfx_set_stats_rows(ctx, args.stats_rows) == 0
    if args.ipc_memory != 2:
        assert lib.cfx_set_ipc_memory_kind(ctx, args.ipc_memory) == 0

    S.L, S.N, S.C = L, N, C = args.layers, N_TOK, C_CH
    S.CODEC = int(K.Codec.INT2 if S.int2 else K.Codec.BINARY)
    S.pkt_bytes = K.packet_bytes(S.CODEC, N, C)
    S.slot = (S.pkt_bytes + 255) // 256 * 256          # per-tensor slot in the exchange buffer, 256-B aligned
    g, x0 = S.warm_state(rank)
    S.xs = [(x0.float() + 0.1 * torch.randn(L, 2, N, C, generator=g, device=dev)).half() for _ in range(2)]
    S.own_base = torch.empty_like(x0)                                          # [L,2,N,C] sender EF state
    S.peer_base = torch.empty(L, W_LOGICAL - 1, 2, N, C, dtype=torch.float16, device=dev)   # receiver states
    del x0
    S.reset_state()
    S.send = torch.zeros(L, 2, S.slot, dtype=torch.uint8, device=dev)           # own packets (K,V) per layer
    # the collective sits in the path at EVERY N (N = 1: a one-rank RCCL communicator - what N = 8 executes minus the wire)
    S.use_dist = not args.no_collective
    if args.own_ef == "gated" and S.use_dist and not S.pipelined:
        raise SystemExit("--own-ef gated (one launch per layer) only exists without a collective between compress and reconstruction: add --no-collective")
    S.recv = torch.zeros(L, S.live, 2, S.slot, dtype=torch.uint8, device=dev) if (S.use_dist and S.real_live > 1) else None
    S.grecv = torch.zeros(L * S.live * 2 * S.slot, dtype=torch.uint8, device=dev) if S.use_dist else None   # grouped receive regions
    S.ws_bytes = lib.cfx_workspace_bytes(S.CODEC, N, C, 0, 2)
    S.ws = torch.empty(S.ws_bytes, dtype=torch.uint8, device=dev)
    S.steps_run = 0
    return S
