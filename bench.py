#!/usr/bin/env python3
"""bench.py - residual-compressed activation exchange, FLUX.1-dev 1024^2 ring-8 per-rank workload.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per
GPU with torch.distributed.run.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2], SURVEY.md §8d "Config 3"): one denoise step of FLUX.1-dev 1024^2 under
ring-attention sequence parallelism of logical degree 8 with the 1-bit residual codec, as seen by ONE rank:
57 attention layers x {K, V}, shard (N, C) = (544, 3072) fp16.  Per layer the rank
  1. compresses its own K and V against its error-feedback state (one batched launch sequence),
  2. exchanges packets (N live ranks all-gather over RCCL, --gather-group layers per collective, issued natively on an
     exchange stream underneath the next fused launch; the 8-N missing logical peers are looped back from the rank's own packet, so the per-GPU codec work is IDENTICAL for every N = weak scaling;
     at N = 8 this is exactly the real exchange, at N = 1 it is the codec path alone),
  3. applies 16 packets in ONE batched dequant+add launch: its own K,V packets onto its own state (the error-feedback
     update of step 1, deferred into this launch) and the 7 peers' K,V onto their state arenas.
Inputs are synthetic and already resident in HBM; the state arenas (3.0 GB) + inputs (0.76 GB) dwarf the 256 MB
Infinity Cache, so every step streams from HBM (cold numbers).

value = whole-job fp16 activation bytes compressed + reconstructed per second (GB/s), i.e.
        n_gpus * 57 * (2 + 14) * 544*3072*2 B / step time.
Replay (--replay): `pipelined` (default) = cfx_plan_run_pipelined: per UNIT of layers ONE fused launch k_binary_pipe =
        [dequant+add of unit u's 16 tensors per layer | finalize of unit u+1's scales | stats + sign bits of unit u+2's K,V],
        a unit = 7 consecutive layers, so the small latency-bound compress kernels ride underneath the bandwidth-bound
        reconstruction and a launch is long enough to amortise its ramp and tail; `inorder` = cfx_plan_run,
        stats -> finalize -> dequant one after the other.  Same results bit for bit (tests/test_gpu_api.py).  The pipelined
        replay reorders work across layers (legal with resident synthetic inputs); the JSON also carries the in-order time.
roofline = the dominant kernel: k_binary_pipe (pipelined; algorithmic bytes 4.125 B/element x (16 + 2) tensors of
        544*3072 elements per layer, 7 layers per launch) or k_binary_dequant (inorder; 4.125 B/element x 16 tensors) / average
        launch duration from hipEvents attached to the dispatch on the launch stream inside the timed region (native
        hooks in libcfx.so).
cpu_baseline = the C oracle (oracle/cfx_oracle.c, OpenMP) timed on this box's host cores on one layer of the same
        workload; reported baseline only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

L_LAYERS, N_TOK, C_CH, W_LOGICAL = 57, 544, 3072, 8
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s float4-copy achievable)
ALG_BYTES_PER_EL = {"compress": 6.125, "decompress": 4.125}   # 1-bit, SURVEY.md §8d


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=int, default=L_LAYERS, help="debug only; the judged workload uses 57")
    ap.add_argument("--rows", type=int, default=0, help="rows per tile override (0 = auto)")
    ap.add_argument("--replay", choices=["inorder", "pipelined"], default="pipelined",
                    help="pipelined (default): cfx_plan_run_pipelined - one fused launch per layer, the statistics / finalize "
                         "work of the next two layers rides underneath the reconstruction of the current one; "
                         "inorder: cfx_plan_run, three launches per layer one after the other (same results, bit for bit)")
    ap.add_argument("--exchange-stream", choices=["main", "side", "prio"], default="prio",
                    help="N > 1, native exchange: 'main' issues every all-gather in order on the compute stream; 'side' / 'prio' "
                         "(prioritised stream) issue it on an exchange stream one unit ahead, underneath the next fused launch")
    ap.add_argument("--gather-group", type=int, default=7,
                    help="N > 1, native exchange: layers (1..7) whose packets travel in ONE all-gather (fewer, larger collectives; "
                         "a group is replayed as one unit of the pipelined schedule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the dominant kernel with events")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="debug: 'gloo' lets several ranks share one GPU to exercise the N>1 path")
    ap.add_argument("--same-gpu", action="store_true", help="debug: every rank uses cuda:0")
    ap.add_argument("--dist-path", action="store_true",
                    help="debug: take the N>1 code path (per-layer collectives) even with one rank, to measure its host overhead")
    ap.add_argument("--exchange", choices=["auto", "native", "torch"], default="auto",
                    help="N>1: who issues the per-layer all-gather - libcfx's own RCCL communicator from the native plan "
                         "(one host call per step) or torch.distributed (one Python call per layer); auto = native, "
                         "falling back to torch if the communicator cannot be created")
    ap.add_argument("--copy-probe", type=int, default=0,
                    help="also launch the 96 MiB float4 copy probe this many times before the timed region "
                         "(known byte count: calibrates FETCH_SIZE / WRITE_SIZE in PMC profiles)")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="bracket every k-th launch of the dominant kernel with hipEvents (an event pair costs a few us of stream time)")
    return ap.parse_args()


def cpu_baseline(seconds: float):
    """C oracle on the host cores: one layer of the workload = 2 compress + 14 decompress at (544, 3072)."""
    import numpy as np
    from oracle import c_oracle as CO
    N, C = N_TOK, C_CH
    rng = np.random.default_rng(0)
    base = rng.standard_normal((N, C)).astype(np.float16)
    xs = [(base.astype(np.float32) + 0.1 * rng.standard_normal((N, C)).astype(np.float32)).astype(np.float16) for _ in range(2)]
    own = [base.copy().view(np.uint16) for _ in range(2)]
    peers = [base.copy().view(np.uint16) for _ in range(14)]
    pk = [np.zeros(CO.load().oracle_packet_bytes(1, N, C, 0) // 2, dtype=np.uint16) for _ in range(2)]
    def one_layer():
        for i in range(2):
            CO.compress("binary", xs[i], own[i], N, C, packet=pk[i], new_base=own[i])
        for j in range(14):
            CO.decompress("binary", pk[j % 2], peers[j], N, C, out=peers[j])

    one_layer()                                       # warm up (tables, threads, page faults)
    # thread count: the box may report more hardware threads than it schedules for us; take the fastest of a short sweep
    most = int(CO.num_threads())
    best_t, best = most, None
    for t in sorted({most} | {c for c in (4, 8, 16, 32, 64, 128, 256) if c <= most}):
        CO.set_num_threads(t)
        one_layer()
        t0 = time.perf_counter()
        one_layer()
        dt1 = time.perf_counter() - t0
        if best is None or dt1 < best:
            best, best_t = dt1, t
    CO.set_num_threads(best_t)
    t0 = time.perf_counter()
    reps = 0
    while True:
        one_layer()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or reps >= 2000:
            break
    act_bytes = reps * 16 * N * C * 2
    return {"value": round(act_bytes / dt / 1e9, 4), "unit": "GB/s", "cores": best_t, "kind": "port",
            "sample": f"{reps} x one layer of the workload (2 compress + 14 decompress, 1-bit, (544,3072) fp16) in {dt:.1f} s, "
                      f"C oracle oracle/cfx_oracle.c with OpenMP ({best_t} of {most} threads: fastest of a sweep; "
                      f"{'F16C conversions' if CO.load().oracle_uses_f16c() else 'software fp16 conversions'})"}


def group_recv_offset(l: int, r: int, kv: int, G: int, L: int, live: int, slot: int) -> int:
    """Byte offset of rank r's packet (kv = 0: K, 1: V) of layer l in the grouped receive buffer.

    Layers travel G at a time: group g = layers [a, b) = [gG, min(L, gG + G)).  One all-gather per group sends
    send[a:b] = [layer][K|V][slot] (contiguous, (b-a)*2*slot bytes per rank) and receives [rank][layer in group][K|V][slot];
    the groups' receive regions follow each other, so the region of group g starts after a*live*2*slot bytes."""
    a = (l // G) * G
    b = min(L, a + G)
    return (a * live * 2 + (r * (b - a) + (l - a)) * 2 + kv) * slot


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.dist_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    live = world                       # live ranks in the logical ring of 8
    assert live <= W_LOGICAL

    from compactfusion_amd import _lib, codecs as K
    lib = _lib.load()
    ctx = K.context(local_rank)
    if args.rows:
        K.set_rows_per_tile(args.rows, local_rank)

    L, N, C = args.layers, N_TOK, C_CH
    CODEC = int(K.Codec.BINARY)
    pkt_bytes = K.packet_bytes(CODEC, N, C)
    slot = (pkt_bytes + 255) // 256 * 256          # per-tensor slot in the exchange buffer, 256-B aligned
    # ---- resident state and inputs --------------------------------------------------------------------------
    def warm_state(src_rank):
        """x_0 of rank `src_rank` (what a WARMUP step leaves in every rank's cache for that rank's shard)."""
        gg = torch.Generator(device=dev).manual_seed(1234 + src_rank)
        return gg, torch.randn(L, 2, N, C, generator=gg, device=dev, dtype=torch.float32).half()

    g, x0 = warm_state(rank)
    xs = [(x0.float() + 0.1 * torch.randn(L, 2, N, C, generator=g, device=dev)).half() for _ in range(2)]
    own_base = torch.empty_like(x0)                                          # [L,2,N,C] sender EF state
    peer_base = torch.empty(L, W_LOGICAL - 1, 2, N, C, dtype=torch.float16, device=dev)   # receiver states
    del x0

    def reset_state():
        """State as a WARMUP step leaves it: every rank holds x_0 of every shard it tracks."""
        x0_ = warm_state(rank)[1]
        own_base.copy_(x0_)
        for p in range(W_LOGICAL - 1):
            if live > 1 and p < live - 1:
                peer_base[:, p] = warm_state((rank + 1 + p) % live)[1]      # a real peer: its own x_0
            else:
                peer_base[:, p] = x0_                                       # looped-back logical peer
    reset_state()
    send = torch.zeros(L, 2, slot, dtype=torch.uint8, device=dev)           # own packets (K,V) per layer
    use_dist = live > 1 or args.dist_path
    recv = torch.zeros(L, live, 2, slot, dtype=torch.uint8, device=dev) if use_dist else None
    ws_bytes = lib.cfx_workspace_bytes(CODEC, N, C, 0, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

    # ---- native plans (one per input set): per layer op 2l = compress K,V ; op 2l+1 = one dequant+add launch over 16
    #      tensors = the sender's own error-feedback update (its K,V packets applied to its own state, which is exactly
    #      what compact_compress(update_cache=True) does, fastpath.py:88-120) + the 7 peers' K,V ---------------------------
    plans = []
    for s in range(2):
        plan = lib.cfx_plan_create(ctx)
        for l in range(L):
            carr = (_lib.CompItem * 2)()
            for kv in range(2):
                carr[kv] = _lib.CompItem(xs[s][l, kv].data_ptr(), own_base[l, kv].data_ptr(), None, send[l, kv].data_ptr())
            rc = lib.cfx_plan_add_compress(plan, CODEC, N, C, 0, 0, 2, carr, ws.data_ptr(), ws_bytes)
            assert rc == 2 * l, (rc, lib.cfx_last_error_string(ctx))
            darr = (_lib.DecompItem * 16)()
            for kv in range(2):
                darr[kv] = _lib.DecompItem(send[l, kv].data_ptr(), own_base[l, kv].data_ptr(), own_base[l, kv].data_ptr())
            i = 2
            for p in range(W_LOGICAL - 1):
                # logical peer p: the first live-1 are real ranks (packets from the all-gather), the rest loop back our own packet
                for kv in range(2):
                    if live > 1 and p < live - 1:
                        pk_ptr = recv[l, (rank + 1 + p) % live, kv].data_ptr()
                    else:
                        pk_ptr = send[l, kv].data_ptr()
                    darr[i] = _lib.DecompItem(pk_ptr, peer_base[l, p, kv].data_ptr(), peer_base[l, p, kv].data_ptr())
                    i += 1
            rc = lib.cfx_plan_add_decompress(plan, CODEC, N, C, 0, 16, darr)
            assert rc == 2 * l + 1, (rc, lib.cfx_last_error_string(ctx))
        plans.append(plan)

    compute = torch.cuda.current_stream(dev)
    sh = compute.cuda_stream

    # ---- N > 1: the whole step as ONE native plan, the all-gathers issued by libcfx's own RCCL communicator.
    #      --gather-group layers share one all-gather (fewer, larger collectives) and form one unit of the pipelined replay;
    #      --exchange-stream prio|side runs the collective of unit u on an exchange stream underneath the fused launch that
    #      follows finalize(u) (one extra unit of look-ahead; the ~10 us cross-stream event hops hide behind an 80 us launch),
    #      'main' keeps everything in order on one stream. ----
    native_comm, step_plans, exchange_mode, stream_mode, build_step_plans = None, None, "none", 0, None
    if use_dist:
        exchange_mode = "torch"
        if args.exchange in ("auto", "native"):
            try:
                from compactfusion_amd.exchange import NativeComm
                native_comm = NativeComm(local_rank)
                native_comm.self_test()
                # grouped exchange buffers: group g = layers [gG, gG + nl); ONE all-gather moves the group's K,V packets of
                # every rank: send = send[gG : gG + nl] (contiguous), recv region laid out [rank][layer in group][K|V][slot]
                G = max(1, min(7, args.gather_group))      # a group must fit one fused launch (<= 7 layers x 16 tensors)
                groups = [(a, min(L, a + G)) for a in range(0, L, G)]
                grecv = torch.zeros(L * live * 2 * slot, dtype=torch.uint8, device=dev)

                def grecv_ptr(l, r, kv):
                    return grecv.data_ptr() + group_recv_offset(l, r, kv, G, L, live, slot)
                def build_step_plans(mode):
                    built = []
                    for s_ in range(2):
                        sp = lib.cfx_plan_create(ctx)
                        src = plans[s_]
                        assert lib.cfx_plan_set_exchange_stream(sp, mode) == 0
                        for a, b in groups:
                            for l in range(a, b):
                                assert lib.cfx_plan_copy_op(sp, src, 2 * l) >= 0
                            rcx = lib.cfx_plan_add_all_gather(sp, native_comm.handle, send[a].data_ptr(), grecv.data_ptr() + a * live * 2 * slot,
                                                              (b - a) * 2 * slot)
                            assert rcx >= 0, rcx
                            for l in range(a, b):
                                darr = (_lib.DecompItem * 16)()
                                for kv in range(2):
                                    darr[kv] = _lib.DecompItem(send[l, kv].data_ptr(), own_base[l, kv].data_ptr(), own_base[l, kv].data_ptr())
                                i = 2
                                for p in range(W_LOGICAL - 1):
                                    for kv in range(2):
                                        # real peers: their slot of the gathered buffer; looped-back logical peers: OUR slot of
                                        # the gathered buffer (so the collective's result is consumed even with one live rank)
                                        pk_ptr = grecv_ptr(l, (rank + 1 + p) % live if (live > 1 and p < live - 1) else rank, kv)
                                        darr[i] = _lib.DecompItem(pk_ptr, peer_base[l, p, kv].data_ptr(), peer_base[l, p, kv].data_ptr())
                                        i += 1
                                assert lib.cfx_plan_add_decompress(sp, CODEC, N, C, 0, 16, darr) >= 0
                        built.append(sp)
                    return built
                # the in-order replay has no wait ops in this plan: everything stays on the compute stream there
                stream_mode = {"main": 0, "side": 1, "prio": 2}[args.exchange_stream] if args.replay == "pipelined" else 0
                step_plans = build_step_plans(stream_mode)
                exchange_mode = "native"
            except Exception as e:  # pragma: no cover
                if args.exchange == "native":
                    raise
                print(f"[bench] native exchange unavailable ({e}); using torch.distributed per layer", file=sys.stderr)
                native_comm, step_plans = None, None

    def check(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: rc={rc} {lib.cfx_last_error_string(ctx)}")

    def one_step(step):
        plan = plans[step & 1]
        run = lib.cfx_plan_run_pipelined if args.replay == "pipelined" else lib.cfx_plan_run
        if not use_dist:
            check(run(plan, 0, 2 * L, sh), "plan_run")     # the whole step from native code
            return
        if step_plans is not None:
            sp = step_plans[step & 1]
            check(run(sp, 0, lib.cfx_plan_size(sp), sh), "plan_run(exchange)")
            return
        # software pipeline: gather(l) runs on RCCL's own stream (async_op: it is ordered after the compute stream's
        # tail at the call and joined back by work.wait()) and overlaps compress(l+1) and reconstruct(l-1)
        works = [None] * L
        check(lib.cfx_plan_run(plan, 0, 1, sh), "compress")
        works[0] = dist.all_gather_into_tensor(recv[0].view(-1), send[0].view(-1), async_op=True)
        for l in range(L):
            if l + 1 < L:
                check(lib.cfx_plan_run(plan, 2 * (l + 1), 1, sh), "compress")
                works[l + 1] = dist.all_gather_into_tensor(recv[l + 1].view(-1), send[l + 1].view(-1), async_op=True)
            works[l].wait()          # compute stream waits for gather(l)
            check(lib.cfx_plan_run(plan, 2 * l + 1, 1, sh), "reconstruct")

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if args.copy_probe:
        nb = 96 * 1024 * 1024
        src = [torch.empty(nb, dtype=torch.uint8, device=dev).random_(0, 255) for _ in range(4)]
        dst = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(4)]
        for i in range(args.copy_probe):
            check(lib.cfx_copy_probe(ctx, dst[i % 4].data_ptr(), src[i % 4].data_ptr(), nb, sh), "copy_probe")
        torch.cuda.synchronize(dev)
        del src, dst

    def states_consistent():
        """What a rank holds for its own shard must be, bit for bit, what every peer reconstructed for that shard."""
        torch.cuda.synchronize(dev)
        # sampled over layers that sit at different positions of an all-gather group, K and V
        samples = sorted({(l, kv) for l in (0, 1, min(L - 1, max(1, args.gather_group) - 1), L // 2, L - 1) for kv in (0, 1) if l < L})
        if live == 1:
            same = all(torch.equal(own_base[l, kv].view(torch.int16), peer_base[l, p, kv].view(torch.int16))
                       for l, kv in samples for p in range(W_LOGICAL - 1))
            return same, "EF state of a looped-back peer diverged from the sender's"
        good = torch.ones(1, dtype=torch.int32, device=dev)
        for l, kv in samples:
            mine = own_base[l, kv].reshape(-1)[:8192].view(torch.int32).contiguous()       # int32: a dtype every backend moves
            allm = torch.empty(live * 4096, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(allm, mine)
            for p in range(live - 1):
                src = (rank + 1 + p) % live
                got = peer_base[l, p, kv].reshape(-1)[:8192].view(torch.int32)
                if not torch.equal(got, allm[src * 4096:(src + 1) * 4096]):
                    good.zero_()
        dist.all_reduce(good, op=dist.ReduceOp.MIN)
        return bool(good.item()), f"rank {rank}: a peer's reconstructed state diverged from its owner's"

    # ---- warmup (+ validation of the exchange path before anything is timed) ---------------------------------------------
    for i in range(max(args.warmup, 1 if use_dist else 0)):
        one_step(i)
    sync_all()
    if use_dist and step_plans is not None:
        ok, why = states_consistent()
        if not ok and stream_mode != 0:
            # the overlapped collectives did not validate on this machine: same native plan, everything in order on one stream
            print("[bench] exchange-stream overlap failed validation; retrying with in-order collectives", file=sys.stderr)
            stream_mode = 0
            step_plans = build_step_plans(0)
            reset_state()
            for i in range(max(args.warmup, 1)):
                one_step(i)
            sync_all()
            ok, why = states_consistent()
        if not ok:
            if args.exchange == "native":
                raise RuntimeError("native exchange produced inconsistent state: " + why)
            print("[bench] native exchange failed validation; falling back to torch.distributed per layer", file=sys.stderr)
            step_plans, exchange_mode = None, "torch"
            reset_state()
            for i in range(max(args.warmup, 1)):
                one_step(i)
            sync_all()

    # ---- timed region -------------------------------------------------------------------------------------------
    # dominant kernel: k_binary_dequant (in-order replay) or the fused k_binary_pipe (pipelined replay; the full
    # three-group launches only - prologue / epilogue launches carry a different id)
    KID_DEQ = 23 if args.replay == "pipelined" else 4
    if not args.no_kernel_events:
        check(lib.cfx_profile_enable(ctx, args.steps * L + 8, 1 << KID_DEQ, args.event_stride), "profile_enable")
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    sync_all()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if live > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms = None
    if not args.no_kernel_events:
        cap = args.steps * L + 8
        ids = (ctypes.c_int * cap)()
        ms = (ctypes.c_float * cap)()
        n = lib.cfx_profile_read(ctx, ids, ms, cap)
        vals = [ms[i] for i in range(n) if ids[i] == KID_DEQ and ms[i] > 0]
        n_samples = len(vals)
        if vals:
            kern_ms = sum(vals) / len(vals)
        lib.cfx_profile_enable(ctx, 0, 0, 1)

    # secondary figure, same K steps replayed layer by layer in order (stats -> finalize -> dequant, three launches per
    # layer): what a caller gets when layer j+1's K,V only exist after layer j's attention.  The pipelined replay reorders
    # work ACROSS layers, which the bench's resident synthetic inputs allow (SURVEY.md section 8d "pure exchange" protocol).
    inorder_ms = None
    if args.replay == "pipelined" and not use_dist:
        sync_all()
        ti = time.perf_counter()
        for i in range(args.steps):
            check(lib.cfx_plan_run(plans[(args.warmup + args.steps + i) & 1], 0, 2 * L, sh), "plan_run(inorder)")
        sync_all()
        inorder_ms = (time.perf_counter() - ti) * 1e3 / args.steps

    # ---- state sanity (bit-exact error-feedback consistency) ---------------------------------------------------------
    ok, why = states_consistent()
    assert ok, why

    # ---- uncompressed RCCL all-gather of the same K/V shards (the north-star comparison), N > 1 only ---------------
    raw_ms = None
    if live > 1:
        try:
            raw_in = xs[0]
            raw_out = torch.empty(live, 2, N, C, dtype=torch.float16, device=dev)
            for _ in range(2):
                for l in range(L):
                    dist.all_gather_into_tensor(raw_out.view(-1), raw_in[l].reshape(-1))
            sync_all()
            tr0 = time.perf_counter()
            reps = max(3, min(args.steps, 10))
            for _ in range(reps):
                for l in range(L):
                    dist.all_gather_into_tensor(raw_out.view(-1), raw_in[l].reshape(-1))
            sync_all()
            raw_ms = (time.perf_counter() - tr0) / reps * 1e3
            t = torch.tensor([raw_ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            raw_ms = float(t.item())
        except Exception as e:  # pragma: no cover
            raw_ms = None
            print(f"[bench] raw all-gather baseline failed: {e}", file=sys.stderr)

    ms_per_step = elapsed / args.steps * 1e3
    act_bytes_rank = L * 16 * N * C * 2
    value = live * act_bytes_rank / (elapsed / args.steps) / 1e9

    out = {
        "metric": "residual_compressed_activation_exchange_throughput",
        "value": round(value, 3),
        "unit": "GB/s",
        "n_gpus": live,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {
            "workload": "FLUX.1-dev 1024x1024 ring-attention SP degree 8 (logical), 1-bit residual + error feedback: per rank per step "
                        f"{L} layers x (compress K,V + reconstruct 7 peers' K,V), shard (544,3072) fp16; {live} live rank(s), "
                        f"{W_LOGICAL - live} peer(s) looped back",
            "codec": "BINARY(1-bit, comp_rank=-1)", "layers": L, "shard": [N, C], "logical_ring": W_LOGICAL,
            "packet_bytes": pkt_bytes, "raw_bytes": N * C * 2,
        },
        "exchange_ms_per_step": round(ms_per_step, 4),
        "exchange_issued_by": exchange_mode,
        "replay": args.replay,
        "inorder_ms_per_step": None if inorder_ms is None else round(inorder_ms, 4),
        "exchange_stream": (["main", "side", "prio"][stream_mode] if (use_dist and step_plans is not None) else None),
        "layers_per_all_gather": (max(1, min(7, args.gather_group)) if (use_dist and step_plans is not None) else None),
        "raw_allgather_ms_per_step": None if raw_ms is None else round(raw_ms, 4),
        "speedup_vs_raw_allgather": None if raw_ms is None else round(raw_ms / ms_per_step, 3),
    }
    if live > 1:
        # wire side of the roofline pair (north star: "fraction of HBM / xGMI roofline"): packets RECEIVED per GPU per step
        # over the step time, against the xGMI links an all-gather among `live` GPUs can use (one link per peer, 7 at most;
        # ~153 GB/s per direction per link, MI355X_MICROARCH.md).  The exchange overlaps the codec work, so this is a lower
        # bound of the link rate actually reached while a collective is in flight.
        wire = (live - 1) * 2 * L * pkt_bytes
        links = min(live - 1, 7)
        out["xgmi"] = {"wire_bytes_per_gpu_per_step": int(wire), "achieved": round(wire / (ms_per_step * 1e-3) / 1e9, 2),
                       "peak": 153.0 * links, "unit": "GB/s", "frac": round(wire / (ms_per_step * 1e-3) / 1e9 / (153.0 * links), 4),
                       "links": links, "raw_bytes_per_gpu_per_step": int((live - 1) * 2 * L * N * C * 2)}
    if kern_ms is not None:
        if args.replay == "pipelined":
            # one steady-state launch = a unit of `ul` layers in each of the three groups: reconstruct 16*ul tensors (bits +
            # state in, state out: 4.125 B/el) + statistics/sign-bit pass of 2*ul tensors of the unit two ahead (x + state in,
            # bits out: 4.125 B/el); the finalize group's traffic is negligible
            ul = max(1, int(os.environ.get("CFX_PIPE_UNIT_LAYERS", "7")))
            if use_dist and step_plans is not None:
                Gq = max(1, min(7, args.gather_group))
                ul = max(Gq, (ul // Gq) * Gq)                   # units are whole all-gather groups
            ul = min(ul, 7, L)
            alg = (ALG_BYTES_PER_EL["decompress"] * 16 + 4.125 * 2) * ul * N * C
            kname = (f"k_binary_pipe (one launch = {ul} layers: dequant+add of {16 * ul} tensors x (544,3072) [own K,V error-feedback "
                     f"update + 7 peers' K,V per layer] + finalize of the next {ul} layers' K,V scales + stats/sign bits of the {ul} "
                     "layers after those)")
            pmc_key, csv_prefix = "k_binary_pipe_bytes_per_launch", "k_binary_pipe<true>"
        else:
            alg = ALG_BYTES_PER_EL["decompress"] * 16 * N * C
            kname = "k_binary_dequant (16 tensors x (544,3072) per launch: own K,V error-feedback update + 7 peers' K,V)"
            pmc_key, csv_prefix = "k_binary_dequant_bytes_per_launch", "k_binary_dequant"
        ach = alg / (kern_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": kname,
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                           "traffic": None, "traffic_source": None, "avg_launch_us": round(kern_ms * 1e3, 3), "algorithmic_bytes_per_launch": int(alg),
                           "event_samples": n_samples, "event_stride": args.event_stride}
        prof = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                out["roofline"]["traffic"] = json.load(open(prof)).get(pmc_key)
                out["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, round 1)"
            except Exception:
                pass
        # cross-reference: the committed rocprofv3 --kernel-trace --stats summary of this same command.  The HIP-event
        # bracket of a single dispatch includes ~1.3-2 us of marker-to-kernel gap (tools/evtest.hip), so `achieved` above
        # is the conservative figure.
        trace_json = os.path.join(REPO, "profiles", "r01_bench_kernel_durations.json")
        if os.path.exists(trace_json):
            try:
                ent = json.load(open(trace_json))["kernels"].get(csv_prefix)
                if ent:
                    out["roofline"]["avg_launch_us_rocprof"] = ent["avg_us"]
                    out["roofline"]["rocprof_source"] = ("profiles/r01_bench_kernel_durations.json (tools/trace_kernel_avg.py over the "
                                                         "rocprofv3 --kernel-trace of this command); the same kernel's row of "
                                                         "profiles/r01_bench_kernel_stats.csv (rocprofv3 --stats) agrees")
            except Exception:
                pass
    else:
        out["roofline"] = None
    if rank == 0 and live == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        except Exception as e:  # pragma: no cover
            out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        if not use_dist:
            # the oracle as the checker of what was just timed: replay every step this process ran (warm-up + timed +
            # in-order) for two tensors on the host and compare the error-feedback states bit for bit - fails loudly
            from oracle import c_oracle as CO
            import numpy as np
            steps_run = args.warmup + args.steps + (args.steps if inorder_ms is not None else 0)
            x0_host = warm_state(rank)[1]
            checked = []
            for l, kv in ((0, 0), (L - 1, 1)):
                state = x0_host[l, kv].cpu().numpy().view(np.uint16).copy()
                pk = np.zeros(pkt_bytes // 2, dtype=np.uint16)
                ins = [xs[s][l, kv].cpu().numpy() for s in range(2)]
                for t in range(steps_run):
                    CO.compress("binary", ins[t & 1], state, N, C, packet=pk, new_base=state)
                for name, got in (("sender state", own_base[l, kv]), ("looped-back peer state", peer_base[l, W_LOGICAL - 2, kv])):
                    if not np.array_equal(got.cpu().numpy().view(np.uint16), state):
                        raise RuntimeError(f"parity spot check failed: layer {l} {'KV'[kv]} {name} differs from the C oracle after {steps_run} steps")
                checked.append(f"layer {l} {'KV'[kv]}")
            out["cpu_baseline"]["parity_spot_check"] = (f"error-feedback states of {', '.join(checked)} (sender and a looped-back peer) after all "
                                                       f"{steps_run} steps of this run == C oracle replay, bit for bit")
    # tear the communicators down first and flush C stdio (RCCL prints a version banner through its own stdio buffer),
    # so that the JSON line is the LAST thing on stdout
    if native_comm is not None:
        try:
            torch.cuda.synchronize(dev)
            native_comm.close()
        except Exception:
            pass
    if use_dist:
        dist.destroy_process_group()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
