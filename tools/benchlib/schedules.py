"""bench.py, part (b): the schedules - which launches a step consists of, as native plans.

  decide            what the command line's --own-ef / --p2p / --exchange select for this run (S.xgate, S.ride, S.gated, S.one_launch)
  build_local_plans the collective-free plan sets (in order, cross-layer pipeline, one gated launch per layer: the N = 1 forms and the legs of
                    the fall-back ladder's last rung)
  setup_streams     the run stream and the ONE exchange stream of the exchange-layer ops
  setup_exchange    libcfx's own communicator, the step as ONE native plan with the collective in it (build_step_plans), and - default - the
                    peer-to-peer exchange layer: packets in IPC-shared memory, read in place (build_p2p_plans)
  one_step          one denoise step of the current schedule
The plan builders only read the Run (workload.py); what is CHECKED after the steps lives in safety.py."""
from __future__ import annotations

import ctypes
import sys

from .workload import W_LOGICAL


def add_layer(S, plan, s_, l, ride, gathered, comm=None, gated=False, relay_=None, xlayer=False):
    """Layer l of the in-order schedule: A = compress (+ previous layer's own EF riding along), X = all-gather, B = reconstruct;
    gated (no X): one launch = A + the 16 reconstructions behind the arrival gate; xlayer: the exchange-layer op (collective in the path)."""
    relay_ = S.relay if relay_ is None else relay_
    if xlayer:
        # ONE op: compress + own EF ; all-gather ; reconstruct 14 - the reconstruction group launched with the compress group,
        # gated on the collective's arrival (cfx_plan_add_exchange_layer)
        assert gathered
        carr = S.comp_items(s_, l, True)
        for kv in range(2):
            carr[kv].new_base = S.own_base[l, kv].data_ptr()
        items = S.peer_items(l, True)
        rc = S.lib.cfx_plan_add_exchange_layer(plan, S.CODEC, S.N, S.C, 0, S._lib.FLAG_UPDATE_CACHE, 2, carr, len(items), (S._lib.DecompItem * len(items))(*items),
                                             comm, S.own_pkt_ptr(l, 0, True), S.grecv.data_ptr() + l * S.live * 2 * S.slot, 2 * S.slot, S.ws.data_ptr(), S.ws_bytes)
        assert rc >= 0, (rc, S.lib.cfx_last_error_string(S.ctx))
        return
    if gated:
        assert comm is None and not gathered
        # CFX_FLAG_UPDATE_CACHE = the rank's own error feedback in the same launch (1-bit: two more gated reconstructions; 2-bit:
        # the statistics workgroups quantise their own tiles from registers); the gated items are the 7 looped-back peers' K,V
        carr = S.comp_items(s_, l)
        for kv in range(2):
            carr[kv].new_base = S.own_base[l, kv].data_ptr()
        items = S.peer_items(l, False)
        rc = S.lib.cfx_plan_add_compress_gated(plan, S.CODEC, S.N, S.C, 0, S._lib.FLAG_UPDATE_CACHE, 2, carr, 0, None, len(items),
                                             (S._lib.DecompItem * len(items))(*items), S.ws.data_ptr(), S.ws_bytes)
        assert rc >= 0, (rc, S.lib.cfx_last_error_string(S.ctx))
        return
    if S.int2:
        # 2-bit: the codes depend on the scales, so compress = statistics + in-launch finalize, then quantise + error feedback
        # (in place on the rank's own state); the reconstruction launch carries the 7 peers' K,V
        carr = S.comp_items(s_, l, gathered)
        for kv in range(2):
            carr[kv].new_base = S.own_base[l, kv].data_ptr()
        rc = S.lib.cfx_plan_add_compress(plan, S.CODEC, S.N, S.C, 0, S._lib.FLAG_UPDATE_CACHE, 2, carr, S.ws.data_ptr(), S.ws_bytes)
        assert rc >= 0, (rc, S.lib.cfx_last_error_string(S.ctx))
        if comm is not None:
            assert S.lib.cfx_plan_add_all_gather(plan, comm, S.own_pkt_ptr(l, 0, True), S.grecv.data_ptr() + l * S.live * 2 * S.slot, 2 * S.slot) >= 0
        items = S.peer_items(l, gathered)
        assert S.lib.cfx_plan_add_decompress(plan, S.CODEC, S.N, S.C, 0, len(items), (S._lib.DecompItem * len(items))(*items)) >= 0
        return
    if ride and l > 0:
        rd = (S._lib.DecompItem * 2)(*S.own_ef_items(l - 1, gathered))
        rc = S.lib.cfx_plan_add_compress_ex(plan, S.CODEC, S.N, S.C, 0, 0, 2, S.comp_items(s_, l, gathered), 2, rd, S.ws.data_ptr(), S.ws_bytes)
    else:
        rc = S.lib.cfx_plan_add_compress(plan, S.CODEC, S.N, S.C, 0, 0, 2, S.comp_items(s_, l, gathered), S.ws.data_ptr(), S.ws_bytes)
    assert rc >= 0, (rc, S.lib.cfx_last_error_string(S.ctx))
    if comm is not None and relay_:
        # ring relay: hop h moves what arrived at hop h-1 (hop 0: our own packets) to rank+1; after hop h the region
        # [rank - h - 1] of the layer's receive area holds that rank's K,V packets - the same layout an all-gather leaves
        base_ptr = S.grecv.data_ptr() + l * S.live * 2 * S.slot
        src = S.own_pkt_ptr(l, 0, True)
        for h in range(S.live - 1):
            dst = base_ptr + ((S.rank - h - 1) % S.live) * 2 * S.slot
            rc = S.lib.cfx_plan_add_ring_hop(plan, comm, src, dst, 2 * S.slot)
            assert rc >= 0, rc
            src = dst
    elif comm is not None:
        rc = S.lib.cfx_plan_add_all_gather(plan, comm, S.own_pkt_ptr(l, 0, True), S.grecv.data_ptr() + l * S.live * 2 * S.slot, 2 * S.slot)
        assert rc >= 0, rc
    items = S.peer_items(l, gathered)
    if not ride or l == S.L - 1:
        items = S.own_ef_items(l, gathered) + items
    darr = (S._lib.DecompItem * len(items))(*items)
    rc = S.lib.cfx_plan_add_decompress(plan, S.CODEC, S.N, S.C, 0, len(items), darr)
    assert rc >= 0, (rc, S.lib.cfx_last_error_string(S.ctx))



def decide(S) -> None:
    # (2-bit: one launch per layer only in the peer-to-peer form, where the exchange runs inside the launch; beside an exchange stream's kernel
    # its layer launch is slower than three launches in stream order)
    S.xgate = (S.args.own_ef == "xgate" and not S.pipelined and (not S.int2 or S.args.p2p == "auto") and S.use_dist and not S.relay and S.args.exchange == "native")
    if S.args.own_ef == "xgate" and not S.xgate:
        S.args.own_ef = "ride"
    S.ride = S.args.own_ef in ("ride", "gated")
    S.gated = S.args.own_ef == "gated" and not S.pipelined and not S.use_dist
    S.one_launch = S.gated or S.xgate



def build_local_plans(S) -> None:
    # ---- native plans without collectives (one per input set) -----------------------------------------------------------------
    #   inorder:   per layer  A(l) [+ EF(l-1)] ; B(l)                        (ops 2l, 2l+1)
    #   pipelined: per layer  compress(l) ; reconstruct own + peers (16)     (the op pattern cfx_plan_run_pipelined recognises)
    #   gated:     per layer  ONE launch: A(l) + own EF(l) + B(l) behind the gate (op l)
    def _build(kind):
        built = []
        for s_ in range(2):
            plan = S.lib.cfx_plan_create(S.ctx)
            for l in range(S.L):
                add_layer(S, plan, s_, l, S.ride if kind == "inorder" else False, False, gated=(kind == "gated"))
            assert S.lib.cfx_plan_finalize(plan) == 0
            built.append(plan)
        return built
    S.plans_inorder = _build("inorder")
    S.plans_pipe = None if S.int2 else _build("pipelined")
    S.plans_gated = _build("gated") if (S.gated or (S.real_live == 1 and not S.args.emulate_live and not S.pipelined and not S.args.no_secondary)) else None
    S.plans = S.plans_pipe if S.pipelined else (S.plans_gated if S.gated else S.plans_inorder)




def setup_streams(S) -> None:
    S.xside = None
    if S.xgate:
        # the exchange-layer op orders its two streams by flag words: the run stream must not be the legacy NULL stream (it serialises
        # with every blocking stream, the CU-masked exchange stream included)
        if S.args.same_gpu and S.world > 1:
            # debug: the ranks share one GPU - a waiting layer launch of one rank must not hold the CUs another rank's compress group needs
            hm = ctypes.c_void_p()
            share = 256 // S.world
            assert S.lib.cfx_stream_create_masked(S.ctx, share * S.rank, share, ctypes.byref(hm)) == 0
            S.torch.cuda.set_stream(S.torch.cuda.ExternalStream(hm.value, device=S.dev))
        else:
            S.torch.cuda.set_stream(S.torch.cuda.Stream(S.dev))
        hx = ctypes.c_void_p()
        assert S.lib.cfx_stream_create_masked(S.ctx, 0, 256, ctypes.byref(hx)) == 0      # ONE exchange stream for every plan: each stream is a hardware queue
        S.xside = hx.value
    S.compute = S.torch.cuda.current_stream(S.dev)
    S.sh = S.compute.cuda_stream




# ---- N > 1: the whole step as ONE native plan, the all-gathers issued by libcfx's own RCCL communicator -----------------
#   inorder:   A(l) ; all-gather(l) ; B(l)   layer by layer, everything in order on the compute stream
#   pipelined: --gather-group layers share one all-gather and form one unit of the pipelined replay; --exchange-stream
#              prio|side runs the collective of unit u on an exchange stream underneath the next fused launch
def build_step_plans(S, mode, relay_=None, xlayer=None, comm_=True, side_=None):
    xlayer = (S.xgate if xlayer is None else xlayer) and not (S.relay if relay_ is None else relay_)
    side_ = side_ or S.xside
    built = []
    for s_ in range(2):
        sp = S.lib.cfx_plan_create(S.ctx)
        if xlayer and side_:
            assert S.lib.cfx_plan_use_exchange_stream(sp, side_) == 0
        elif not xlayer:
            assert S.lib.cfx_plan_set_exchange_stream(sp, mode) == 0
        if not S.pipelined:
            for l in range(S.L):
                add_layer(S, sp, s_, l, S.ride or S.xgate, True, S.native_comm.handle if comm_ else None, relay_=relay_,
                          xlayer=xlayer)
        else:
            for a, b in S.groups:
                for l in range(a, b):
                    assert S.lib.cfx_plan_add_compress(sp, S.CODEC, S.N, S.C, 0, 0, 2, S.comp_items(s_, l, True), S.ws.data_ptr(), S.ws_bytes) >= 0
                rcx = S.lib.cfx_plan_add_all_gather(sp, S.native_comm.handle, S.own_pkt_ptr(a, 0, True),
                                                  S.grecv.data_ptr() + a * S.live * 2 * S.slot, (b - a) * 2 * S.slot)
                assert rcx >= 0, rcx
                for l in range(a, b):
                    items = S.own_ef_items(l, True) + S.peer_items(l, True)
                    assert S.lib.cfx_plan_add_decompress(sp, S.CODEC, S.N, S.C, 0, 16, (S._lib.DecompItem * 16)(*items)) >= 0
        assert S.lib.cfx_plan_finalize(sp) == 0
        built.append(sp)
    return built



def build_p2p_plans(S):
    built = []
    for s_ in range(2):
        sp = S.lib.cfx_plan_create(S.ctx)
        assert S.lib.cfx_plan_use_exchange_stream(sp, S.xside) == 0
        for l in range(S.L):
            carr = (S._lib.CompItem * 2)()
            for kv in range(2):
                carr[kv] = S._lib.CompItem(S.xs[s_][l, kv].data_ptr(), S.own_base[l, kv].data_ptr(), S.own_base[l, kv].data_ptr(),
                                         S.p2p_ptr.value + (l * 2 + kv) * S.slot)
            items = []
            for p in range(W_LOGICAL - 1):
                real = p < S.world - 1
                src = S.p2p_peer[(S.rank + 1 + p) % S.world] if real else S.p2p_ptr.value        # a looped-back logical peer reads OUR packets
                for kv in range(2):
                    items.append(S._lib.DecompItem(src + (l * 2 + kv) * S.slot, S.peer_base[l, p, kv].data_ptr(), S.peer_base[l, p, kv].data_ptr()))
            pf = (ctypes.c_void_p * max(1, S.world - 1))(*[S.p2p_peer[q] + S.p2p_flags_off + (s_ * S.L + l) * 64 for q in sorted(S.p2p_peer)])
            rc_ = S.lib.cfx_plan_add_exchange_layer_p2p(sp, S.CODEC, S.N, S.C, 0, S._lib.FLAG_UPDATE_CACHE, 2, carr, len(items),
                                                      (S._lib.DecompItem * len(items))(*items), S.p2p_ptr.value + S.p2p_flags_off + (s_ * S.L + l) * 64,
                                                      S.world - 1, pf, S.ws.data_ptr(), S.ws_bytes)
            assert rc_ >= 0, (rc_, S.lib.cfx_last_error_string(S.ctx))
        assert S.lib.cfx_plan_finalize(sp) == 0
        built.append(sp)
    return built



def setup_exchange(S) -> None:
    S.native_comm, S.step_plans, S.exchange_mode, S.stream_mode = None, None, "none", 0
    S.setup_fallback = None
    S.p2p_ptr, S.p2p_peer = None, {}
    if S.use_dist:
        S.exchange_mode = "torch"
        if S.args.exchange == "torch" and S.world == 1:
            raise SystemExit("--exchange torch needs N > 1 (torch.distributed is not initialised for one rank)")
        if S.args.exchange == "native":
            try:
                from compactfusion_amd.exchange import NativeComm
                try:
                    S.native_comm = NativeComm(S.local_rank, solo_ranks=S.live if S.world == 1 else 0, library=S.args.rccl_lib)
                    if not S.args.emulate_live:
                        S.native_comm.self_test()
                except Exception as e_comm:
                    # --same-gpu (debug): RCCL refuses two ranks on one device; the peer-to-peer exchange needs no collective library
                    if not (S.args.same_gpu and S.world > 1 and S.xgate and S.args.p2p == "auto"):
                        raise
                    S.native_comm = None
                    if S.rank == 0:
                        print(f"[bench] no collective library here ({e_comm}); peer-to-peer exchange only, raw-exchange legs skipped", file=sys.stderr)
                S.groups = [(a, min(S.L, a + S.G)) for a in range(0, S.L, S.G)]

                S.stream_mode = {"main": 0, "side": 1, "prio": 2}[S.args.exchange_stream] if S.pipelined else 0
                if S.native_comm is not None:
                    S.step_plans = build_step_plans(S, S.stream_mode)
                    S.exchange_mode = "native"
                if S.xgate and S.args.p2p == "auto" and not S.args.emulate_live:
                    # ---- no collective at all: packets stay in IPC-shared memory, the peers read them in place ---------------------------
                    S.p2p_flags_off = flags_off = S.L * 2 * S.slot
                    S.p2p_ptr, p2p_handle = ctypes.c_void_p(), ctypes.create_string_buffer(64)
                    check_rc = S.lib.cfx_ipc_alloc(S.ctx, flags_off + 2 * S.L * 64, ctypes.byref(S.p2p_ptr), p2p_handle)
                    ok_all = S.torch.tensor([1 if check_rc == 0 else 0], device=S.dev, dtype=S.torch.int32)
                    if S.world > 1:
                        S.dist.all_reduce(ok_all, op=S.dist.ReduceOp.MIN)
                    if int(ok_all.item()) == 1:
                        handles = [bytes(p2p_handle.raw)]
                        if S.world > 1:
                            handles = [None] * S.world
                            S.dist.all_gather_object(handles, bytes(p2p_handle.raw))
                        S.p2p_peer = {}
                        opened = 1
                        for q in range(S.world):
                            if q != S.rank:
                                pq = ctypes.c_void_p()
                                if S.lib.cfx_ipc_open(S.ctx, handles[q], ctypes.byref(pq)) != 0:
                                    opened = 0
                                    break
                                S.p2p_peer[q] = pq.value
                        ok_all = S.torch.tensor([opened], device=S.dev, dtype=S.torch.int32)
                        if S.world > 1:
                            S.dist.all_reduce(ok_all, op=S.dist.ReduceOp.MIN)
                    if int(ok_all.item()) == 1:
                        for pl_ in (S.step_plans or []):
                            S.lib.cfx_plan_destroy(pl_)
                        S.step_plans = build_p2p_plans(S)
                        S.exchange_mode = "p2p"
                    elif S.rank == 0:
                        print("[bench] IPC-shared packet buffers unavailable; the collective stays in the path (ncclAllGather on the exchange stream)", file=sys.stderr)
            except Exception as e:  # pragma: no cover
                if S.world == 1:
                    raise SystemExit(f"[bench] native exchange unavailable ({e})")
                # never end a multi-rank run while a collective fall-back exists: torch.distributed per layer (the line says so)
                print(f"[bench] native exchange unavailable ({e}); FALLBACK to torch.distributed per layer", file=sys.stderr)
                S.native_comm, S.step_plans, S.exchange_mode = None, None, "torch"
                S.setup_fallback = f"the native exchange could not be set up ({e}); torch.distributed per layer instead"



def one_step(S, step):
    run_native = S.lib.cfx_plan_run_pipelined if S.pipelined else S.lib.cfx_plan_run
    plan = S.plans[step & 1]
    if not S.use_dist:
        S.check(run_native(plan, 0, S.lib.cfx_plan_size(plan), S.sh), "plan_run")     # the whole step from native code
        return
    if S.step_plans is not None:
        sp = S.step_plans[step & 1]
        S.check(run_native(sp, 0, S.lib.cfx_plan_size(sp), S.sh), "plan_run(exchange)")
        return
    # torch.distributed per layer (fallback / --exchange torch): layer by layer in order
    pl = S.plans_inorder[step & 1]
    for l in range(S.L):
        S.check(S.lib.cfx_plan_run(pl, 2 * l, 1, S.sh), "compress")
        S.dist.all_gather_into_tensor(S.recv[l].view(-1), S.send[l].view(-1))
        S.check(S.lib.cfx_plan_run(pl, 2 * l + 1, 1, S.sh), "reconstruct")

