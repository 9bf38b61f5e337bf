"""bench.py: warm-up, the timed region, the validate-then-fall-back loop around them, and the secondary legs of the bench process.

The checks themselves (what "consistent" means, the ladder of schedules) are safety.py's; this module applies them to a Run: after the
warm-up steps and again after the timed region every rank validates, and on any failure every rank resets its states, rebuilds the step
on the ladder's next rung and runs warm-up + timed region again."""
from __future__ import annotations

import ctypes
import sys
import time

from . import safety
from .schedules import build_step_plans, one_step
from .workload import W_LOGICAL, GateTripped


def consistent(S):
    """This rank's part of the (collective) consistency check; synchronises the device first."""
    S.torch.cuda.synchronize(S.dev)
    return safety.states_consistent(S.torch, S.dist, S.own_base, S.peer_base, S.rank, S.live, S.real_live, S.G, W_LOGICAL - 1)


def check_run(S, label):
    """None when every rank is fine, else what tripped (the same on every rank): gate time-outs + states, safety.validate."""
    if not S.use_dist:
        return None
    S.torch.cuda.synchronize(S.dev)
    ge = S.lib.cfx_gate_errors(S.ctx) + (1 if S.gate_tripped else 0)   # (reads and clears the count; a step this rank refused to issue counts too)
    S.gate_tripped = False
    return safety.validate(S.torch, S.dist, label, S.use_dist, S.world, ge, lambda: consistent(S), S.dev)


def guarded_step(S, i) -> None:
    """one_step, except that a refused launch (an earlier wait of this rank timed out: CFX_ERR_GATE) does not end the run: this rank stops
    issuing steps until the next validation - its peers' waits run into their timeouts in turn and they stop as well - where every rank
    reports the time-out and the ladder takes the run down a rung.  (The schedules below the flag-ordered ones have no in-kernel waits:
    on them a step is never refused, so no rank ever leaves another alone inside a collective.)"""
    if S.gate_tripped:
        return
    try:
        one_step(S, i)
    except GateTripped as e:
        S.gate_tripped = True
        print(f"[bench] rank {S.rank}: step {i} refused ({e}); waiting for the validation", file=sys.stderr)


def current_rung(S) -> str:
    return "p2p" if S.exchange_mode == "p2p" else ("xgate" if S.xgate else ("native" if S.step_plans is not None else "torch"))


def fall_back(S, reason):
    """Every rank together: the next schedule down (safety.Ladder), then rebuild what that rung runs."""
    ladder = safety.Ladder(current_rung(S), S.schedule_fallback)
    if S.rank == 0:
        print(f"[bench] {ladder.NAMES[ladder.rung]} failed validation ({reason}); falling back in-process", file=sys.stderr)
    new = ladder.down(reason, S.native_comm is not None, S.world, S.stream_mode)
    for pl_ in (S.step_plans or []):
        S.lib.cfx_plan_destroy(pl_)
    S.step_plans = None
    S.xgate = S.one_launch = False
    S.args.own_ef = "ride"
    S.ride = True
    if new == "native":
        S.exchange_mode, S.stream_mode = "native", 0
        S.step_plans = build_step_plans(S, 0, xlayer=False)
    else:
        S.exchange_mode = "torch"
    S.schedule_fallback = ladder.text


def maybe_poison(S, step_no):
    """--poison-after-step (debug): what a stale line in a reader's cache would leave behind - a reconstruction that differs from its
    owner's state - planted once, while the peer-to-peer schedule runs."""
    if S.args.poison_after_step >= 0 and not S.poisoned[0] and S.exchange_mode == "p2p" and step_no == S.args.poison_after_step and S.rank == 0:
        S.torch.cuda.synchronize(S.dev)
        S.peer_base[0, 0, 0].view(S.torch.int16)[0, :8] += 1
        S.poisoned[0] = True


def timed_region(S) -> None:
    """Warm-up, validation, the timed K steps (barrier + synchronize on both sides, MAX over the ranks), validation; S.elapsed, S.kern_us."""
    S.n_warm = max(S.args.warmup, 1 if S.use_dist else 0)
    S.schedule_fallback = S.setup_fallback
    if S.exchange_mode == "torch":
        S.xgate = S.one_launch = False
        S.args.own_ef = "ride"
        S.ride = True
    S.KIDS, S.prof_cap = (), 0
    S.poisoned = [False]
    while True:
        S.reset_state()
        S.steps_run = 0
        S.gate_tripped = False
        S.sync_all()
        first_short = S.xgate and S.live > 1
        if first_short:
            # (libcfx's code object is loaded by its first launch - tens of milliseconds that would otherwise sit inside the first step's
            # 300 ms gates on some ranks and not on others)
            warm = S.torch.zeros(2, 4096, dtype=S.torch.uint8, device=S.dev)
            S.check(S.lib.cfx_copy_probe(S.ctx, warm[0].data_ptr(), warm[1].data_ptr(), 4096, S.sh), "copy_probe")
            S.sync_all()
            S.lib.cfx_set_gate_timeout_ms(S.ctx, 300)         # (the ranks enter the first step together: a gate that cannot open gives up quickly)
        why_bad = None
        for i in range(S.n_warm):
            guarded_step(S, i)
            maybe_poison(S, i)
            if i == 0 and first_short:
                S.sync_all()
                S.lib.cfx_set_gate_timeout_ms(S.ctx, 5000)
                if S.n_warm > 1:
                    # a transport that does not work at all (packets or flags that never arrive) shows in the very first step: say so now -
                    # the next step's launch would be refused on the rank that timed out while its peers wait 5 s per layer for it
                    why_bad = check_run(S, "in the first warm-up step")
                    if why_bad is not None:
                        break
        S.steps_run = S.n_warm
        S.sync_all()
        if why_bad is None:
            why_bad = check_run(S, "after the warm-up steps")
        if why_bad is not None:
            fall_back(S, why_bad)
            continue
        # profiled kernels: in-order replay: k_binary_dequant (4, launch B, dominant) and k_absmean_compress<bits> (27, launch A);
        # pipelined replay: the fused k_binary_pipe (23: full three-group launches; 24: prologue / epilogue / ragged launches)
        S.KIDS = (23, 24) if S.pipelined else ((31,) if S.one_launch else ((6, 28, 5) if S.int2 else (4, 27)))
        S.prof_cap = (S.args.steps * 2 * S.L) // max(1, S.args.event_stride) + 64
        if not S.args.no_kernel_events:
            mask = 0
            for k in S.KIDS:
                mask |= 1 << k
            S.check(S.lib.cfx_profile_enable(S.ctx, S.prof_cap, mask, S.args.event_stride), "profile_enable")
        S.sync_all()
        S.step_events = []          # gated schedule: hipEvents on the launch stream around every 4th step (every launch of a step is the
        t0 = time.perf_counter()  # same kernel, so elapsed / layers = its average duration with the kernel boundaries in)
        for i in range(S.args.steps):
            if S.one_launch and not S.args.no_kernel_events and i % 4 == 1:
                ea, eb = S.torch.cuda.Event(enable_timing=True), S.torch.cuda.Event(enable_timing=True)
                ea.record(S.compute)
                guarded_step(S, S.steps_run + i)
                eb.record(S.compute)
                S.step_events.append((ea, eb))
            else:
                guarded_step(S, S.steps_run + i)
            maybe_poison(S, S.steps_run + i)
        S.sync_all()
        t1 = time.perf_counter()
        S.steps_run += S.args.steps
        S.elapsed = t1 - t0
        why_bad = check_run(S, "after the timed region")
        if why_bad is not None:
            if not S.args.no_kernel_events:
                ids_ = (ctypes.c_int * S.prof_cap)()
                ms_ = (ctypes.c_float * S.prof_cap)()
                S.lib.cfx_profile_read(S.ctx, ids_, ms_, S.prof_cap)
                S.lib.cfx_profile_enable(S.ctx, 0, 0, 1)
            fall_back(S, why_bad)
            continue
        break
    if S.world > 1:
        t = S.torch.tensor([S.elapsed], device=S.dev, dtype=S.torch.float64)
        S.dist.all_reduce(t, op=S.dist.ReduceOp.MAX)
        S.elapsed = float(t.item())

    S.kern_us = {}
    if not S.args.no_kernel_events:
        ids = (ctypes.c_int * S.prof_cap)()
        ms = (ctypes.c_float * S.prof_cap)()
        n = S.lib.cfx_profile_read(S.ctx, ids, ms, S.prof_cap)
        for k in S.KIDS:
            vals = [ms[i] * 1e3 for i in range(n) if ids[i] == k and ms[i] > 0]
            if vals:
                S.kern_us[k] = (sum(vals) / len(vals), len(vals))
        S.lib.cfx_profile_enable(S.ctx, 0, 0, 1)




def timed_leg(S, n_steps, fn):
    S.sync_all()
    ta = time.perf_counter()
    for i in range(n_steps):
        fn(i)
    S.sync_all()
    dt = time.perf_counter() - ta
    if S.world > 1:
        tt = S.torch.tensor([dt], device=S.dev, dtype=S.torch.float64)
        S.dist.all_reduce(tt, op=S.dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt * 1e3 / n_steps


def secondary_legs(S) -> None:
    """(no events) a long run of the same replay, the other schedules on the same states - N = 1 only for the collective-free forms."""
    S.long_ms, S.other_ms, S.two_ms, S.loop_ms, S.relay_ms, S.part_ms, S.coll_ms = None, None, None, None, None, None, None
    if not S.args.no_secondary:
        base_step = S.steps_run
        if S.args.long_steps > 0:
            S.long_ms = timed_leg(S, S.args.long_steps, lambda i: guarded_step(S, base_step + i))
            S.steps_run += S.args.long_steps
            if S.gate_tripped:                    # (after two clean validations: a late rank; the line is the timed region's, this leg says nothing)
                S.long_ms, S.gate_tripped = None, False
                S.lib.cfx_gate_errors(S.ctx)

        def side_leg(plset, run_fn, what):
            """args.steps steps of a collective-free plan set on the same states (every replay advances them identically)."""
            b0 = S.steps_run
            fn = lambda i: S.check(run_fn(plset[(b0 + i) & 1], 0, S.lib.cfx_plan_size(plset[0]), S.sh), what)      # noqa: E731
            for i in range(2):
                fn(i)
            b0 += 2
            ms_ = timed_leg(S, S.args.steps, fn)
            S.steps_run += 2 + S.args.steps
            return ms_
        if S.real_live == 1 and not S.args.emulate_live:
            # (looped-back peers only) the cross-layer pipeline and the one-launch-per-layer form: neither can carry a collective
            if not S.int2 and not S.pipelined:
                S.other_ms = side_leg(S.plans_pipe, S.lib.cfx_plan_run_pipelined, "plan_run(pipelined)")
            if S.pipelined:
                S.other_ms = side_leg(S.plans_inorder, S.lib.cfx_plan_run, "plan_run(in order)")
            if S.plans_gated is not None and not S.gated:
                S.loop_ms = side_leg(S.plans_gated, S.lib.cfx_plan_run, "plan_run(one launch per layer, loop-back)")
            if S.gated:
                S.two_ms = side_leg(S.plans_inorder, S.lib.cfx_plan_run, "plan_run(two launches)")
            if S.xgate and S.exchange_mode in ("p2p", "native"):
                def step_leg(plset, what, stream_handle=S.sh):
                    b0 = S.steps_run
                    fn = lambda i: S.check(S.lib.cfx_plan_run(plset[(b0 + i) & 1], 0, S.lib.cfx_plan_size(plset[0]), stream_handle), what)      # noqa: E731
                    for i in range(2):
                        fn(i)
                    b0 += 2
                    ms_ = timed_leg(S, S.args.steps, fn)
                    S.steps_run += 2 + S.args.steps
                    for pl_ in plset:
                        S.lib.cfx_plan_destroy(pl_)
                    return ms_
                # the same exchange-layer launch with ncclAllGather in the path (flag-wait kernel ; ncclAllGather ; flag-set kernel on the exchange stream)
                if S.exchange_mode == "p2p" and S.native_comm is not None:
                    S.coll_ms = step_leg(build_step_plans(S, 0), "plan_run(exchange layer, ncclAllGather in the path)")
                # the same step, collective in the path, as two launches per layer in stream order (round 2's deployable schedule)
                S.two_ms = step_leg(build_step_plans(S, 0, xlayer=False), "plan_run(two launches, collective in the path)")
                # no communicator: the exchange stream only relays the flag (one kernel instead of wait ; ncclAllGather ; set)
                S.relay_ms = step_leg(build_step_plans(S, 0, comm_=False), "plan_run(exchange layer, flag relay)")
                # the configuration a run with MORE than one rank uses: run stream on CUs [0, 224), exchange stream on the other 32
                hm, hx = ctypes.c_void_p(), ctypes.c_void_p()
                assert S.lib.cfx_stream_create_masked(S.ctx, 0, 224, ctypes.byref(hm)) == 0 and S.lib.cfx_stream_create_masked(S.ctx, 224, 32, ctypes.byref(hx)) == 0
                S.torch.cuda.synchronize(S.dev)
                S.part_ms = step_leg(build_step_plans(S, 0, side_=hx.value), "plan_run(exchange layer, CU partition)", hm.value)
                S.torch.cuda.synchronize(S.dev)
                S.lib.cfx_stream_destroy(S.ctx, hm); S.lib.cfx_stream_destroy(S.ctx, hx)



def raw_exchange_legs(S) -> None:
    """The north-star comparison at N > 1: the UNCOMPRESSED exchange of the same K,V shards, and the compressed step in the other pattern."""
    # ---- the north-star comparison, N > 1: the UNCOMPRESSED exchange of the same K,V shards (reference patchpara/fwd.py:108-109,
    # ring.py:193-195) issued the same way as the compressed one - a native plan, one host call per step - as a direct all-gather
    # and as the reference's W-1-hop ring relay; and the compressed step in the OTHER exchange pattern -------------------------
    S.raw_legs, S.other_pattern_ms = {}, None
    if S.live > 1 and S.native_comm is not None and not S.pipelined and not S.args.no_raw_baseline:
        raw_in = S.xs[0]                                                       # [L, 2, N, C]: one layer's K,V = 2 x 3.3 MB per rank
        raw_buf = S.torch.empty(S.live, 2, S.N, S.C, dtype=S.torch.float16, device=S.dev)    # a layer's gathered K,V (consumed before the next layer's)
        raw_bytes = 2 * S.N * S.C * 2
        reps = max(3, min(S.args.steps, 10))
        for pattern in ("allgather", "relay"):
            rp = S.lib.cfx_plan_create(S.ctx)
            assert S.lib.cfx_plan_set_exchange_stream(rp, 0) == 0
            for l in range(S.L):
                if pattern == "allgather":
                    assert S.lib.cfx_plan_add_all_gather(rp, S.native_comm.handle, raw_in[l].data_ptr(), raw_buf.data_ptr(), raw_bytes) >= 0
                else:
                    src = raw_in[l].data_ptr()
                    for h in range(S.live - 1):
                        dst = raw_buf[(S.rank - h - 1) % S.live].data_ptr()
                        assert S.lib.cfx_plan_add_ring_hop(rp, S.native_comm.handle, src, dst, raw_bytes) >= 0
                        src = dst
            assert S.lib.cfx_plan_finalize(rp) == 0
            fnr = lambda i: S.check(S.lib.cfx_plan_run(rp, 0, S.lib.cfx_plan_size(rp), S.sh), "plan_run(raw " + pattern + ")")      # noqa: E731
            fnr(0); fnr(1)
            S.raw_legs[pattern] = timed_leg(S, reps, fnr)
            S.torch.cuda.synchronize(S.dev)
            S.lib.cfx_plan_destroy(rp)
        # the compressed step in the other pattern (same states: every replay advances them identically)
        if S.step_plans is not None and S.G == 1:
            op_plans = build_step_plans(S, 0, relay_=not S.relay)
            b0 = S.steps_run
            fno = lambda i: S.check(S.lib.cfx_plan_run(op_plans[(b0 + i) & 1], 0, S.lib.cfx_plan_size(op_plans[0]), S.sh), "plan_run(other pattern)")   # noqa: E731
            fno(0); fno(1)
            b0 += 2
            S.other_pattern_ms = timed_leg(S, S.args.steps, fno)
            S.steps_run += 2 + S.args.steps
            ok, why = consistent(S)
            assert ok, why
    S.raw_ms = S.raw_legs.get("relay" if S.relay else "allgather")

