"""bench.py, part (d): the JSON line - value, the schedule's description, the secondary figures, `xgmi`, `roofline` (+ the committed profiles,
quoted only on a matching configuration key and source hash), `cpu_baseline` with the oracle replay of every step the process ran."""
from __future__ import annotations

import ctypes
import json
import os
import sys

from .workload import HBM_PEAK_GBS, REPO, W_LOGICAL, config_key, cpu_baseline


def build_line(S) -> dict:
    """The contract line of this run (rank 0 prints it)."""
    ms_per_step = S.elapsed / S.args.steps * 1e3
    act_bytes_rank = S.L * 16 * S.N * S.C * 2
    value = S.real_live * act_bytes_rank / (S.elapsed / S.args.steps) / 1e9
    inorder_ms = S.other_ms if S.pipelined else ms_per_step
    pipe_ms = ms_per_step if S.pipelined else S.other_ms

    P2P = S.exchange_mode == "p2p"
    XNAME = ("no collective (--no-collective)" if not S.use_dist else
             ("ring relay: " + str(S.live - 1) + " grouped ncclSend/ncclRecv hops" if S.relay else "ncclAllGather, in place (packets are written straight into the rank's slot of the gather buffer)")
             + f" over libcfx's own {'loop-back stand-in' if S.args.emulate_live else 'RCCL'} communicator of {S.live} rank(s), issued from the native plan")
    out = {
        "metric": "residual_compressed_activation_exchange_throughput",
        "value": round(value, 3),
        "unit": "GB/s",
        "n_gpus": S.real_live,
        "steps": S.args.steps,
        "warmup": S.args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {
            "workload": f"FLUX.1-dev 1024x1024 ring-attention SP degree 8 (logical), {'2' if S.int2 else '1'}-bit residual + error feedback: per rank per step "
                        f"{S.L} layers x (compress K,V + reconstruct 7 peers' K,V), shard (544,3072) fp16, layer by layer in order; "
                        f"{S.live} live rank(s), {W_LOGICAL - S.live} peer(s) looped back",
            "codec": "INT2(2-bit fastpath)" if S.int2 else "BINARY(1-bit, comp_rank=-1)", "layers": S.L, "shard": [S.N, S.C], "logical_ring": W_LOGICAL,
            "packet_bytes": S.pkt_bytes, "raw_bytes": S.N * S.C * 2,
        },
        "exchange_ms_per_step": round(ms_per_step, 4),
        "exchange_issued_by": S.exchange_mode,
        "exchange_pattern": (S.args.exchange_pattern if S.use_dist else None),
        "replay": S.args.replay,
        "schedule": ("cross-layer software pipeline (NOT deployable: needs every layer's K,V resident)" if S.pipelined else
                     ("layer by layer in order (deployable): ONE launch per layer, two groups of workgroups and two arrival gates: statistics + "
                      "in-launch finalize of own K,V, then every statistics workgroup quantises its own tile (+ error feedback) from the registers "
                      "it loaded -> reconstruction of the 7 looped-back peers' K,V (state tiles already in registers)") if (S.int2 and S.gated) else
                     ("layer by layer in order (deployable): per layer A1 = statistics + in-launch finalize of own K,V, A2 = quantise + error "
                      "feedback, X = " + XNAME + ", B = reconstruct 7 peers' K,V") if S.int2 else
                     ("layer by layer in order, LOOP-BACK ONLY (no collective can sit inside it): ONE launch per layer = compress K,V [statistics + sign bits + in-launch "
                      "finalize] + the 16 reconstructions its packets feed (own error feedback, 7 looped-back peers' K,V): their workgroups "
                      "pull the state tiles into registers while the scale reduction completes, wait on an arrival gate, finish from registers") if S.gated else
                     ("layer by layer in order (deployable), NO collective: every rank's packets stay in IPC-shared memory of its own GPU, the peers' "
                      "reconstruction workgroups read them in place over xGMI.  Per layer ONE codec launch on the run stream = compress K,V [statistics + "
                      "sign bits + in-launch finalize] + own error-feedback update + reconstruction of the 7 peers' K,V, whose workgroups pull their state "
                      "tiles into registers and then wait for a gate word; workgroup 0 of the same launch waits for the launch's packets, publishes "
                      f"a word the {S.live - 1} live peer(s) have mapped, waits for their words and opens the gate (cfx_plan_add_exchange_layer_p2p): no second launch, no second stream.  " +
                      ("One live rank: no peer to read from or to wait for - the same op, launch and kernel as any N, minus the remote reads "
                       "(`collective_in_the_path`: the same launch around ncclAllGather)" if S.live == 1 else
                       "Validated after the warm-up steps and again after the timed region (gate time-outs, every rank's reconstruction of a shard against its owner's state)")) if (S.xgate and P2P) else
                     ("layer by layer in order (deployable), the collective in the path: per layer ONE codec launch on the run stream = compress K,V "
                      "[statistics + sign bits + in-launch finalize; packets written straight into the rank's slot of the gather buffer] + own "
                      "error-feedback update + reconstruction of the 7 peers' K,V, whose workgroups pull their state tiles into registers and "
                      "then wait for a gate word; on the exchange stream: flag-wait kernel (this launch's packets complete) ; X = " + XNAME +
                      " ; flag-set kernel (opens the gate).  " +
                      ("One live rank: the collective enqueues no kernel." if S.live == 1 else
                       "More than one live rank: the collective is a kernel that is placed beside the waiting workgroups (the reconstruction "
                       "group leaves >= 32 workgroup slots free); validated after the warm-up steps (first step: 300 ms gate timeout) and after the timed region")) if S.xgate else
                     "layer by layer in order (deployable): per layer A = compress K,V [statistics + sign bits + in-launch finalize"
                     + (" + previous layer's own error-feedback update riding along" if S.ride else "") + "], X = " + XNAME + ", B = reconstruct "
                     + ("7 peers' K,V" if S.ride else "own + 7 peers' K,V")),
        "schedule_fallback": S.schedule_fallback,
        "launches_per_layer": None if S.pipelined else (1 if S.one_launch else (3 if S.int2 else 2)),
        "two_launches_per_layer": None if S.two_ms is None else {
            "ms_per_step": round(S.two_ms, 4),
            "what": ("the same layer-ordered step as A1 = statistics + finalize ; A2 = quantise + error feedback ; B = reconstruct 7 peers" if S.int2 else
                     "the same layer-ordered step as A = compress (+ previous layer's own error feedback riding along) ; B = reconstruct 7 peers")
                    + (" ; the collective between them, everything in stream order (the fall-back schedule)" if S.xgate else
                       " - the schedule a collective between compress and reconstruction forces (N > 1)")},
        "collective_in_the_path": None if S.coll_ms is None else {
            "ms_per_step": round(S.coll_ms, 4),
            "what": "the same exchange-layer launch with a collective library in the path: flag-wait kernel ; ncclAllGather (in place, libcfx's own RCCL "
                    "communicator of this many ranks) ; flag-set kernel on the exchange stream - round 3's earlier default, `--p2p off`"},
        "flag_relay_no_communicator": None if S.relay_ms is None else {
            "ms_per_step": round(S.relay_ms, 4),
            "what": "the same exchange-layer plans built WITHOUT a communicator: the exchange stream runs one relay kernel per layer (wait + set) instead of "
                    "flag-wait kernel ; ncclAllGather ; flag-set kernel - what the two kernel boundaries around the collective cost, and the launch "
                    "structure of the peer-to-peer exchange layer runs with N > 1 use (there the one kernel also publishes a word and waits for the peers')"},
        "with_cu_partition": None if S.part_ms is None else {
            "ms_per_step": round(S.part_ms, 4),
            "what": "the same exchange-layer plans with the run stream masked to CUs [0, 224) and the exchange stream to [224, 256): CUs of its own for a "
                    "collective kernel whatever the shape; any partial CU mask costs this launch ~5 us, so the streams are not partitioned (the "
                    "reconstruction group of this shape leaves 32 workgroup slots free, which is room enough)"},
        "inorder_ms_per_step": None if inorder_ms is None else round(inorder_ms, 4),
        "pure_exchange_upper_bound": None if pipe_ms is None else {
            "ms_per_step": round(pipe_ms, 4),
            "what": "cfx_plan_run_pipelined: statistics / finalize of later layers run beside the reconstruction of earlier ones; "
                    "legal only with every layer's K,V resident before the step (this bench's synthetic inputs) - a model cannot run it"},
        "long_run": None if S.long_ms is None else {"steps": S.args.long_steps, "ms_per_step": round(S.long_ms, 4)},
        "exchange_stream": (["main", "side", "prio"][S.stream_mode] if (S.use_dist and S.step_plans is not None) else None),
        "layers_per_all_gather": (S.G if (S.use_dist and S.step_plans is not None) else None),
        "raw_allgather_ms_per_step": None if S.raw_ms is None else round(S.raw_ms, 4),
        "raw_exchange_ms_per_step": {k_: round(v_, 4) for k_, v_ in S.raw_legs.items()} or None,
        "speedup_vs_raw_allgather": None if S.raw_ms is None else round(S.raw_ms / ms_per_step, 3),
        "loopback_one_launch_per_layer": None if S.loop_ms is None else {
            "ms_per_step": round(S.loop_ms, 4),
            "what": "cfx_compress_batch_gated: the layer as ONE launch (reconstruction behind an in-launch arrival gate). Exists only when the "
                    "packets a reconstruction needs are produced by the same launch - looped-back peers, no collective - so it is NOT what N > 1 runs"},
    }
    if S.live > 1:
        # wire side of the roofline pair (north star: "fraction of HBM / xGMI roofline"): bytes RECEIVED per GPU per step over the step
        # time, against the xGMI links the pattern can use: a direct all-gather among `live` GPUs one link per peer (7 at most), the ring
        # relay ONE link (every hop receives from rank-1); ~153 GB/s per direction per link (MI355X_MICROARCH.md).  The compressed
        # exchange shares its step with the codec launches, so its figure is a lower bound of the link rate while a collective is in flight.
        def xg(wire, ms_, pattern):
            links = 1 if pattern == "relay" else min(S.live - 1, 7)
            if ms_ is None:
                return None
            o = {"ms_per_step": round(ms_, 4), "achieved": round(wire / (ms_ * 1e-3) / 1e9, 2), "peak": 153.0 * links, "unit": "GB/s", "links": links}
            # a fraction of a LINK roofline only where links carried the bytes: over the loop-back library the "wire" is a device copy, its
            # rate says nothing about xGMI and may exceed the link peak - no `frac` key there
            if S.args.emulate_live:
                o["loopback_device_copy"] = True
            elif S.args.same_gpu:
                o["same_gpu"] = True             # rank processes sharing ONE GPU (protocol test): the peers' packets are read from the same HBM
            else:
                o["frac"] = round(wire / (ms_ * 1e-3) / 1e9 / (153.0 * links), 4)
            return o
        wire = (S.live - 1) * 2 * S.L * S.pkt_bytes
        wire_raw = (S.live - 1) * 2 * S.L * S.N * S.C * 2
        this_p, other_p = ("relay", "allgather") if S.relay else ("allgather", "relay")
        out["xgmi"] = dict(xg(wire, ms_per_step, this_p), wire_bytes_per_gpu_per_step=int(wire), raw_bytes_per_gpu_per_step=int(wire_raw),
                           pattern=this_p,
                           compressed={this_p: xg(wire, ms_per_step, this_p), other_p: xg(wire, S.other_pattern_ms, other_p)},
                           raw={k_: xg(wire_raw, v_, k_) for k_, v_ in S.raw_legs.items()},
                           issued_by="every leg is one native plan per step (cfx_plan_run): no Python-issued collective on either side")
        if S.args.emulate_live:
            out["xgmi"]["note"] = "--emulate-live: loop-back collective library on ONE GPU - device copies, not xGMI links; layout and plumbing only"
        elif S.args.same_gpu:
            out["xgmi"]["note"] = "--same-gpu: the rank processes share ONE GPU - no link carried a byte; protocol and plumbing only"
    # ---- roofline --------------------------------------------------------------------------------------------------------
    # step level (every launch of the step, edge layers included), SURVEY.md §8d: own tensors compress + error feedback 6.125 B/el,
    # peers' tensors 4.125 B/el
    EL = S.N * S.C
    step_alg = S.L * (2 * S.alg["compress"] + 14 * S.alg["decompress"]) * EL
    step_obj = {"algorithmic_bytes": int(step_alg), "achieved": round(step_alg / (ms_per_step * 1e-3) / 1e9, 1), "unit": "GB/s",
                "frac": round(step_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "floor_ms_at_peak": round(step_alg / (HBM_PEAK_GBS * 1e9) * 1e3, 4)}
    dom = 23 if S.pipelined else (31 if S.one_launch else (6 if S.int2 else 4))
    if dom in S.kern_us:
        us, n_samples = S.kern_us[dom]
        if S.pipelined:
            ul = 7                                              # cfx_plan_set_pipe_unit_layers default
            if S.use_dist and S.step_plans is not None:
                ul = max(S.G, (ul // S.G) * S.G)                      # units are whole all-gather groups
            ul = min(ul, 7, S.L)
            # one steady-state launch: reconstruct 14*ul peers' tensors (4.125) + own 2*ul tensors' error-feedback pass and, two units
            # ahead, their statistics pass: together the own tensors' compress + EF = 6.125 B/el (the second read of x / state is
            # implementation traffic)
            alg = (S.alg["decompress"] * 14 + S.alg["compress"] * 2) * ul * EL
            kname = (f"k_binary_pipe (one launch = {ul} layers: dequant+add of {16 * ul} tensors x (544,3072) + finalize of the next {ul} "
                     f"layers' K,V scales + stats/sign bits of the {ul} layers after those)")
        elif S.xgate:
            alg = (S.alg["compress"] * 2 + S.alg["decompress"] * 14) * EL
            kname = (("k_int2_compress_gated" if S.int2 else "k_absmean_compress<bits,gated>") + " (the layer's only codec launch: compress + error feedback of own K,V at " + str(S.alg["compress"]) +
                     " B/el, 7 peers' K,V at " + str(S.alg["decompress"]) + " B/el; between reading K,V and the first reconstructed byte sit a global "
                     "reduction - the scales - and the collective's arrival)")
        elif S.gated:
            alg = (S.alg["compress"] * 2 + S.alg["decompress"] * 14) * EL
            kname = (("k_int2_compress_gated" if S.int2 else "k_absmean_compress<bits,gated>") + " (the layer's only launch: compress + error feedback of own K,V at " + str(S.alg["compress"]) + " B/el, "
                     "7 looped-back peers' K,V at " + str(S.alg["decompress"]) + " B/el; a global reduction - the scales - sits between reading K,V and the first "
                     "reconstructed byte)")
        elif S.int2:
            alg = S.alg["decompress"] * 14 * EL
            kname = "k_int2_dequant (launch B: 7 peers K,V = 14 tensors x (544,3072) per launch)"
        else:
            n_t = (14 * (S.L - 1) + 16) / S.L if S.ride else 16.0     # tensors per launch B, averaged over the step's launches
            alg = S.alg["decompress"] * n_t * EL
            kname = (f"k_binary_dequant (launch B: {'7 peers K,V = 14' if S.ride else 'own + 7 peers K,V = 16'} tensors x (544,3072) per launch"
                     + ("; the last layer's launch carries 16" if S.ride else "") + ")")
        ach = alg / (us * 1e-6) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": kname,
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                           "traffic": None, "traffic_source": None, "avg_launch_us": round(us, 3), "algorithmic_bytes_per_launch": int(alg),
                           "event_samples": n_samples, "event_stride": S.args.event_stride, "step": step_obj}
        if S.one_launch and S.step_events:
            # one launch per layer: hipEvents around whole steps / the launches of a step = the launch duration with the kernel
            # boundaries in (what rocprofv3's per-kernel durations add up to); a dispatch that itself carries profiling events runs
            # ~1.5 us longer on this kernel, so the roofline uses the step-bracketing events and keeps the other figure beside it
            us_ev = sum(a_.elapsed_time(b_) for a_, b_ in S.step_events) * 1e3 / len(S.step_events) / S.L
            ach2 = alg / (us_ev * 1e-6) / 1e9
            out["roofline"].update({"avg_launch_us_dispatch_events": out["roofline"]["avg_launch_us"], "avg_launch_us": round(us_ev, 3),
                                    "achieved": round(ach2, 1), "frac": round(ach2 / HBM_PEAK_GBS, 4),
                                    "event_samples": len(S.step_events) * S.L,
                                    "event_method": "hipEvents on the launch stream around every 4th step of the timed region / launches per step"})
        if S.int2 and 28 in S.kern_us and 5 in S.kern_us:
            out["roofline"]["compress_launches"] = {
                "k_absmean_compress (A1: statistics + in-launch finalize)": round(S.kern_us[28][0], 3),
                "k_int2_quant (A2: codes + error feedback, own K,V)": round(S.kern_us[5][0], 3), "unit": "us",
                "algorithmic_bytes_per_layer": int(S.alg["compress"] * 2 * EL)}
        if not S.pipelined and 27 in S.kern_us:
            usa, na = S.kern_us[27]
            # launch A: the rank's own K,V - compress now, error feedback of the previous layer riding along: 6.125 B/el algorithmic
            alga = S.alg["compress"] * 2 * EL if S.ride else 4.125 * 2 * EL
            out["roofline"]["compress_launch"] = {
                "kernel": "k_absmean_compress<bits> (launch A: statistics + sign bits + in-launch finalize of own K,V"
                          + (" + previous layer's own error-feedback update" if S.ride else "") + ")",
                "avg_launch_us": round(usa, 3), "event_samples": na, "algorithmic_bytes_per_launch": int(alga),
                "achieved": round(alga / (usa * 1e-6) / 1e9, 1), "frac": round(alga / (usa * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "latency-bound: a global reduction (scales) sits between reading K,V and the packet being complete"}
        # PMC traffic / rocprof cross-reference: only when the committed profile was taken with THIS configuration
        # ... AND from this tree's kernel sources (tools/provenance.py): a stale profile is not quoted
        prof = os.path.join(REPO, "profiles", "r06_pmc_traffic.json")
        cfg_key = config_key(S.args, S.live)
        sys.path.insert(0, os.path.join(REPO, "tools"))
        from provenance import source_sha
        src_sha = source_sha()
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof))
                if pj.get("config") == cfg_key and pj.get("source_sha") != src_sha:
                    out["roofline"]["traffic_source"] = "profiles/r06_pmc_traffic.json was taken from other kernel sources (source_sha differs): not quoted"
                if pj.get("config") == cfg_key and pj.get("source_sha") == src_sha:
                    pk_ = "k_binary_pipe<true>" if S.pipelined else (("k_int2_compress_gated" if S.int2 else "k_absmean_compress<true, 4, true") if S.one_launch else "k_binary_dequant")
                    out["roofline"]["traffic"] = next((v for k_, v in pj["bytes_per_launch"].items() if k_.startswith(pk_)), None)
                    out["roofline"]["traffic_source"] = ("profiles/r06_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes" +
                                                         ("; " + pj["measured_with"] + ")" if pj.get("measured_with") else " of this command)"))
                    if pj.get("measured_with"):
                        out["roofline"]["step"]["traffic_source"] = "the same counter passes (loop-back form of the step: no flag kernels, no collective call)"
                    out["roofline"]["step"]["traffic"] = pj.get("bytes_per_step")
            except Exception:
                pass
        trace_json = os.path.join(REPO, "profiles", "r06_bench_kernel_durations.json")
        if os.path.exists(trace_json):
            try:
                tj = json.load(open(trace_json))
                if tj.get("config") == cfg_key and tj.get("source_sha") != src_sha:
                    out["roofline"]["rocprof_source"] = "profiles/r06_bench_kernel_durations.json was taken from other kernel sources (source_sha differs): not quoted"
                if tj.get("config") == cfg_key and tj.get("source_sha") == src_sha:
                    pk_ = "k_binary_pipe<true>" if S.pipelined else (("k_int2_compress_gated" if S.int2 else "k_absmean_compress<true, 4, true") if S.one_launch else "k_binary_dequant")
                    ent = next((v for k_, v in tj["kernels"].items() if k_.startswith(pk_)), None)
                    if ent:
                        out["roofline"]["avg_launch_us_rocprof"] = ent["avg_us"]
                        out["roofline"]["median_launch_us_rocprof"] = ent.get("median_us")
                        out["roofline"]["rocprof_source"] = "profiles/r06_bench_kernel_durations.json (rocprofv3 --kernel-trace of this command)"
            except Exception:
                pass
    else:
        out["roofline"] = {"bound": "hbm", "kernel": None, "achieved": step_obj["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": step_obj["frac"], "traffic": None, "step": step_obj}
    if S.copy_rate is not None:
        # the same fraction against what THIS box's HBM sustains on a plain copy (SURVEY.md section 8d asks for both)
        out["roofline"].update(S.copy_rate)
        out["roofline"]["frac_of_achievable"] = round(out["roofline"]["achieved"] / S.copy_rate["achievable_gbs"], 4)
        out["roofline"]["step"]["frac_of_achievable"] = round(step_obj["achieved"] / S.copy_rate["achievable_gbs"], 4)

    return out


def add_cpu_baseline(S, out) -> None:
    """cpu_baseline (the C oracle on the host cores, rank 0 at N = 1) and the oracle as the CHECKER of what was just timed."""
    if S.rank == 0 and S.real_live == 1 and not S.args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(S.args.cpu_seconds, S.args.codec)
        except Exception as e:  # pragma: no cover
            out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        if S.real_live == 1:
            # the oracle as the checker of what was just timed: replay every step this process ran (warm-up + timed + long +
            # the other replay) for two tensors on the host and compare the error-feedback states bit for bit - fails loudly
            from oracle import c_oracle as CO
            import numpy as np
            x0_host = S.warm_state(S.rank)[1]
            checked = []
            for l, kv in ((0, 0), (S.L - 1, 1)):
                state = x0_host[l, kv].cpu().numpy().view(np.uint16).copy()
                pk = np.zeros(S.pkt_bytes // 2, dtype=np.uint16)
                ins = [S.xs[s][l, kv].cpu().numpy() for s in range(2)]
                for t in range(S.steps_run):
                    CO.compress(S.args.codec, ins[t & 1], state, S.N, S.C, packet=pk, new_base=state)
                for name, got in (("sender state", S.own_base[l, kv]), ("looped-back peer state", S.peer_base[l, W_LOGICAL - 2, kv])):
                    if not np.array_equal(got.cpu().numpy().view(np.uint16), state):
                        raise RuntimeError(f"parity spot check failed: layer {l} {'KV'[kv]} {name} differs from the C oracle after {S.steps_run} steps")
                checked.append(f"layer {l} {'KV'[kv]}")
            out["cpu_baseline"]["parity_spot_check"] = (f"error-feedback states of {', '.join(checked)} (sender and a looped-back peer) after all "
                                                       f"{S.steps_run} steps of this run == C oracle replay, bit for bit")


def teardown(S) -> None:
    """Communicators, plans, streams, IPC mappings - torn down before the line is written (RCCL prints through its own stdio buffer)."""
    # tear the communicators down first and flush C stdio (RCCL prints a version banner through its own stdio buffer),
    # so that the JSON line is the LAST thing on stdout
    S.torch.cuda.synchronize(S.dev)
    if S.p2p_ptr is not None and S.p2p_ptr.value:
        S.sync_all()                                   # nobody unmaps or frees while a peer may still read
        for q_, pq_ in S.p2p_peer.items():
            S.lib.cfx_ipc_close(S.ctx, ctypes.c_void_p(pq_))
        S.sync_all()
        S.lib.cfx_ipc_free(S.ctx, S.p2p_ptr)
    for plset in (S.step_plans, S.plans_inorder, S.plans_pipe, S.plans_gated):
        for pl_ in (plset or []):
            S.lib.cfx_plan_destroy(pl_)
    if S.xside:
        S.lib.cfx_stream_destroy(S.ctx, ctypes.c_void_p(S.xside))
    if S.native_comm is not None:
        try:
            S.torch.cuda.synchronize(S.dev)
            S.native_comm.close()
        except Exception:
            pass
    if S.world > 1:
        S.dist.destroy_process_group()

