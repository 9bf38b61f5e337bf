#!/usr/bin/env python3
"""bench.py - residual-compressed activation exchange, FLUX.1-dev 1024^2 ring-8 per-rank workload.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per
GPU with torch.distributed.run.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2], SURVEY.md §8d "Config 3"): one denoise step of FLUX.1-dev 1024^2 under
ring-attention sequence parallelism of logical degree 8 with the 1-bit residual codec, as seen by ONE rank:
57 attention layers x {K, V}, shard (N, C) = (544, 3072) fp16.  The step is replayed LAYER BY LAYER IN ORDER, the way a
model runs it (layer l+1's K,V only exist after layer l's attention, reference xfuser/compact/ring.py:188-206): layer l+1's
compress cannot start before layer l's reconstruction has finished (a kernel boundary), nothing is reordered across layers.
Per layer:
  A. compress K,V (k_absmean_compress: statistics + sign bits + in-launch finalize of the scales),
  X. exchange the packets (N live ranks exchange for real; the 8-N missing logical peers are looped back from the rank's own packets, so the
     per-GPU codec work is IDENTICAL for every N = weak scaling),
  B. reconstruct the 7 peers' K,V (14 tensors) onto their state arenas, and the rank's own error-feedback update.
Default (--own-ef xgate --p2p auto) = WHAT THE PLUGIN API RUNS (compactfusion_amd/compact/xlayer.py: compact_fwd's gather schedule and
compact_all_gather_kv issue this op, one native call per layer; `plugin_path` times the same step through that API): A and B are ONE launch on
the run stream (the exchange-layer op, cfx_plan_add_exchange_layer_p2p): B's workgroups are launched with A's, pull their state tiles into
registers while the scale reduction runs, and wait for a gate word.  X is NOT a collective: every rank's packets stay in UNCACHED IPC device
memory of its own GPU (cfx_ipc_alloc), the peers' reconstruction workgroups read them in place over xGMI, and workgroup 0 of the SAME launch
waits for A's packets, publishes a word the live peers have mapped, waits for theirs and opens the gate: one launch per layer on one stream,
nothing else.  At N = 1 there is no live peer: the same op, launch and kernel minus the remote reads and the waiting - `value` at N = 1 prices the launch structure every N
executes, not a wire.  At N > 1 the run is VALIDATED after the warm-up steps and again after the timed region (gate time-outs; every rank's
reconstruction of a shard against its owner's state); on any failure every rank falls back IN-PROCESS - p2p -> compress ; ncclAllGather ;
reconstruct in stream order -> torch.distributed per layer - re-runs warm-up and timed region, and `schedule_fallback` says which check tripped.
`collective_in_the_path` (secondary at N = 1; --p2p off): the same launch with flag-wait kernel ; ncclAllGather (libcfx's own RCCL
communicator, in place) ; flag-set kernel on the exchange stream.
`two_launches_per_layer` (secondary at N = 1; --own-ef ride): A ; X ; B as two codec launches in stream order, the previous layer's
own error-feedback update riding in A (nothing reads that state before the next denoise step, ring.py:207-209) - the fall-back schedule.
`with_cu_partition` (secondary at N = 1): the default with the run stream masked to CUs [0, 224) and the exchange stream to [224, 256):
any partial CU mask costs the layer launch ~5 us, so the streams are not partitioned.
`loopback_one_launch_per_layer` (secondary, N = 1): the layer as ONE launch (cfx_compress_batch_gated: reconstruction behind an
in-launch arrival gate) - only possible when the packets a reconstruction needs are produced by the same launch, i.e. with
looped-back peers and NO exchange in between; never `value`.
`plugin_path` (N = 1): SURVEY 8d protocol 1 THROUGH THE PLUGIN API - the same 57-layer step issued by compact_all_gather_kv (what patch_gather_fwd
calls) and by compact_fwd with a no-op attention, one native op per layer (tools/plugin_path_bench.py as a child process).
`configs` (N = 1): every BASELINE.json configuration's step (tools/config_table.py): ms per step, algorithmic bytes, fraction of the HBM roofline.
`low_rank_presets` (N = 1): compress time per K,V pair of the reference's LOW_RANK / LOW_RANK_Q presets on the same shard (one persistent
launch each, csrc/cfx_lrslab.hip).
`overlap_with_attention` (N = 1): SURVEY 8d protocol 2 - compact_fwd on the exchange lane beside real SDPA attention: what the exchange adds
to a model step (tools/overlap_bench.py as a child process); `exposed_exchange_ms_per_step` carries its figure at the top level.
Inputs are synthetic and already resident in HBM; the state arenas (3.0 GB) + inputs (0.76 GB) dwarf the 256 MB
Infinity Cache, so every step streams from HBM (cold numbers).

value = whole-job fp16 activation bytes compressed + reconstructed per second (GB/s), i.e.
        n_gpus * 57 * (2 + 14) * 544*3072*2 B / step time, from the in-order replay.
`pure_exchange_upper_bound` = the same step through cfx_plan_run_pipelined, which DOES reorder across layers (statistics of
        layers j+7.. beside the reconstruction of layers j..): only legal because the synthetic inputs of all layers are resident;
        a model cannot run it.  Reported as a secondary figure, never as `value`.
roofline = the dominant kernel of the in-order step - the layer's only codec launch, k_absmean_compress<bits,gated>: (2 x 6.125 + 14 x 4.125)
        B/element x 544 x 3072 / its average duration from hipEvents on the run stream around every 4th step of the timed
        region / launches per step; with two launches per layer it is launch B, k_binary_dequant; `roofline.step` prices the WHOLE step.
cpu_baseline = the C oracle (oracle/cfx_oracle.c, OpenMP) timed on ALL host hardware threads of this box on one layer of the same
        workload - `value` = the fastest thread count of a short sweep, `all_threads` beside it; reported baseline only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

# HIP multiplexes streams over a pool of hardware queues; with the variable unset, a stream created after RCCL has initialised can end up
# time-sliced against the CU-masked exchange stream's queue (measured: the flag-ordered layer launch 24.6 -> 50 us).  Any explicit value
# restores one queue per stream; must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

from benchlib import report, runner, schedules          # noqa: E402
from benchlib.workload import HBM_PEAK_GBS, config_key, group_recv_offset, measure_copy_rate, parse, setup          # noqa: E402,F401


def main():
    # One process per GPU.  The four parts (tools/benchlib/):
    #   workload.py   (a) ranks, the synthetic workload resident in HBM, packet buffers, the context's switches          setup()
    #   schedules.py  (b) the plans: N = 1 forms, the step with the collective in it, the peer-to-peer exchange layer     decide / build_* / setup_exchange
    #   safety.py     (c) states_consistent / validate / the fall-back ladder - the N > 1 safety net                     (applied by runner.py)
    #   report.py     (d) the JSON line: value, schedules, xgmi, roofline (+ committed profiles), cpu_baseline + the oracle spot check
    # then tools/bench_secondary.py: protocol 2 beside attention, the plugin path, every BASELINE config, the presets
    args = parse()
    if args.print_config_key:
        print(json.dumps(config_key(args, args.gpus)))
        return
    # stdout carries ONE line, the JSON.  Whatever a library prints on the way (RCCL announces its version on stdout when a communicator
    # comes up) goes to stderr: file descriptor 1 points there until the line is written through the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    S = setup(args)                               # (a)
    schedules.decide(S)                           # (b)
    schedules.build_local_plans(S)
    schedules.setup_streams(S)
    schedules.setup_exchange(S)
    torch = S.torch
    S.copy_rate = None
    if S.real_live == 1 and not args.emulate_live and not args.no_copy_rate:
        S.copy_rate = measure_copy_rate(S.lib, S.ctx, S.dev, S.sh)
    if args.copy_probe:
        nb = 96 * 1024 * 1024
        src = [torch.empty(nb, dtype=torch.uint8, device=S.dev).random_(0, 255) for _ in range(4)]
        dst = [torch.empty(nb, dtype=torch.uint8, device=S.dev) for _ in range(4)]
        for i in range(args.copy_probe):
            S.check(S.lib.cfx_copy_probe(S.ctx, dst[i % 4].data_ptr(), src[i % 4].data_ptr(), nb, S.sh), "copy_probe")
        torch.cuda.synchronize(S.dev)
        del src, dst
    runner.timed_region(S)                        # warm-up ; validate ; K timed steps ; validate - falling back in-process (c)
    # Everything below this line is commentary on a measurement that has been taken AND validated.  With more than one rank process it
    # must not cost the line: the long run, the uncompressed exchange (ring hops no RCCL has seen from this code) and the other pattern run
    # here for the first time on real links, and an error in them - the same on every rank - is reported in the line
    # (`secondary_legs_error`) instead of ending the run.  One process: as loud as ever.
    S.long_ms = S.other_ms = S.two_ms = S.loop_ms = S.relay_ms = S.part_ms = S.coll_ms = None
    S.raw_legs, S.other_pattern_ms, S.raw_ms, S.secondary_error = {}, None, None, None
    try:
        runner.secondary_legs(S)
        torch.cuda.synchronize(S.dev)
        ge = S.lib.cfx_gate_errors(S.ctx)
        if ge != 0:
            raise RuntimeError(f"cfx_gate_errors = {ge}: a gated launch gave up waiting for its packets")
        ok, why = runner.consistent(S)            # state sanity (bit-exact error-feedback consistency)
        if not ok:
            raise RuntimeError(why)
        runner.raw_exchange_legs(S)
    except Exception as e:  # noqa: BLE001
        if S.world == 1:
            raise SystemExit(f"[bench] {type(e).__name__}: {e}")
        S.secondary_error = f"{type(e).__name__}: {e}"
        print(f"[bench] rank {S.rank}: a leg after the timed region failed ({S.secondary_error}); the line carries the timed region", file=sys.stderr)
    out = report.build_line(S)                    # (d)
    if S.secondary_error is not None:
        out["secondary_legs_error"] = S.secondary_error
    report.add_cpu_baseline(S, out)

    def write_line():
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        if S.rank == 0:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
    if S.world > 1:
        # the line first: descriptor 1 points at stderr, so whatever the communicators print while they are torn down cannot get in front
        # of it - and a tear-down that hangs on hardware this code has never seen (IPC unmapping, ncclCommDestroy) must not cost the line
        write_line()
        report.teardown(S)
        return
    report.teardown(S)
    if S.rank == 0 and S.world == 1 and not args.emulate_live and not args.no_secondary:
        # the secondary legs that need nothing of this process's state (tools/bench_secondary.py): protocol 2 beside real attention and the
        # plugin path (child processes), every BASELINE configuration through the native replay AND through the plugin API, the presets
        del S.xs, S.own_base, S.peer_base
        torch.cuda.empty_cache()
        torch.cuda.synchronize(S.dev)
        import bench_secondary as BS
        if args.overlap_steps > 0:
            BS.overlap_leg(out, args.overlap_steps, S.L)
        if args.plugin_steps > 0:
            BS.plugin_leg(out, args.plugin_steps, S.L)
        if not args.no_config_table:
            BS.configs_leg(out, HBM_PEAK_GBS)
            if args.plugin_steps > 0:
                BS.plugin_configs_leg(out)
        if args.overlap_steps > 0:
            BS.overlap_presets_leg(out, max(4, args.overlap_steps // 2), S.L)
        BS.lowrank_leg(out, S.dev, S.N, S.C)
    write_line()


if __name__ == "__main__":
    main()
