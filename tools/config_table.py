#!/usr/bin/env python3
"""Per-BASELINE-configuration exchange-step timing on ONE MI355X (developer tool; output committed under profiles/).

For each configuration of BASELINE.json (shapes from SURVEY.md section 8d) one rank's codec work of one denoise step is
replayed layer by layer, in order, with the other ranks looped back (their packets = our packets, their states distinct
buffers), through the C-ABI:
  ring configs (3, 4):   per layer compress(K,V) with error-feedback update, then reconstruct (W-1) peers x {K,V}
  gather configs (2, 5): per layer compress(K,V) without update, then reconstruct all W ranks x {K,V} (compact_all_gather)
  config 1:              one tensor, int8 residual round trip (the reference's own CPU-runnable case)
Reported: ms per step (wall clock over >= 20 steps, states rotating over all layers so nothing is cache-resident),
GB/s of fp16 activations compressed + reconstructed, and the C oracle on the host cores for the same step (bounded
sample; n/a for the low-rank codec, which has no C restatement).
"""
import json
import os
import sys
import ctypes
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from compactfusion_amd import _lib, codecs as K

CONFIGS = [
    # name, codec id, param, (N, C), layers, tensors compressed per layer, tensors reconstructed per layer, update on compress
    ("1 int8 residual [1,4096,1152], world 1", 4, 0, (4096, 1152), 1, 1, 1, True),
    ("2 PixArt-a 512^2 SP2 patch-gather INT4", 3, 0, (1024, 1152), 28, 2, 4, False),
    ("3 FLUX.1 1024^2 ring 8, 1-bit (bench.py workload, in order)", 1, 0, (544, 3072), 57, 2, 14, True),
    ("4 CogVideoX-5B SP4 ring INT4", 3, 0, (4448, 3072), 42, 2, 6, True),
    ("5 SD3 1024^2 SP8 patch-gather top-k 1:8", 5, 8, (512, 1536), 24, 2, 16, False),
    ("5 SD3 1024^2 SP8 patch-gather LOW_RANK r=8", 101, 8, (512, 1536), 24, 2, 16, False),
    ("5 SD3 1024^2 SP8 patch-gather LOW_RANK r=16", 101, 16, (512, 1536), 24, 2, 16, False),
]
NAMES = {1: "binary", 3: "int4", 4: "int8", 5: "topk"}
# algorithmic bytes per element (compress + error feedback, reconstruct); low-rank: x + state in, state out (6), state in / out (4)
ALG = {1: (6.125, 4.125), 2: (6.25, 4.25), 3: (6.5, 4.5), 4: (7.0, 5.0), 5: (6 + 2.5 / 8, 4 + 2.5 / 8), 101: (6.0, 4.0)}


def alg_bytes(cid, N, C, L, ncomp, nrec, update):
    """Algorithmic HBM bytes of one step: `ncomp` tensors compressed (+ error feedback when the compress updates the state; a compress
    that does not - gather mode - reads x and the state and writes the packet only: 2 B/el less) and `nrec` reconstructed, per layer."""
    c, d = ALG[cid]
    return int(L * N * C * (ncomp * (c if update else c - 2.0) + nrec * d))


_STREAMS = []
_IPC = []


def _streams(ctx, lib):
    """(run stream, exchange stream handle): the exchange-layer ops order two streams by flags - a run stream that is not the legacy NULL
    stream and ONE CU-masked (full mask) exchange stream for every plan."""
    if not _STREAMS:
        hx = ctypes.c_void_p()
        assert lib.cfx_stream_create_masked(ctx, 0, torch.cuda.get_device_properties(0).multi_processor_count, ctypes.byref(hx)) == 0
        _STREAMS.extend([torch.cuda.Stream(), hx.value])
    return _STREAMS


def gpu_step(cid, param, N, C, L, ncomp, nrec, update, min_steps=20, budget_s=0.2):
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    ctx = K.context(0)
    g = torch.Generator(device=dev).manual_seed(1)
    Lb = max(L, min(64, int(2.0e9 // ((ncomp + nrec) * N * C * 2)) or 1))      # enough distinct state to defeat the 256 MB cache
    # synthetic data as bench.py's: states that track their activations, the step-to-step drift 0.1 of the activations' scale (independent
    # random states would be residuals of |d| ~ 1.1 - at that size the 1-bit / 2-bit codecs' row partials (512 channels) leave their 32-bit
    # words and every U job takes a second round trip for the 64-bit ones: 1.61 instead of 1.45 ms per step at config 3)
    own = torch.randn(Lb, ncomp, N, C, generator=g, device=dev).half()
    x = [(own.float() + 0.1 * torch.randn(Lb, ncomp, N, C, generator=g, device=dev)).half() for _ in range(2)]
    peers = torch.randn(Lb, nrec, N, C, generator=g, device=dev).half()
    lowrank = cid >= 100
    run_stream = _streams(ctx, lib)[0]
    sh = run_stream.cuda_stream
    if not lowrank:
        slot = (K.packet_bytes(cid, N, C, param) + 255) // 256 * 256
        wsb = lib.cfx_workspace_bytes(cid, N, C, param, ncomp)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        plans = []
        # ONE exchange-layer op per layer, the product's default form (cfx_plan_add_exchange_layer_p2p; what compact/xlayer.py issues): the
        # packets in the uncached IPC arena, compress + the exchange (no live peer here: one published word) + the reconstruction of every
        # tensor whose packet it feeds in one launch for the codecs that have the layer form, in stream order for the others
        one_op = nrec <= 16
        flags_off = Lb * ncomp * slot
        ipc, handle = ctypes.c_void_p(), (ctypes.c_ubyte * 64)()
        assert lib.cfx_ipc_alloc(ctx, flags_off + 2 * Lb * 64, ctypes.byref(ipc), handle) == 0, lib.cfx_last_error_string(ctx)
        _IPC.append(ipc)

        def pkp(l, i):
            return ipc.value + (l * ncomp + i) * slot
        for s in range(2):
            plan = lib.cfx_plan_create(ctx)
            for l in range(Lb):
                c = (_lib.CompItem * ncomp)(*[_lib.CompItem(x[s][l, i].data_ptr(), own[l, i].data_ptr(), own[l, i].data_ptr() if update else None,
                                                            pkp(l, i)) for i in range(ncomp)])
                if one_op:
                    d = (_lib.DecompItem * nrec)(*[_lib.DecompItem(pkp(l, j % ncomp), peers[l, j].data_ptr(), peers[l, j].data_ptr())
                                                   for j in range(nrec)])
                    rc = lib.cfx_plan_add_exchange_layer_p2p(plan, cid, N, C, param, 1 if update else 0, ncomp, c, nrec, d,
                                                             ipc.value + flags_off + (s * Lb + l) * 64, 0, (ctypes.c_void_p * 1)(), ws.data_ptr(), wsb)
                    assert rc >= 0, (rc, lib.cfx_last_error_string(ctx))
                    continue
                assert lib.cfx_plan_add_compress(plan, cid, N, C, param, 1 if update else 0, ncomp, c, ws.data_ptr(), wsb) >= 0
                for a in range(0, nrec, 16):
                    n = min(16, nrec - a)
                    d = (_lib.DecompItem * n)(*[_lib.DecompItem(pkp(l, (a + j) % ncomp), peers[l, a + j].data_ptr(), peers[l, a + j].data_ptr())
                                                for j in range(n)])
                    assert lib.cfx_plan_add_decompress(plan, cid, N, C, param, n, d) >= 0
            plans.append(plan)
        ops_per_layer = 1 if one_op else 1 + (nrec + 15) // 16

        def step(i):
            first = (i * L) % Lb
            n = min(L, Lb - first)
            assert lib.cfx_plan_run(plans[i & 1], first * ops_per_layer, n * ops_per_layer, sh) == 0
            if n < L:
                assert lib.cfx_plan_run(plans[i & 1], 0, (L - n) * ops_per_layer, sh) == 0
    else:
        # the low-rank layer as plan ops too (cfx_plan_add_lr_compress / _decompress): replayed natively like the other rows
        q = False
        pkh = K.lr_packet_halves(q, N, C, param)
        pk = torch.zeros(Lb, ncomp, (pkh + 127) // 128 * 128, dtype=torch.float16, device=dev)
        rp = K.lr_rank_pad(param)
        q0 = [torch.randn(C, rp, generator=g, device=dev) for _ in range(ncomp)]
        wsb = lib.cfx_lr_workspace_bytes(int(q), N, C, param, max(ncomp, 16))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        qp = (ctypes.c_void_p * ncomp)(*[t.data_ptr() for t in q0])
        plans = []
        for s in range(2):
            plan = lib.cfx_plan_create(ctx)
            for l in range(Lb):
                c = (_lib.CompItem * ncomp)(*[_lib.CompItem(x[s][l, i].data_ptr(), own[l, i].data_ptr(), own[l, i].data_ptr() if update else None,
                                                            pk[l, i].data_ptr()) for i in range(ncomp)])
                assert lib.cfx_plan_add_lr_compress(plan, int(q), N, C, param, 1 if update else 0, ncomp, c, qp, ws.data_ptr(), wsb) >= 0
                for a in range(0, nrec, 16):
                    n = min(16, nrec - a)
                    d = (_lib.DecompItem * n)(*[_lib.DecompItem(pk[l, (a + j) % ncomp].data_ptr(), peers[l, a + j].data_ptr(), peers[l, a + j].data_ptr())
                                                for j in range(n)])
                    assert lib.cfx_plan_add_lr_decompress(plan, int(q), N, C, param, n, d, ws.data_ptr(), wsb) >= 0
            plans.append(plan)
        ops_per_layer = 1 + (nrec + 15) // 16

        def step(i):
            first = (i * L) % Lb
            n = min(L, Lb - first)
            assert lib.cfx_plan_run(plans[i & 1], first * ops_per_layer, n * ops_per_layer, sh) == 0
            if n < L:
                assert lib.cfx_plan_run(plans[i & 1], 0, (L - n) * ops_per_layer, sh) == 0
    torch.cuda.synchronize()          # (the inputs were made on the default stream, the steps run on the run stream)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    assert lib.cfx_gate_errors(ctx) == 0
    steps = max(min_steps, int(budget_s / max(1e-6, L * 40e-6)))
    t0 = time.perf_counter()
    for i in range(steps):
        step(3 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    for plan in plans:
        lib.cfx_plan_destroy(plan)
    while _IPC:
        lib.cfx_ipc_free(ctx, _IPC.pop())
    return ms


def cpu_step(cid, param, N, C, L, ncomp, nrec, update, budget=8.0):
    if cid >= 100:
        return None, 0
    from oracle import c_oracle as CO
    name = NAMES[cid]
    rng = np.random.default_rng(0)
    base = rng.standard_normal((N, C)).astype(np.float16)
    x = (base.astype(np.float32) + 0.1 * rng.standard_normal((N, C)).astype(np.float32)).astype(np.float16)
    own = [base.copy().view(np.uint16) for _ in range(ncomp)]
    peers = [base.copy().view(np.uint16) for _ in range(nrec)]
    pk = [np.zeros(CO.load().oracle_packet_bytes(cid, N, C, param) // 2, dtype=np.uint16) for _ in range(ncomp)]

    def layer():
        for i in range(ncomp):
            CO.compress(name, x, own[i], N, C, param, update=update, packet=pk[i], new_base=own[i] if update else None)
        for j in range(nrec):
            CO.decompress(name, pk[j % ncomp], peers[j], N, C, param, out=peers[j])
    layer()
    most = int(CO.num_threads())
    best, best_t = None, most
    for t in sorted({most} | {c for c in (8, 16, 32, 64, 128) if c <= most}):
        CO.set_num_threads(t)
        layer()
        t0 = time.perf_counter()
        layer()
        d = time.perf_counter() - t0
        if best is None or d < best:
            best, best_t = d, t
    CO.set_num_threads(best_t)
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < budget and n < 50 * L):
        layer()
        n += 1
    return (time.perf_counter() - t0) / n * L * 1e3, best_t


def main():
    rows = []
    only = os.environ.get("ONLY")              # e.g. ONLY=2: one configuration (for kernel traces), GPU side only
    for name, cid, param, (N, C), L, ncomp, nrec, update in CONFIGS:
        if only and not name.startswith(only):
            continue
        if only:
            print(name, gpu_step(cid, param, N, C, L, ncomp, nrec, update), "ms/step")
            continue
        ms = gpu_step(cid, param, N, C, L, ncomp, nrec, update)
        act = L * (ncomp + nrec) * N * C * 2
        cms, thr = cpu_step(cid, param, N, C, L, ncomp, nrec, update)
        rows.append({"config": name, "shape": [N, C], "layers": L, "compressed_per_layer": ncomp, "reconstructed_per_layer": nrec,
                     "gpu_ms_per_step": round(ms, 4), "gpu_GBps": round(act / ms / 1e6, 1),
                     "cpu_ms_per_step": None if cms is None else round(cms, 2), "cpu_GBps": None if cms is None else round(act / cms / 1e6, 2),
                     "cpu_threads": thr})
        print(rows[-1], flush=True)
        torch.cuda.empty_cache()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "config_table.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rows, open(out, "w"), indent=1)
    print("| BASELINE config | shard (N,C) | layers | GPU ms/step (in order) | GPU GB/s fp16 | C oracle ms/step (threads) | CPU GB/s |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        cpu = "n/a" if r["cpu_ms_per_step"] is None else f"{r['cpu_ms_per_step']} ({r['cpu_threads']})"
        print(f"| {r['config']} | ({r['shape'][0]},{r['shape'][1]}) | {r['layers']} | {r['gpu_ms_per_step']} | {r['gpu_GBps']} | {cpu} | "
              f"{'n/a' if r['cpu_GBps'] is None else r['cpu_GBps']} |")


if __name__ == "__main__":
    main()
