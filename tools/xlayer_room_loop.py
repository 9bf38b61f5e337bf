#!/usr/bin/env python3
"""Developer probe: the exchange-layer launch beside a collective KERNEL with the register footprint of RCCL's (tests/fake_rccl,
CFX_FAKE_RCCL_FAT=1; 256 threads x 280 VGPRs, 105 workgroups per collective) on the exchange stream, unpartitioned, 150 ms gate timeout:
repetitions of 4 steps x 6 layers at the FLUX shard, each checked against compress ; all-gather ; reconstruct.  SHARE=1 (default): one run
stream and one exchange stream for every repetition; SHARE=0: a fresh pair per repetition (hardware-queue churn)."""
import ctypes, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.environ["GPU_MAX_HW_QUEUES"] = "8"
import xlayer_cases as T
from compactfusion_amd import _lib, codecs as K
lib = _lib.load()
os.environ["CFX_FAKE_RCCL_MODE"] = "loopback"; os.environ.setdefault("CFX_FAKE_RCCL_FAT", "1")
warm = os.environ.get("WARM", "1") == "1"
ctx = lib.cfx_create(0)
assert lib.cfx_prepare(ctx) == 0 and lib.cfx_set_gate_timeout_ms(ctx, int(os.environ.get("TMO", 150))) == 0
assert lib.cfx_rccl_load(T._fake_path().encode()) == 0
uid = ctypes.create_string_buffer(128); assert lib.cfx_comm_unique_id(ctx, uid) == 0
comm = lib.cfx_comm_create(ctx, uid, 4, 0)
W = T.Layers(6, 544, 3072, 7, seed=9, live=4)
ref = T._reference(lib, _lib, ctx, W, 4, comm=comm)
if warm:
    s = torch.cuda.Stream()
    assert lib.cfx_comm_all_gather(comm, W.buf[0, 0].data_ptr(), W.buf[0].data_ptr(), 2 * W.slot, s.cuda_stream) == 0
    torch.cuda.synchronize()
bad = 0
share = os.environ.get("SHARE", "1") == "1"
run0 = torch.cuda.Stream()
side0 = T._masked(lib, ctx, 0, 256) if share else None
for rep in range(int(os.environ.get("REPS", 40))):
    W.reset()
    run = run0 if share else torch.cuda.Stream()
    plans = T._plans(lib, _lib, ctx, W, "xlayer", comm=comm, side=side0)
    for i in range(4):
        rc = lib.cfx_plan_run(plans[i & 1], 0, lib.cfx_plan_size(plans[i & 1]), run.cuda_stream)
        if rc != 0: break
    torch.cuda.synchronize()
    e = lib.cfx_gate_errors(ctx)
    ok = rc == 0 and e == 0 and torch.equal(W.own, ref[0]) and torch.equal(W.peer, ref[1])
    bad += (not ok)
    if not ok: print("rep", rep, "rc", rc, "gate errors", e, flush=True)
    for p in plans: lib.cfx_plan_destroy(p)
print("warm", warm, "share", share, "bad reps", bad, flush=True)
