#!/usr/bin/env python3
"""Developer probe: the exchange-layer op (reconstruction launched with the compress group, gated on the exchange stream's flag) against
the in-order two-launch layer on the FLUX shard, looped-back peers, no collective library: per-step time and bit-equality of the states."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compactfusion_amd import _lib, codecs as K

lib = _lib.load()
ctx = K.context(0)
L, N, C, P = int(os.environ.get("L", 57)), 544, 3072, 7
CODEC = int(os.environ.get("CODEC", 1))        # 1 = 1-bit, 2 = 2-bit (its exchange-layer form needs a -DCFX_EXP_INT2_XLAYER build)
dev = "cuda"
if os.environ.get("MAIN") == "hexmask":
    # arbitrary CU mask (8 hex words, bit i = CU i/8 of XCD i%8), straight from the HIP runtime
    hip = ctypes.CDLL("libamdhip64.so")
    words = [int(w, 16) for w in os.environ["MASK"].split(",")]
    arr = (ctypes.c_uint32 * len(words))(*words)
    h = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), len(words), arr) == 0
    main = torch.cuda.ExternalStream(h.value)
elif os.environ.get("MAIN") == "masked":
    h = ctypes.c_void_p()
    assert lib.cfx_stream_create_masked(ctx, 0, int(os.environ.get("MAIN_CUS", 256)), ctypes.byref(h)) == 0
    main = torch.cuda.ExternalStream(h.value)
else:
    main = torch.cuda.Stream()
torch.cuda.set_stream(main)
g = torch.Generator(device=dev).manual_seed(1)
x0 = torch.randn(L, 2, N, C, generator=g, device=dev).half()
xs = [(x0.float() + 0.1 * torch.randn(L, 2, N, C, generator=g, device=dev)).half() for _ in range(2)]
slot = (K.packet_bytes(CODEC, N, C) + 255) // 256 * 256
wsb = lib.cfx_workspace_bytes(CODEC, N, C, 0, 2)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

comm = None
if os.environ.get("COMM"):
    from compactfusion_amd.exchange import NativeComm
    ncomm = NativeComm(0, solo_ranks=1)
    comm = ncomm.handle if os.environ["COMM"] != "init" else None

ipc, _h = ctypes.c_void_p(), ctypes.create_string_buffer(64)
assert lib.cfx_ipc_alloc(ctx, L * 2 * slot + 2 * L * 64, ctypes.byref(ipc), _h) == 0


def state():
    return x0.clone(), x0.unsqueeze(1).repeat(1, P, 1, 1, 1).contiguous()

extra = []
for _ in range(int(os.environ.get("EXTRA_MASKED", 0))):        # idle CU-masked streams: each is a hardware queue of its own
    hh = ctypes.c_void_p()
    assert lib.cfx_stream_create_masked(ctx, 0, 256, ctypes.byref(hh)) == 0
    extra.append(hh.value)
    if os.environ.get("EXTRA_USED"):                              # (touch it once: a queue only exists once something was launched on it)
        torch.zeros(1, device=dev)
        lib.cfx_flag_set(ctx, torch.zeros(16, dtype=torch.int32, device=dev).data_ptr(), 1, hh.value)
torch.cuda.synchronize()
shared_side = None
if os.environ.get("SIDE") == "shared":
    hs = ctypes.c_void_p()
    assert lib.cfx_stream_create_masked(ctx, 0, 256, ctypes.byref(hs)) == 0
    shared_side = hs.value

def build(kind, own, peer, send):
    plans = []
    for s in range(2):
        plan = lib.cfx_plan_create(ctx)
        if shared_side and kind == "xlayer":
            assert lib.cfx_plan_use_exchange_stream(plan, shared_side) == 0
        for l in range(L):
            c = (_lib.CompItem * 2)(*[_lib.CompItem(xs[s][l, b].data_ptr(), own[l, b].data_ptr(), own[l, b].data_ptr(), send[l, b].data_ptr()) for b in range(2)])
            items = [_lib.DecompItem(send[l, b].data_ptr(), peer[l, p, b].data_ptr(), peer[l, p, b].data_ptr()) for p in range(P) for b in range(2)]
            d = (_lib.DecompItem * len(items))(*items)
            if kind == "p2p":
                # the peer-to-peer exchange layer with no live peer: packets and flag word in cfx_ipc_alloc memory, the exchange inside the launch
                o = (l * 2) * slot
                cp = (_lib.CompItem * 2)(*[_lib.CompItem(xs[s][l, b].data_ptr(), own[l, b].data_ptr(), own[l, b].data_ptr(), ipc.value + o + b * slot) for b in range(2)])
                it2 = [_lib.DecompItem(ipc.value + o + b * slot, peer[l, p, b].data_ptr(), peer[l, p, b].data_ptr()) for p in range(P) for b in range(2)]
                rc = lib.cfx_plan_add_exchange_layer_p2p(plan, CODEC, N, C, 0, _lib.FLAG_UPDATE_CACHE, 2, cp, len(it2), (_lib.DecompItem * len(it2))(*it2),
                                                         ipc.value + L * 2 * slot + (s * L + l) * 64, 0, (ctypes.c_void_p * 1)(), ws.data_ptr(), wsb)
            elif kind == "xlayer":
                rc = lib.cfx_plan_add_exchange_layer(plan, CODEC, N, C, 0, _lib.FLAG_UPDATE_CACHE, 2, c, len(items), d, comm, send[l].data_ptr() if comm else None, send[l].data_ptr() if comm else None, 2 * slot, ws.data_ptr(), wsb)
            elif kind in ("gated", "gated+poller"):
                rc = lib.cfx_plan_add_compress_gated(plan, CODEC, N, C, 0, _lib.FLAG_UPDATE_CACHE, 2, c, 0, None, len(items), d, ws.data_ptr(), wsb)
            else:
                rc = lib.cfx_plan_add_compress(plan, CODEC, N, C, 0, _lib.FLAG_UPDATE_CACHE, 2, c, ws.data_ptr(), wsb)
                assert rc >= 0
                rc = lib.cfx_plan_add_decompress(plan, CODEC, N, C, 0, len(items), d)
            assert rc >= 0, (rc, lib.cfx_last_error_string(ctx))
        assert lib.cfx_plan_finalize(plan) == 0
        plans.append(plan)
    return plans

def run(kind, steps=20):
    own, peer = state()
    send = torch.zeros(L, 2, slot, dtype=torch.uint8, device=dev)
    plans = build(kind, own, peer, send)
    def step(i):
        rc = lib.cfx_plan_run(plans[i & 1], 0, lib.cfx_plan_size(plans[i & 1]), main.cuda_stream)
        assert rc == 0, (rc, lib.cfx_last_error_string(ctx))
    for i in range(4): step(i)
    poll = None
    if kind == "gated+poller":
        # the loop-back gated launches while a one-wave polling kernel sits on ANOTHER (CU-masked) stream's queue for the whole timed region
        hp = ctypes.c_void_p()
        assert lib.cfx_stream_create_masked(ctx, 0, 256, ctypes.byref(hp)) == 0
        poll = (hp, torch.zeros(16, dtype=torch.int32, device=dev))
        torch.cuda.synchronize()
        assert lib.cfx_flag_wait(ctx, poll[1].data_ptr(), 1, hp.value) == 0
    torch.cuda.synchronize() if poll is None else main.synchronize(); t0 = time.perf_counter()
    for i in range(4, 4 + steps): step(i)
    main.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    if poll is not None:
        assert lib.cfx_flag_set(ctx, poll[1].data_ptr(), 1, main.cuda_stream) == 0
        torch.cuda.synchronize()
        lib.cfx_stream_destroy(ctx, poll[0])
    # host issue time alone
    t0 = time.perf_counter(); step(0); th = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    lib.cfx_profile_enable(ctx, 4096, 0xffffffff, 1)
    step(1); torch.cuda.synchronize()
    ids = (ctypes.c_int * 4096)(); ms = (ctypes.c_float * 4096)()
    k = lib.cfx_profile_read(ctx, ids, ms, 4096)
    lib.cfx_profile_enable(ctx, 0, 0, 1)
    agg = {}
    for i in range(k): agg.setdefault(lib.cfx_kernel_name(ids[i]).decode(), []).append(ms[i] * 1e3)
    print(f"{kind:8s} {dt:.3f} ms/step  host issue {th:.3f} ms  gate errors {lib.cfx_gate_errors(ctx)}  "
          f"{ {a: (round(sum(v) / len(v), 2), len(v)) for a, v in agg.items()} }", flush=True)
    for p in plans: lib.cfx_plan_destroy(p)
    return own, peer

ref = run("inorder")
for kind in os.environ.get("KINDS", "gated,xlayer").split(","):
    got = run(kind)
    print("   states equal to in-order:", torch.equal(got[0], ref[0]), torch.equal(got[1], ref[1]))
