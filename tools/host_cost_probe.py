#!/usr/bin/env python3
"""Developer probe: where the host time of one layer's exchange goes (us per layer): the native call alone (cfx_plan_run_x of the
exchange-layer op), LayerOp.run, compact_all_gather_kv (steady path), compact_fwd with a no-op attention.  GPU box: python tools/host_cost_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--steps", "1", "--quiet"]
import runpy
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "plugin_path_bench.py"))
torch, cm, ring, L = ns["torch"], ns["cm"], ns["ring"], ns["L"]
ks, vs, q0, T, CT = ns["ks"], ns["vs"], ns["q0"], ns["T"], ns["CT"]
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ns["init"]()
    for i in range(4):
        ns["gather"](i)
    torch.cuda.synchronize()
    exs = [cm._kv_exchanges[(f"{l}-k", f"{l}-v", None)] for l in range(L)]
    sh = torch.cuda.current_stream().cuda_stream

    def t(fn, reps=30):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
            best = min(best, dt)
        torch.cuda.synchronize()
        return best / L * 1e6
    k, v = ks[0], vs[0]

    def native():
        for l in range(L):
            op = exs[l].xop
            reg = op.region
            o = reg.n_exec & 1; reg.n_exec += 1
            op._xs[0], op._xs[1] = k[l].data_ptr(), v[l].data_ptr()
            op._run_x(op._plans[sh][0], o, 1, op._xs, 2, sh)

    def layerop():
        for l in range(L):
            exs[l].xop.run(k[l], v[l], sh)

    def steady():
        for l in range(L):
            exs[l].step_steady(k[l], v[l])

    tags = [(f"{l}-k", f"{l}-v") for l in range(L)]

    def api():
        for l in range(L):
            cm.compact_all_gather_kv(tags[l][0], tags[l][1], k[l], v[l], CT, group=None)

    def cur_stream():
        for l in range(L):
            torch.cuda.current_stream(k[l].device).cuda_stream
    print("us per layer (best of 30): native cfx_plan_run_x %.2f | LayerOp.run %.2f | step_steady %.2f | compact_all_gather_kv %.2f | (current_stream lookup alone %.2f)"
          % (t(native), t(layerop), t(steady), t(api), t(cur_stream)))
