"""How far apart are the REFERENCE's own two execution modes on the G12 LOW_RANK_Q-32 trace?  (BUILD container only: imports /root/reference.)

The reference decorates subspace_iter and the int4 quantiser with @torch.compile (compress_lowrank.py:14, compress_quantize.py:522-640):
on its own hardware it runs the COMPILED form, the goldens of this repo were captured in EAGER mode (TORCHDYNAMO_DISABLE=1), and the two
round differently (inductor keeps fp32 between fused ops, eager rounds to fp16 after every op; SURVEY.md section 0, "Numerics trap").
LOW_RANK_Q quantises the fp16 factors to int4 levels, so a last-bit difference in a factor moves whole quantisation levels.  This script
runs the same 28-step trace (make_golden_quality.py: same seeds, same start matrices) once per mode, each in its own process, and writes
    tests/golden/g12_lrq32_modes.npz          eager / compiled traces [step][rel_err_k, rel_err_v, attention PSNR dB]
    tests/golden/g12_lrq32_modes.json         the spread between them (max over steps of |rel err difference| / rel err, |PSNR difference|)
tests/test_gpu_quality.py holds the HIP path's LOW_RANK_Q trace to the eager golden within THAT spread - the reference's own
reproducibility band - instead of a number picked by hand.

usage: python tests/golden/measure_lrq_spread.py            (about two minutes: inductor compiles on the CPU)
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def one_mode():
    sys.path.insert(0, HERE)
    import types
    import torch
    import make_golden_quality as G
    m = types.ModuleType("xfuser")
    m.__path__ = [os.path.join(G.REF, "xfuser")]
    sys.modules["xfuser"] = m
    from xfuser.prof import Profiler
    Profiler.instance().disable()
    from xfuser.collector import collector
    collector.init(collector.Collector("/tmp/cfx_golden_collector", enabled=False))
    import xfuser.compact.main as cm
    from xfuser.compact.utils import CompactConfig, COMPACT_COMPRESS_TYPE as T
    Tn = 28
    ks, vs, q = G.drift(G.SEED_X, Tn), G.drift(G.SEED_X + 1, Tn), G.query(G.SEED_X + 2)
    tname, kw = G.PRESETS["lrq32"]
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: None, residual=1, ef=True, simulate=False, log_stats=False, **kw))
    rows = []
    for t in range(Tn):
        typ = T.WARMUP if t == 0 else T[tname]
        rec = []
        for kv, x in enumerate((ks[t], vs[t])):
            torch.manual_seed(G.SEED_Q + 2 * t + kv)
            pkt = cm.compact_compress(f"0-0-{'kv'[kv]}", x.view(1, G.N, G.HEADS, G.HD), typ, update_cache=True)
            r = cm.compact_decompress(f"0-1-{'kv'[kv]}", pkt.clone(), typ, (1, G.N, G.HEADS, G.HD), update_cache=True)
            rec.append(r.reshape(G.N, G.C).clone())
        rows.append(G.metrics(q, ks[t], vs[t], rec[0], rec[1]))
    np.save(sys.argv[2], np.array(rows, dtype=np.float64))


def main():
    traces = {}
    for mode, dis in (("eager", "1"), ("compiled", "0")):
        out = f"/tmp/cfx_lrq_{mode}.npy"
        env = dict(os.environ, TORCHDYNAMO_DISABLE=dis, TRITON_INTERPRET="1")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", out], check=True, env=env)
        traces[mode] = np.load(out)
    e, c = traces["eager"], traces["compiled"]
    gold = np.load(os.path.join(HERE, "g12_quality.npz"))["lrq32/trace"]
    rel = np.abs(e[1:, :2] - c[1:, :2]) / e[1:, :2]
    spread = {"steps": int(e.shape[0]),
              "eager_equals_the_committed_golden": bool(np.array_equal(e, gold)),
              "max_rel_err_difference_relative": float(rel.max()),
              "max_psnr_difference_db": float(np.abs(e[1:, 2] - c[1:, 2]).max()),
              "mean_rel_err_difference_relative": float(rel.mean()),
              "rel_err_k_last_step": {"eager": float(e[-1, 0]), "compiled": float(c[-1, 0])},
              "psnr_last_step_db": {"eager": float(e[-1, 2]), "compiled": float(c[-1, 2])},
              "what": "reference LOW_RANK_Q r=32, residual 1 + EF, 28-step drift trace of make_golden_quality.py, eager (TORCHDYNAMO_DISABLE=1) vs "
                      "@torch.compile (inductor on the CPU): the reference's own two execution modes"}
    np.savez_compressed(os.path.join(HERE, "g12_lrq32_modes.npz"), eager=e, compiled=c)
    with open(os.path.join(HERE, "g12_lrq32_modes.json"), "w") as f:
        json.dump(spread, f, indent=1)
    print(json.dumps(spread, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one_mode()
    else:
        main()
