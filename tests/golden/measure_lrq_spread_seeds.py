"""LOW_RANK_Q-32 on EIGHT more seeds of the G12 trace, in the reference's two execution modes.  (BUILD container only: imports /root/reference.)

tests/golden/measure_lrq_spread.py measured, on the one committed G12 trace, how far the reference's eager and @torch.compile modes are apart
(5.8e-3 relative in the reconstruction error, 0.26 dB in attention-output PSNR) and tests/test_gpu_quality.py holds the HIP path inside that
band.  One trace says little about margin: LOW_RANK_Q quantises fp16 factors to 16 levels, a last-bit difference flips whole levels, error
feedback carries the flip on - the distance between any two correct implementations is a random variable.  This script draws 8 further
input / start-matrix seeds (the recipe of make_golden_quality.py with other seeds), runs the reference's trace once per mode and seed, and
writes
    tests/golden/g12_lrq32_seeds.npz      eager[s], compiled[s]: [step][rel_err_k, rel_err_v, attention PSNR dB], seeds[s] = (SEED_X, SEED_Q)
    tests/golden/g12_lrq32_seeds.json     per seed: the two-mode spread (max over steps, K and V separately, PSNR)
tests/test_gpu_quality.py::test_lowrank_q_gap_over_seeds runs the HIP path on the same seeds and compares ITS distance from the eager trace
with the reference's own eager-to-compiled distance, seed by seed.

usage: python tests/golden/measure_lrq_spread_seeds.py          (about 15 minutes: inductor compiles on the CPU for every seed)
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NSEEDS = 8


def seeds(s):
    """(SEED_X, SEED_Q) of extra seed s (the committed G12 trace is make_golden_quality's own pair)"""
    return 4242 + 1000 * (s + 1), 900000 + 100000 * (s + 1)


def one_mode():
    s, out = int(sys.argv[2]), sys.argv[3]
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
    sx, sq = seeds(s)
    ks, vs, q = G.drift(sx, Tn), G.drift(sx + 1, Tn), G.query(sx + 2)
    tname, kw = G.PRESETS["lrq32"]
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: None, residual=1, ef=True, simulate=False, log_stats=False, **kw))
    rows = []
    for t in range(Tn):
        typ = T.WARMUP if t == 0 else T[tname]
        rec = []
        for kv, x in enumerate((ks[t], vs[t])):
            torch.manual_seed(sq + 2 * t + kv)
            pkt = cm.compact_compress(f"0-0-{'kv'[kv]}", x.view(1, G.N, G.HEADS, G.HD), typ, update_cache=True)
            r = cm.compact_decompress(f"0-1-{'kv'[kv]}", pkt.clone(), typ, (1, G.N, G.HEADS, G.HD), update_cache=True)
            rec.append(r.reshape(G.N, G.C).clone())
        rows.append(G.metrics(q, ks[t], vs[t], rec[0], rec[1]))
    np.save(out, np.array(rows, dtype=np.float64))


def gap(a, b):
    """distance of trace a from trace b as the quality test measures it: max over steps of the relative difference of the reconstruction
    errors (K, V), of the PSNR difference"""
    rel = np.abs(a[1:, :2] - b[1:, :2]) / b[1:, :2]
    return float(rel[:, 0].max()), float(rel[:, 1].max()), float(np.abs(a[1:, 2] - b[1:, 2]).max())


def main():
    eager, compiled, per = [], [], []
    for s in range(NSEEDS):
        tr = {}
        for mode, dis in (("eager", "1"), ("compiled", "0")):
            out = f"/tmp/cfx_lrq_seed{s}_{mode}.npy"
            env = dict(os.environ, TORCHDYNAMO_DISABLE=dis, TRITON_INTERPRET="1")
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(s), out], check=True, env=env)
            tr[mode] = np.load(out)
        eager.append(tr["eager"]); compiled.append(tr["compiled"])
        gk, gv, gp = gap(tr["compiled"], tr["eager"])
        per.append({"seed_x": seeds(s)[0], "seed_q": seeds(s)[1], "rel_err_k": gk, "rel_err_v": gv, "psnr_db": gp})
        print(s, per[-1], flush=True)
    np.savez_compressed(os.path.join(HERE, "g12_lrq32_seeds.npz"), eager=np.array(eager), compiled=np.array(compiled),
                        seeds=np.array([seeds(s) for s in range(NSEEDS)], dtype=np.int64))
    summary = {"seeds": per,
               "what": "reference LOW_RANK_Q r=32, residual 1 + EF, the 28-step drift recipe of make_golden_quality.py on 8 further seeds: distance "
                       "of the @torch.compile trace from the eager trace (max over steps of the relative difference of the reconstruction errors "
                       "of K and of V, of the attention-output PSNR difference) - the reference's own two execution modes"}
    with open(os.path.join(HERE, "g12_lrq32_seeds.json"), "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one_mode()
    else:
        main()
