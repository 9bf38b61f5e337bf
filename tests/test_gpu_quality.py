"""Quality band without model weights (-m gpu): the HIP path against the REFERENCE's error-vs-step trace (golden G12,
tests/golden/make_golden_quality.py: the reference's compact_compress / compact_decompress over a 28-step drift at (128, 3072)
for its shipped presets, examples/configs.py:39-98).  Per step the relative reconstruction error of K and V and the PSNR of the
attention output must stay within 1e-3 relative (north-star fp tolerance) of the reference's.  BASELINE.json's images-per-second
/ PSNR-LPIPS half needs FLUX weights, which do not exist on the box; this pins the codec-level quantity that drives it."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden_quality", os.path.join(HERE, "golden", "make_golden_quality.py"))
GEN = importlib.util.module_from_spec(spec)
spec.loader.exec_module(GEN)          # only its seeded input recipe and metric definitions are used; main() needs the reference
GOLD = os.path.join(HERE, "golden", "g12_quality.npz")


def _trace(name):
    if not os.path.exists(GOLD):
        pytest.skip("G12 golden vectors not generated")
    g = np.load(GOLD)
    if f"{name}/trace" not in g.files:
        pytest.skip(f"G12 has no {name} trace")
    return g[f"{name}/trace"]


def _inputs(T):
    import hashlib
    meta = json.load(open(os.path.join(HERE, "golden", "g12_quality_meta.json")))
    ks, vs, q = GEN.drift(GEN.SEED_X, meta["steps"]), GEN.drift(GEN.SEED_X + 1, meta["steps"]), GEN.query(GEN.SEED_X + 2)
    sha = lambda t: hashlib.sha256(t.contiguous().view(torch.int16).numpy().tobytes()).hexdigest()   # noqa: E731
    assert sha(ks[-1]) == meta["sha_k_last"] and sha(vs[-1]) == meta["sha_v_last"] and sha(q) == meta["sha_q"], \
        "torch CPU RNG produced different inputs than when G12 was captured"
    return ks[:T], vs[:T], q


# LOW_RANK_Q re-quantises the factors to int4 (16 levels over the column range): a last-bit difference in a factor entry flips whole
# quantisation levels, and error feedback carries the flip forward.  The reference itself does not reproduce this trace more closely
# than that: its two execution modes - eager, in which the goldens were captured, and @torch.compile, which it runs on its own hardware
# (compress_lowrank.py:14, compress_quantize.py:522-640) - differ on this very trace by up to 5.8e-3 relative in the reconstruction error
# and 0.26 dB in attention-output PSNR (tests/golden/measure_lrq_spread.py -> g12_lrq32_modes.json, measured by importing the reference).
# The HIP path (Cholesky-QR in fp64 where the reference runs Householder QR in fp32) is held to the eager golden within THAT band - the
# reference's own reproducibility - and every other preset to the north-star 1e-3.
def _lrq_band():
    with open(os.path.join(HERE, "golden", "g12_lrq32_modes.json")) as f:
        m = json.load(f)
    assert m["eager_equals_the_committed_golden"]
    return m["max_rel_err_difference_relative"], m["max_psnr_difference_db"]


TOL = {"lrq32": _lrq_band()}
GAPS = {}


def _check(name, rows, want, who="hip"):
    rows = np.array(rows)
    tol_rel, tol_db = TOL.get(name, (1e-3, 0.02))
    assert rows.shape == want.shape
    assert np.all(rows[0, :2] == 0) and rows[0, 2] > 200            # WARMUP step is exact
    gaps = [float((np.abs(rows[1:, c] - want[1:, c]) / want[1:, c]).max()) for c in (0, 1)] + [float(np.abs(rows[1:, 2] - want[1:, 2]).max())]
    GAPS[f"{name}/{who}"] = {"steps": int(rows.shape[0]), "rel_err_k": gaps[0], "rel_err_v": gaps[1], "psnr_db": gaps[2], "band": [tol_rel, tol_db]}
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "quality_gaps.json"), "w") as f:
            json.dump(GAPS, f, indent=1)
    for col, what in ((0, "relative error of K"), (1, "relative error of V")):
        rel = np.abs(rows[1:, col] - want[1:, col]) / want[1:, col]
        assert rel.max() < tol_rel, f"{name}: {what} departs from the reference trace by {rel.max():.2e} (step {1 + int(rel.argmax())})"
    # PSNR is 10 log10 of a squared error: 1e-3 relative in the error = 0.0087 dB
    d = np.abs(rows[1:, 2] - want[1:, 2])
    assert d.max() < tol_db, f"{name}: attention-output PSNR departs from the reference trace by {d.max():.3f} dB"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["binary", "int2", "lr8", "lrq32"])
def test_hip_path_stays_on_the_reference_quality_trace(name, tmp_path):
    want = _trace(name)
    T = want.shape[0]
    ks, vs, q = _inputs(T)
    from compactfusion_amd.collector import collector
    collector.init(collector.Collector(str(tmp_path), enabled=False))
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as CT, CompactConfig, lowrank
    tname, kw = GEN.PRESETS[name]
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: None, residual=1, ef=True, simulate=False, log_stats=False, **kw))
    rows = []
    try:
        for t in range(T):
            typ = CT.WARMUP if t == 0 else CT[tname]
            rec = []
            for kv, x in enumerate((ks[t], vs[t])):
                if tname.startswith("LOW_RANK"):
                    torch.manual_seed(GEN.SEED_Q + 2 * t + kv)      # the start matrix the reference drew (compress_lowrank.py:41)
                    lowrank.set_init_q(torch.randn(GEN.C, kw["comp_rank"], dtype=torch.float))
                skey, rkey = f"0-0-{'kv'[kv]}", f"0-1-{'kv'[kv]}"
                pkt = cm.compact_compress(skey, x.cuda().view(1, GEN.N, GEN.HEADS, GEN.HD), typ, update_cache=True)
                r = cm.compact_decompress(rkey, pkt.clone(), typ, (1, GEN.N, GEN.HEADS, GEN.HD), update_cache=True)
                rec.append(r.reshape(GEN.N, GEN.C).cpu().clone())
                assert torch.equal(cm.compact_cache().get_base(skey), cm.compact_cache().get_base(rkey)), "sender / receiver diverged"
            rows.append(GEN.metrics(q, ks[t], vs[t], rec[0], rec[1]))
    finally:
        lowrank.set_init_q(None)
    _check(name, rows, want)


def _hip_trace(name, ks, vs, q, seed_q, tmp_path):
    """The HIP path's [step][rel_err_k, rel_err_v, PSNR] on a drift (the loop of test_hip_path_stays_on_the_reference_quality_trace)."""
    from compactfusion_amd.collector import collector
    collector.init(collector.Collector(str(tmp_path), enabled=False))
    import compactfusion_amd.compact.main as cm
    from compactfusion_amd.compact import COMPACT_COMPRESS_TYPE as CT, CompactConfig, lowrank
    tname, kw = GEN.PRESETS[name]
    cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: None, residual=1, ef=True, simulate=False, log_stats=False, **kw))
    rows = []
    try:
        for t in range(len(ks)):
            typ = CT.WARMUP if t == 0 else CT[tname]
            rec = []
            for kv, x in enumerate((ks[t], vs[t])):
                torch.manual_seed(seed_q + 2 * t + kv)
                lowrank.set_init_q(torch.randn(GEN.C, kw["comp_rank"], dtype=torch.float))
                skey, rkey = f"0-0-{'kv'[kv]}", f"0-1-{'kv'[kv]}"
                pkt = cm.compact_compress(skey, x.cuda().view(1, GEN.N, GEN.HEADS, GEN.HD), typ, update_cache=True)
                r = cm.compact_decompress(rkey, pkt.clone(), typ, (1, GEN.N, GEN.HEADS, GEN.HD), update_cache=True)
                rec.append(r.reshape(GEN.N, GEN.C).cpu().clone())
            rows.append(GEN.metrics(q, ks[t], vs[t], rec[0], rec[1]))
    finally:
        lowrank.set_init_q(None)
    return np.array(rows)


@pytest.mark.gpu
def test_lowrank_q_gap_over_seeds(tmp_path):
    """LOW_RANK_Q-32 on 8 more seeds of the G12 recipe (tests/golden/measure_lrq_spread_seeds.py ran the REFERENCE on them in both of its
    execution modes).  A last-bit difference in a factor flips whole int4 levels and error feedback carries the flip on, so the distance
    between two correct implementations is a random variable; the one committed trace of G12 (HIP 3.0e-3 / 5.3e-3 for K / V against a band
    of 5.8e-3) says little about margin.  Here, seed by seed: the HIP path's distance from the reference's eager trace beside the
    reference's OWN eager-to-compiled distance.  Asserted: on every seed the HIP distance is inside the range of the reference's own
    distances (its largest over the seeds), and the MEDIAN HIP distance is not above the median reference distance - i.e. the HIP path
    is as close to the eager reference as the reference's compiled mode is.  The table goes to gpurun_out/lrq32_seeds.json
    (committed as profiles/r06_lrq32_seeds.json)."""
    f = os.path.join(HERE, "golden", "g12_lrq32_seeds.npz")
    if not os.path.exists(f):
        pytest.skip("g12_lrq32_seeds.npz not generated")
    g = np.load(f)
    eager, compiled, seeds = g["eager"], g["compiled"], g["seeds"]

    def gap(a, b):
        rel = np.abs(a[1:, :2] - b[1:, :2]) / b[1:, :2]
        return [float(rel[:, 0].max()), float(rel[:, 1].max()), float(np.abs(a[1:, 2] - b[1:, 2]).max())]
    rows = []
    for s in range(len(seeds)):
        sx, sq = int(seeds[s][0]), int(seeds[s][1])
        T = eager[s].shape[0]
        ks, vs, q = GEN.drift(sx, T), GEN.drift(sx + 1, T), GEN.query(sx + 2)
        hip = _hip_trace("lrq32", ks, vs, q, sq, tmp_path)
        assert np.all(hip[0, :2] == 0)
        rows.append({"seed_x": sx, "hip_vs_eager": gap(hip, eager[s]), "compiled_vs_eager": gap(compiled[s], eager[s]),
                     "hip_vs_compiled": gap(hip, compiled[s])})
    H = np.array([r["hip_vs_eager"] for r in rows])
    Rf = np.array([r["compiled_vs_eager"] for r in rows])
    summary = {"columns": ["rel_err_k", "rel_err_v", "psnr_db"], "seeds": rows,
               "hip_vs_eager": {"median": np.median(H, 0).tolist(), "max": H.max(0).tolist()},
               "reference_compiled_vs_eager": {"median": np.median(Rf, 0).tolist(), "max": Rf.max(0).tolist()},
               "ratio_of_medians": (np.median(H, 0) / np.median(Rf, 0)).tolist(),
               "what": "distance = max over the 27 compressed steps of |err - err_ref| / err_ref (reconstruction error of K, of V at the receiver) and "
                       "of the attention-output PSNR difference in dB; reference traces: tests/golden/g12_lrq32_seeds.npz"}
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "lrq32_seeds.json"), "w") as fo:
            json.dump(summary, fo, indent=1)
    for c, what in enumerate(("relative error of K", "relative error of V", "attention PSNR")):
        assert H[:, c].max() <= Rf[:, c].max() * 1.0, f"LOW_RANK_Q {what}: HIP is further from the eager reference ({H[:, c].max():.2e}) than the reference's own modes ever are ({Rf[:, c].max():.2e})"
        assert np.median(H[:, c]) <= np.median(Rf[:, c]) * 1.0, f"LOW_RANK_Q {what}: median HIP distance {np.median(H[:, c]):.2e} above the reference's own {np.median(Rf[:, c]):.2e}"


@pytest.mark.parametrize("name", ["binary", "int2"])
def test_oracle_stays_on_the_reference_quality_trace(name):
    """The CPU oracle on the same trace (keeps the oracle pinned to the reference over a long error-feedback chain)."""
    from oracle import ref_np as R
    want = _trace(name)
    T = min(want.shape[0], 10)
    ks, vs, q = _inputs(T)
    rows, state = [], [None, None]
    for t in range(T):
        rec = []
        for kv, x in enumerate((ks[t], vs[t])):
            xa = x.numpy()
            if t == 0:
                state[kv] = xa.copy()
            else:
                _, state[kv] = R.residual_compress(name, xa, state[kv], 0)
            rec.append(torch.from_numpy(R.bits(state[kv]).view(np.int16).copy()).view(torch.float16).reshape(GEN.N, GEN.C))
        rows.append(GEN.metrics(q, ks[t], vs[t], rec[0], rec[1]))
    _check(name, rows, want[:T], who="oracle")
