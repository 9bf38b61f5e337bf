"""StatsLogger (compactfusion_amd/compact/stats.py) against golden G11: the reference's StatsLogger records for seeded
call sequences at residual 0 / 1 / 2 (tests/golden/make_golden_stats.py regenerates them from the reference)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "g11_stats.npz"))
spec = importlib.util.spec_from_file_location("make_golden_stats", os.path.join(HERE, "golden", "make_golden_stats.py"))
GEN = importlib.util.module_from_spec(spec)
spec.loader.exec_module(GEN)          # only its seeded input recipe and field list are used; main() needs the reference

LOWRANK = GEN.FIELDS.index("delta_before_feedback_lowrank_similarity")


@pytest.fixture()
def stats(monkeypatch):
    from compactfusion_amd.compact import stats as S
    monkeypatch.setattr(S, "CALC_SIMILARITY", True)
    monkeypatch.setattr(S, "CALC_MORE_SIMILARITY", True)
    S.stats_clear()
    yield S
    S.stats_clear()


@pytest.mark.parametrize("residual", [0, 1, 2])
def test_records_match_reference(stats, residual, tmp_path):
    for call in GEN.inputs(residual):
        stats.log(*call, residual)
    lg = stats.stats_log()
    for k in GEN.KEYS:
        want = G[f"r{residual}/{k}"]
        got = np.array([[np.nan if row[f] is None else float(row[f]) for f in GEN.FIELDS] for row in lg.stats[k]])
        assert got.shape == want.shape
        assert np.array_equal(np.isnan(got), np.isnan(want)), (k, "presence of optional fields differs")
        cols = [i for i in range(len(GEN.FIELDS)) if i != LOWRANK]
        np.testing.assert_allclose(got[:, cols], want[:, cols], rtol=1e-5, atol=1e-6, equal_nan=True)
        lr = got[:, LOWRANK]                      # random start matrix inside: range check only
        assert np.all(np.isnan(lr) | ((lr >= -1.0001) & (lr <= 1.0001)))
    assert [lg.total_original_volume, lg.total_compressed_volume] == list(G[f"r{residual}/volumes"])
    e = lg.dump_average_error_vs_steps(str(tmp_path))
    n = lg.dump_average_norms_and_similarity_vs_steps(str(tmp_path))
    enc = lambda xs: np.array([np.nan if v is None else v for v in xs], dtype=np.float64)   # noqa: E731
    np.testing.assert_allclose(np.stack([enc(e["avg_comp_errors"]), enc(e["avg_total_errors"])]), G[f"r{residual}/dump_err"], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(np.stack([enc(n["avg_act_norms"]), enc(n["avg_delta_norms"]), enc(n["avg_act_similarities"])]),
                               G[f"r{residual}/dump_norms"], rtol=1e-6, equal_nan=True)
    # the files are what the reference's analysis scripts load
    back = torch.load(os.path.join(tmp_path, "average_error_vs_steps.pt"))
    assert set(back) == {"steps", "avg_comp_errors", "avg_total_errors"} and back["steps"] == list(range(GEN.STEPS))
    back = torch.load(os.path.join(tmp_path, "average_norms_and_similarity_vs_steps.pt"))
    assert set(back) == {"steps", "avg_act_norms", "avg_delta_norms", "avg_act_similarities"}
    x0 = GEN.inputs(residual)[0][3]
    np.testing.assert_allclose([lg._compute_strided_row_similarity(x0.float(), s) for s in (1, 3)], G[f"r{residual}/row_sim"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(lg._compute_eigenvalues(x0), G[f"r{residual}/svals"], rtol=1e-4)


def test_summaries_and_spectra(stats, capsys, tmp_path, monkeypatch):
    for call in GEN.inputs(2):
        stats.log(*call, 2)
    lg = stats.stats_log()
    t = lg.total_averages()
    rows = [r for rs in lg.stats.values() for r in rs]
    assert t["delta_delta_norm"] == pytest.approx(np.mean([r["delta_delta_norm"] for r in rows]))
    assert t["activation_norm"] == pytest.approx(np.mean([np.mean([r["activation_norm"] for r in rs]) for rs in lg.stats.values()]))
    assert t["relative_error"] == pytest.approx(t["error"] / t["activation_norm"])
    stats.stats_verbose()
    stats.stats_verbose_steps(steps=[0, 4, 9], keys=["3-0-k"])
    out = capsys.readouterr().out
    assert "[3-0-k] res=2 (over 5 steps)" in out and "dd/d=" in out and "Ratio" in out and "avg comp error" in out
    assert "=== Step 4 ===" in out and "Step 9 is out of range" in out
    v = lg.compression_volume()
    assert v["ratio"] == pytest.approx(v["raw_bytes"] / v["wire_bytes"])
    # spectra are captured only for the configured steps / layers and saved one file per (key, step, kind)
    monkeypatch.setattr(stats, "EIGENVALUES_PLOT_STEPS", [1])
    monkeypatch.setattr(stats, "EIGENVALUES_PLOT_LAYERS", [3])
    stats.stats_clear()
    for call in GEN.inputs(1)[:6]:
        stats.log(*call, 1)
    lg = stats.stats_log()
    assert set(lg.eigenvalues) == {"3-0-k", "3-0-v"} and set(lg.eigenvalues["3-0-k"]) == {1}
    assert len(lg.eigenvalues["3-0-k"][1]["delta"][0]) == min(GEN.SHAPE)
    stats.save_eigenvalues(str(tmp_path / "eig"))
    assert sorted(os.listdir(tmp_path / "eig")) == sorted(f"{k}_1_{kind}.pt" for k in ("3-0-k", "3-0-v") for kind in ("activation", "delta", "delta_delta"))
    with pytest.raises(ValueError):
        stats.log("3-0-k", None, None, call[3], call[4], call[5], 3)


def test_dump_and_total_error_against_dumped_activations(stats, tmp_path, monkeypatch):
    """DUMP_ACTIVATIONS in one run, CALC_TOTAL_ERROR in the next: total_error = |recv - dumped activation| (stats.py:139-166)."""
    monkeypatch.setattr(stats, "REF_ACTIVATION_PATH", str(tmp_path / "acts"))
    monkeypatch.setattr(stats, "DUMP_ACTIVATIONS", True)
    calls = GEN.inputs(1)[:6]
    for call in calls:
        stats.log(*call, 1)
    assert sorted(os.listdir(tmp_path / "acts")) == sorted(f"{k}_step{s}.pt" for k in GEN.KEYS for s in (0, 1))
    stats.stats_clear()
    monkeypatch.setattr(stats, "DUMP_ACTIVATIONS", False)
    monkeypatch.setattr(stats, "CALC_TOTAL_ERROR", True)
    for call in calls:
        stats.log(*call, 1)
    lg = stats.stats_log()
    for k, base, dbase, x, recv, comp in calls[:3]:
        assert lg.stats[k][0]["total_error"] == pytest.approx(float(torch.norm(recv - x)))
        assert lg.stats[k][0]["total_error"] == pytest.approx(lg.stats[k][0]["error"])
