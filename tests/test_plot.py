"""compact/plot.py - the statistics logger's figures (reference xfuser/compact/plot.py:1-558): file names, the four key / step selections,
the cumulative-share arithmetic, the per-step dumps.  Host-side only (matplotlib, Agg)."""
import os

import numpy as np
import pytest
import torch

pytest.importorskip("matplotlib")


def _spectra():
    g = np.random.default_rng(3)
    mk = lambda n: np.sort(np.abs(g.standard_normal(n)))[::-1].astype(np.float32)
    return {"0-0-k": {1: {"activation": [mk(32)], "delta": [mk(32)], "delta_delta": []}, 3: {"activation": [mk(32)], "delta": [mk(32)], "delta_delta": []}},
            "5-1-v": {1: {"activation": [mk(16)], "delta": [], "delta_delta": []}}}


def test_cumulative_share_and_gaussian_reference():
    from compactfusion_amd.compact import plot as P
    c = P._cdf([1.0, 3.0, 2.0, 2.0])
    assert np.allclose(c, [3 / 8, 5 / 8, 7 / 8, 1.0])                        # descending order, normalised (plot.py:129-130)
    assert P._cdf([0.0, 0.0]).tolist() == [0.0, 0.0]
    g = P.gaussian_reference_cdf()
    assert g.shape == (2176,) and abs(g[-1] - 1.0) < 1e-9 and np.all(np.diff(g) > 0) and g is P.gaussian_reference_cdf()
    # a standard normal matrix's spectrum is flat-ish: the top tenth of the singular values carries well under a third
    assert 0.1 < g[217] < 0.33


def test_figures_for_every_key_step_selection(tmp_path, capsys, monkeypatch):
    from compactfusion_amd.compact import plot as P
    monkeypatch.setattr(P, "_gauss_cdf", np.linspace(0.01, 1.0, 64))          # (skip the 2176 x 3072 SVD here)
    e = _spectra()
    d = str(tmp_path)
    w = P.plot_eigenvalue_cumsum(e, save_dir=d)                               # all keys, all steps: three figures (one spectrum list is empty-free)
    assert sorted(os.path.basename(x) for x in w) == ["0-0-k_activation_cdf_step1.png", "0-0-k_activation_cdf_step3.png", "5-1-v_activation_cdf_step1.png"]
    assert all(os.path.getsize(x) > 1000 for x in w)
    assert [os.path.basename(x) for x in P.plot_eigenvalue_cumsum(e, key="0-0-k", data_type="delta", save_dir=d)] == \
        ["0-0-k_delta_cdf_step1.png", "0-0-k_delta_cdf_step3.png"]
    assert [os.path.basename(x) for x in P.plot_eigenvalue_cumsum(e, step=3, save_dir=d)] == ["0-0-k_activation_cdf_step3.png"]
    assert [os.path.basename(x) for x in P.plot_eigenvalue_distribution(e, key="5-1-v", step=1, save_dir=d, num_bins=8)] == ["5-1-v_activation_step1.png"]
    capsys.readouterr()
    assert P.plot_eigenvalue_cumsum(e, key="nope", save_dir=d) == [] and "No eigenvalue data for key nope." in capsys.readouterr().out
    assert P.plot_eigenvalue_cumsum(e, key="0-0-k", step=2, save_dir=d) == [] and "and step 2" in capsys.readouterr().out
    assert P.plot_eigenvalue_distribution(e, key="5-1-v", step=1, data_type="delta", save_dir=d) == []
    assert "No delta eigenvalue data for key 5-1-v and step 1." in capsys.readouterr().out
    assert P.plot_eigenvalue_cumsum({}, save_dir=d) == [] and "No eigenvalue data available." in capsys.readouterr().out


def test_factor_and_surface_figures(tmp_path):
    from compactfusion_amd.compact import plot as P
    u, v = torch.randn(40, 8), torch.randn(8, 64)
    p = P.plot_low_rank_factors(u, v, "3-0-k", 2, str(tmp_path))
    assert os.path.basename(p) == "3-0-k_step2_uv.png" and os.path.getsize(p) > 1000
    with pytest.raises(ValueError):
        P.plot_low_rank_factors(u, v, "3-0-k", None, str(tmp_path))
    f = str(tmp_path / "surface.png")
    P.plot_3d(torch.randn(12, 20), "t", filename=f)
    assert os.path.getsize(f) > 1000


def test_dumps_equal_the_loggers_own(tmp_path):
    from compactfusion_amd.compact import plot as P
    from compactfusion_amd.compact.stats import StatsLogger
    rows = lambda n, o: [{"error": 0.1 * (i + o), "total_error": None if i == 0 else 0.2 * i, "activation_norm": 1.0 + i, "delta_norm": 0.5 * i,
                          "activation_similarity": 0.9} for i in range(n)]
    stats = {"0-0-k": rows(3, 0), "0-0-v": rows(2, 1)}
    a = P.dump_average_error_vs_steps(stats, str(tmp_path / "a"))
    assert a["steps"] == [0, 1, 2] and np.allclose(a["avg_comp_errors"], [0.05, 0.15, 0.2]) and a["avg_total_errors"][0] is None
    assert np.allclose(a["avg_total_errors"][1:], [0.2, 0.4])
    b = P.dump_average_norms_and_similarity_vs_steps(stats, str(tmp_path / "a"))
    assert np.allclose(b["avg_act_norms"], [1.0, 2.0, 3.0]) and np.allclose(b["avg_delta_norms"], [0.0, 0.5, 1.0])
    assert torch.load(str(tmp_path / "a" / "average_error_vs_steps.pt"))["avg_comp_errors"] == a["avg_comp_errors"]
    lg = StatsLogger()
    lg.stats = stats
    assert lg.dump_average_error_vs_steps(str(tmp_path / "b"))["avg_comp_errors"] == a["avg_comp_errors"]
    assert P.dump_average_error_vs_steps({}, str(tmp_path / "c")) is None


def test_logger_hooks_reach_the_figures(tmp_path, monkeypatch):
    from compactfusion_amd.compact import plot as P, stats as S
    monkeypatch.setattr(P, "_gauss_cdf", np.linspace(0.01, 1.0, 64))
    lg = S.StatsLogger()
    lg.eigenvalues = _spectra()
    assert len(lg.plot_eigenvalue_cumsum(save_dir=str(tmp_path))) == 3
    assert len(lg.plot_eigenvalue_distribution(key="0-0-k", save_dir=str(tmp_path))) == 2
    monkeypatch.setattr(S, "_stats", lg)
    assert len(S.plot_eigenvalues(key="0-0-k", step=1, save_dir=str(tmp_path), cum_sum=True)) == 1
    # the module constants select layers / steps (empty by default, as upstream: stats.py:13-16)
    assert lg.plot_low_rank_factors(torch.randn(16, 4), torch.randn(4, 32), "10-0-k", 10, str(tmp_path)) is None
    monkeypatch.setattr(S, "UV_PLOT_LAYERS", [10, 20])
    monkeypatch.setattr(S, "UV_PLOT_STEPS", [10])
    layer, step = 10, 10
    p = lg.plot_low_rank_factors(torch.randn(16, 4), torch.randn(4, 32), f"{layer}-0-k", step, str(tmp_path))
    assert p and os.path.exists(p) and os.path.exists(os.path.join(str(tmp_path), f"uv_{layer}-0-k_{step}.pt"))
    assert lg.plot_low_rank_factors(torch.randn(16, 4), torch.randn(4, 32), "99999-0-k", step, str(tmp_path)) is None
