"""Figures of the statistics logger - the observability row's plotting half (`/root/reference/xfuser/compact/plot.py:1-558`): the spectrum
of a layer's activation / delta / delta-delta as a cumulative curve against a Gaussian matrix's or as a density histogram, the U / V factors of
a low-rank packet as images, a tensor as a surface, and the two per-step dumps.  Same function names, arguments, titles and file names
(`{key}_{data_type}_cdf_step{step}.png`, `{key}_{data_type}_step{step}.png`, `{key}_step{step}_uv.png`, `3d_{title}.png`).

Written around one selector (`_selected`) instead of four copies of the drawing code per function; the reference's loops rebind the spectra
dictionary to the array they just sorted (`plot.py:129`), which ends a multi-key sweep after its first figure - here a sweep draws every figure.
Host-side only: nothing here touches the GPU path; matplotlib is imported on first use with the non-interactive backend when no display exists.
"""
import os
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

PLOT_DIR = "plots"
DATA_TYPES = ("activation", "delta", "delta_delta")
_GAUSS_SHAPE = (2176, 3072)          # the reference's comparison curve: singular values of a standard normal matrix of FLUX's (tokens, channels)
_gauss_cdf = None


def _plt():
    import matplotlib
    if not os.environ.get("DISPLAY") and matplotlib.get_backend().lower() not in ("agg", "pdf", "svg", "ps"):
        matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def _cdf(values) -> np.ndarray:
    v = np.sort(np.asarray(values, dtype=np.float64).ravel())[::-1]
    total = v.sum()
    return np.cumsum(v) / total if total > 0 else np.zeros_like(v)


def gaussian_reference_cdf() -> np.ndarray:
    """Cumulative share of the singular values of a (2176, 3072) standard normal matrix (plot.py:113-116), computed once (seeded: the curve is
    the same in every figure and every run)."""
    global _gauss_cdf
    if _gauss_cdf is None:
        g = torch.Generator().manual_seed(0)
        _gauss_cdf = _cdf(torch.linalg.svdvals(torch.randn(*_GAUSS_SHAPE, generator=g, dtype=torch.float32)).numpy())
    return _gauss_cdf


def _selected(eigenvalues, key, step, data_type) -> Iterator[Tuple[str, int, list]]:
    """(key, step, list of spectra) for one key / one step / both / neither given - the four cases of plot.py:118-267 - with the
    reference's messages for what is missing."""
    if key is not None and key not in eigenvalues:
        print(f"No eigenvalue data for key {key}.")
        return
    for k in ([key] if key is not None else list(eigenvalues)):
        per_step = eigenvalues[k]
        if step is not None and step not in per_step:
            if key is not None:
                print(f"No eigenvalue data for key {k} and step {step}.")
            continue
        for s in ([step] if step is not None else list(per_step)):
            spectra = per_step[s].get(data_type)
            if spectra is None:
                continue
            if not spectra:
                print(f"No {data_type} eigenvalue data for key {k} and step {s}." if (key is not None and step is not None)
                      else f"Skipping empty eigenvalue data for {k}, step {s}, type {data_type}")
                continue
            yield k, s, spectra


def _finish(plt, fig, save_dir, name):
    if save_dir:
        path = os.path.join(save_dir, name)
        fig.savefig(path, dpi=300, bbox_inches="tight")
        print(f"Plot saved to {path}")
        plt.close(fig)
        return path
    plt.show()
    return None


def plot_eigenvalue_cumsum(eigenvalues, key: Optional[str] = None, step: Optional[int] = None, data_type: str = "activation",
                           save_dir: Optional[str] = None, log_scale: bool = True, top_k: Optional[int] = None):
    """Cumulative share of the (descending) spectrum per captured key / step, beside the Gaussian reference curve (plot.py:85-267).
    Returns the paths written."""
    if not eigenvalues:
        print("No eigenvalue data available.")
        return []
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
    plt = _plt()
    written = []
    for k, s, spectra in _selected(eigenvalues, key, step, data_type):
        print(f"Plotting {k} {data_type} CDF for step {s}")
        fig = plt.figure(figsize=(10, 6))
        plt.plot(_cdf(spectra[0]), label=f"Step {s}")                       # (the first spectrum captured at that step, plot.py:129)
        plt.plot(gaussian_reference_cdf(), label="Gaussian distribution")
        plt.title(f"{k} {data_type.capitalize()} Eigenvalue CDF (Step {s})" + (f" (Top {top_k} mentioned)" if top_k is not None else ""))
        plt.ylabel("Cumulative Probability")
        if log_scale:
            plt.xscale("log")
        plt.grid(True, which="both", linestyle="--", linewidth=0.5)
        plt.legend()
        written.append(_finish(plt, fig, save_dir, f"{k}_{data_type}_cdf_step{s}.png"))
    return written


def plot_eigenvalue_distribution(eigenvalues, key: Optional[str] = None, step: Optional[int] = None, data_type: str = "activation",
                                 save_dir: Optional[str] = None, log_scale: bool = True, top_k: Optional[int] = None, num_bins: int = 100):
    """Spectral density histogram per captured key / step (plot.py:269-411).  Returns the paths written."""
    if not eigenvalues:
        print("No eigenvalue data available.")
        return []
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
    plt = _plt()
    written = []
    for k, s, spectra in _selected(eigenvalues, key, step, data_type):
        print(f"Plotting {k} {data_type} spectral density for step {s}")
        fig = plt.figure(figsize=(10, 6))
        plt.hist([np.asarray(v, dtype=np.float64).ravel() for v in spectra], bins=num_bins, density=True, alpha=0.7, log=log_scale)
        plt.title(f"{k} {data_type.capitalize()} Spectral Density (Step {s})" + (f" (Top {top_k} mentioned)" if top_k is not None else ""))
        plt.xlabel("Eigenvalue Magnitude")
        plt.ylabel("Spectral Density")
        plt.grid(True, which="both", linestyle="--", linewidth=0.5)
        written.append(_finish(plt, fig, save_dir, f"{k}_{data_type}_step{s}.png"))
    return written


def plot_low_rank_factors(u: torch.Tensor, v: torch.Tensor, key: str, step: Optional[int], save_dir: Optional[str] = None):
    """U (N, K) and V (K, C) of a LOW_RANK packet side by side as images (plot.py:30-82)."""
    if step is None:
        raise ValueError(f"Step is None for key {key}, cannot save U/V plot with step index.")
    plt = _plt()
    mats = (u.detach().cpu().float().numpy(), v.detach().cpu().float().numpy())
    fig, axes = plt.subplots(1, 2, figsize=(12, 6))
    fig.suptitle(f"Low-Rank Factors for {key} (step{step})")
    for ax, m, name, xl, yl in zip(axes, mats, ("U", "V"), ("Rank (K)", "Channels (C)"), ("Tokens (N)", "Rank (K)")):
        im = ax.imshow(m, aspect="auto", cmap="viridis")
        ax.set_title(f"{name} Matrix (Shape: {m.shape})")
        ax.set_xlabel(xl)
        ax.set_ylabel(yl)
        fig.colorbar(im, ax=ax)
    fig.tight_layout(rect=[0, 0.03, 1, 0.95])
    if not save_dir:
        plt.show()
        return None
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, f"{key}_step{step}_uv.png")
    fig.savefig(path, dpi=150, bbox_inches="tight")
    print(f"Saved U/V plot to {path}")
    plt.close(fig)
    return path


def plot_3d(tensor, title, filename=None):
    """A (tokens, channels) tensor as a surface (plot.py:8-27)."""
    plt = _plt()
    z = tensor.detach().cpu().float().numpy() if isinstance(tensor, torch.Tensor) else np.asarray(tensor, dtype=np.float32)
    fig = plt.figure(figsize=(10, 6))
    ax = fig.add_subplot(111, projection="3d")
    x, y = np.meshgrid(np.arange(z.shape[1]), np.arange(z.shape[0]))
    ax.plot_surface(x, y, z, cmap="coolwarm", linewidth=0, antialiased=False)
    ax.set_xlabel("Channel")
    ax.set_ylabel("Token")
    ax.set_zlabel("Tensor")
    plt.title(title)
    if filename is None:
        os.makedirs(PLOT_DIR, exist_ok=True)
        filename = f"{PLOT_DIR}/3d_{title}.png"
    fig.savefig(filename, dpi=300, bbox_inches="tight")
    plt.close(fig)
    return fig, ax


def _per_step_mean(stats_data, field):
    n = max((len(rows) for rows in stats_data.values()), default=0)
    out = []
    for s in range(n):
        vals = [rows[s][field] for rows in stats_data.values() if s < len(rows) and rows[s].get(field) is not None]
        out.append(float(np.mean(vals)) if vals else None)
    return out


def dump_average_error_vs_steps(stats_data, save_dir: str):
    """Per step, over all keys: mean compression error and mean total error -> `average_error_vs_steps.pt` (plot.py:413-478)."""
    if not stats_data:
        print("Error: No statistics data provided. Cannot dump error data.")
        return None
    comp, total = _per_step_mean(stats_data, "error"), _per_step_mean(stats_data, "total_error")
    data = {"steps": list(range(len(comp))), "avg_comp_errors": comp, "avg_total_errors": total}
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, "average_error_vs_steps.pt")
    torch.save(data, path)
    print(f"Saved average error data to {path}")
    return data


def dump_average_norms_and_similarity_vs_steps(stats_data, save_dir: str):
    """Per step, over all keys: mean activation norm, delta norm, activation similarity -> `average_norms_and_similarity_vs_steps.pt`
    (plot.py:481-558)."""
    if not stats_data:
        print("Error: No statistics data provided. Cannot dump norms/similarity data.")
        return None
    act = _per_step_mean(stats_data, "activation_norm")
    data = {"steps": list(range(len(act))), "avg_act_norms": act, "avg_delta_norms": _per_step_mean(stats_data, "delta_norm"),
            "avg_act_similarities": _per_step_mean(stats_data, "activation_similarity")}
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, "average_norms_and_similarity_vs_steps.pt")
    torch.save(data, path)
    print(f"Saved average norms and similarity data to {path}")
    return data
