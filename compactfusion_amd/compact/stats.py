"""Compression statistics: per key and step the reconstruction error, norms, step-to-step similarities and wire volume.

Mirror of the reference's `xfuser/compact/stats.py` observability surface (StatsLogger :37-671 and the module functions
:673-771) as far as numbers go: `StatsLogger.stats[key]` is a list of per-step dicts with the reference's field names,
the volume counters, the environment switches (CALC_SIMILARITY, CALC_MORE_SIMILARITY, PRINT_ALL_ERROR,
REF_ACTIVATION_PATH, DUMP_ACTIVATIONS, CALC_TOTAL_ERROR), the singular-value capture and the two `.pt` dumps
(`average_error_vs_steps.pt`, `average_norms_and_similarity_vs_steps.pt`, same keys as `plot.py:413-560` writes) are
kept, so the reference's analysis scripts read our output.  Pinned by golden G11 (`tests/golden/g11_stats.npz`).
The matplotlib figures are `compact/plot.py` (`plot_eigenvalues`, `plot_low_rank_factors` call into it).

Everything here is diagnostics run with `CompactConfig(log_stats=True)`; it is torch glue, never on the timed path.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch

# steps / layers whose singular-value spectra are captured (empty by default, as in the reference :8-11)
EIGENVALUES_PLOT_STEPS: List[int] = []
EIGENVALUES_PLOT_LAYERS: List[int] = []
UV_PLOT_STEPS: List[int] = []
UV_PLOT_LAYERS: List[int] = []


def _flag(name: str) -> bool:
    return os.environ.get(name, "0") == "1"


CALC_SIMILARITY = _flag("CALC_SIMILARITY")
CALC_MORE_SIMILARITY = _flag("CALC_MORE_SIMILARITY")
PRINT_ALL_ERROR = _flag("PRINT_ALL_ERROR")
REF_ACTIVATION_PATH = os.environ.get("REF_ACTIVATION_PATH", "ref_activations")
DUMP_ACTIVATIONS = _flag("DUMP_ACTIVATIONS")
CALC_TOTAL_ERROR = _flag("CALC_TOTAL_ERROR")

_FIELDS = ("error", "total_error", "activation_norm", "delta_norm", "delta_delta_norm", "delta_before_feedback_norm",
           "activation_similarity", "delta_similarity", "delta_before_feedback_similarity",
           "delta_before_feedback_lowrank_similarity", "transmitted_delta_similarity", "residual", "original_size_bytes",
           "compressed_size_bytes")


def stats_hello():
    print("--- statistics configuration ---")
    for name in ("CALC_SIMILARITY", "CALC_MORE_SIMILARITY", "PRINT_ALL_ERROR", "REF_ACTIVATION_PATH", "DUMP_ACTIVATIONS",
                 "CALC_TOTAL_ERROR"):
        print(f"{name}: {globals()[name]}")
    print("--------------------------------")


def _cos(a: torch.Tensor, b: torch.Tensor) -> float:
    """Cosine similarity of two tensors taken as flat vectors (eps 1e-8, as torch's functional form)."""
    return float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten().to(a.device), dim=0, eps=1e-8))


def _mean(values):
    values = [v for v in values if v is not None]
    return float(np.mean(values)) if values else None


class StatsLogger:
    """Per-key lists of per-step records (`stats`), total volumes, and the previous step's tensors needed for the
    step-to-step similarities (kept on the CPU)."""

    def __init__(self):
        self.stats: Dict[str, List[dict]] = {}
        self.prev_activations: Dict[str, torch.Tensor] = {}
        self.prev_deltas: Dict[str, torch.Tensor] = {}
        self.prev_transmitted_deltas: Dict[str, torch.Tensor] = {}
        self.prev_delta_before_feedback: Dict[str, torch.Tensor] = {}
        self.prev_delta_before_feedback_lowrank: Dict[str, torch.Tensor] = {}
        self.total_original_volume = 0
        self.total_compressed_volume = 0
        self.step_counts: Dict[str, int] = {}
        self.eigenvalues: Dict[str, dict] = {}

    # -- helpers -------------------------------------------------------------------------------------------------
    def _compute_strided_row_similarity(self, tensor: torch.Tensor, stride: int = 1) -> Optional[float]:
        """Mean cosine similarity between rows `stride` apart of an (N, C) tensor (stats.py:58-105)."""
        assert tensor is not None and tensor.ndim == 2 and tensor.shape[0] > stride
        assert torch.isfinite(tensor).all()
        a, b = tensor[:-stride], tensor[stride:]
        keep = (torch.linalg.norm(a, dim=1) > 1e-8) & (torch.linalg.norm(b, dim=1) > 1e-8)
        assert keep.any(), "no row pair with non-zero norms"
        sims = torch.nn.functional.cosine_similarity(a[keep], b[keep], dim=1, eps=1e-8)
        return float(sims.mean())

    def _compute_eigenvalues(self, tensor: torch.Tensor) -> np.ndarray:
        """Singular values of the tensor viewed as (-1, last dim), fp32, on the CPU (stats.py:330-348)."""
        t = tensor.detach().cpu()
        if t.dim() > 2:
            t = t.reshape(-1, t.shape[-1])
        return torch.linalg.svdvals(t.float()).numpy()

    @staticmethod
    def _lowrank_view(t: torch.Tensor, rank: int = 8) -> torch.Tensor:
        """Rank-8 subspace-iteration approximation (what the reference gets from sim_compress(LOW_RANK), :187-190)."""
        from .lowrank import subspace_iter
        m = t if t.dim() == 2 else t.reshape(-1, t.shape[-1])
        u, v, _ = subspace_iter(m, min(rank, *m.shape), 2)
        return torch.matmul(u, v).reshape(t.shape).cpu()

    # -- logging (stats.py:107-328) ---------------------------------------------------------------------------------
    def log(self, key, base, delta_base, before_comp_activation, recv_activation, compressed_tensor, compress_residual):
        if compress_residual not in (0, 1, 2):
            raise ValueError("invalid residual")
        x, rec = before_comp_activation, recv_activation
        nth = self.step_counts.get(key, 0)          # how often this key was logged before: names the dump files
        self.step_counts[key] = nth + 1
        if DUMP_ACTIVATIONS:
            os.makedirs(REF_ACTIVATION_PATH, exist_ok=True)
            torch.save(x.detach().cpu(), os.path.join(REF_ACTIVATION_PATH, f"{key}_step{nth}.pt"))
        total_error = None
        if CALC_TOTAL_ERROR:                        # against activations dumped by an uncompressed run
            truth = torch.load(os.path.join(REF_ACTIVATION_PATH, f"{key}_step{nth}.pt"), map_location="cpu")
            total_error = float(torch.norm(rec.cpu() - truth))

        raw_bytes = x.numel() * x.element_size()
        wire_bytes = compressed_tensor.numel() * compressed_tensor.element_size()
        self.total_original_volume += raw_bytes
        self.total_compressed_volume += wire_bytes

        prev_x = self.prev_activations.get(key)
        dbf = dbf_lr = None                         # "delta before feedback": against the previous TRUE activation
        if prev_x is not None:
            dbf = x - prev_x.to(x.device)
            dbf_lr = self._lowrank_view(dbf)
        delta = tx_delta = ddelta = None
        if compress_residual >= 1:
            delta = x - base
            tx_delta = rec - base
            if compress_residual == 2:
                ddelta = x - base - delta_base

        row = dict.fromkeys(_FIELDS)
        row.update(error=float(torch.norm(x - rec)), total_error=total_error, activation_norm=float(torch.norm(x)),
                   delta_norm=None if delta is None else float(torch.norm(delta)),
                   delta_delta_norm=None if ddelta is None else float(torch.norm(ddelta)),
                   delta_before_feedback_norm=None if dbf is None else float(torch.norm(dbf)),
                   residual=compress_residual, original_size_bytes=raw_bytes, compressed_size_bytes=wire_bytes)
        if CALC_SIMILARITY:
            if prev_x is not None:
                row["activation_similarity"] = _cos(x, prev_x)
            if delta is not None and key in self.prev_deltas:
                row["delta_similarity"] = _cos(delta, self.prev_deltas[key])
            if CALC_MORE_SIMILARITY:
                if tx_delta is not None and key in self.prev_transmitted_deltas:
                    row["transmitted_delta_similarity"] = _cos(tx_delta, self.prev_transmitted_deltas[key])
                if dbf is not None and key in self.prev_delta_before_feedback:
                    row["delta_before_feedback_similarity"] = _cos(dbf, self.prev_delta_before_feedback[key])
                if dbf_lr is not None and key in self.prev_delta_before_feedback_lowrank:
                    row["delta_before_feedback_lowrank_similarity"] = _cos(dbf_lr, self.prev_delta_before_feedback_lowrank[key])

        layer, step_no = int(str(key).split("-")[0]), self.step_counts[key]
        if step_no in EIGENVALUES_PLOT_STEPS and layer in EIGENVALUES_PLOT_LAYERS:
            slot = self.eigenvalues.setdefault(key, {}).setdefault(step_no, {"activation": [], "delta": [], "delta_delta": []})
            slot["activation"].append(self._compute_eigenvalues(x))
            if delta is not None:
                slot["delta"].append(self._compute_eigenvalues(delta))
            if ddelta is not None:
                slot["delta_delta"].append(self._compute_eigenvalues(ddelta))

        self.stats.setdefault(key, []).append(row)
        self.prev_activations[key] = x.detach().cpu()
        for store, val in ((self.prev_deltas, delta), (self.prev_transmitted_deltas, tx_delta),
                           (self.prev_delta_before_feedback, dbf), (self.prev_delta_before_feedback_lowrank, dbf_lr)):
            if val is not None:
                store[key] = val.detach().cpu()

    # -- summaries -------------------------------------------------------------------------------------------------
    def averages(self, rows: List[dict]) -> dict:
        """Field-wise means over a list of records (None entries skipped) plus the derived ratios the summaries print."""
        out = {f: _mean([r[f] for r in rows]) for f in _FIELDS if f not in ("residual", "original_size_bytes", "compressed_size_bytes")}
        act = out["activation_norm"]
        out["relative_error"] = (out["error"] / act) if act and act > 1e-8 else float("inf")
        return out

    def summary_over_keys(self, step_range=None, key=None):
        """One block per key and residual level: error, relative error, norms and ratios, similarities (stats.py:414-506)."""
        if not self.stats:
            print("No statistics logged yet.")
            return
        for k in ([key] if key else list(self.stats)):
            if k not in self.stats:
                print(f"No data for key {k}")
                continue
            rows = self.stats[k][step_range[0]:step_range[1]] if step_range else self.stats[k]
            if not rows:
                print(f"No data for key {k} in step range {step_range}")
                continue
            for res in sorted({r["residual"] for r in rows}):
                sel = [r for r in rows if r["residual"] == res]
                a = self.averages(sel)
                print(f"[{k}] res={res} (over {len(sel)} steps):")
                if PRINT_ALL_ERROR:
                    print(f"all error: {[r['error'] for r in sel]}")
                line = f"err: {a['error']:.3f}, rel_err: {a['relative_error']:.1%}"
                if a["total_error"] is not None:
                    line += f", total_err: {a['total_error']:.3f}"
                line += f", act: {a['activation_norm']:.3f}"
                if res >= 1 and a["delta_norm"] is not None:
                    line += f", delta={a['delta_norm']:.3f}, d/a={a['delta_norm'] / a['activation_norm']:.2f}"
                    if a["delta_before_feedback_norm"] is not None:
                        line += f", dbf={a['delta_before_feedback_norm']:.3f}, dbf/a={a['delta_before_feedback_norm'] / a['activation_norm']:.2f}"
                if res >= 2 and a["delta_delta_norm"] is not None and a["delta_norm"]:
                    line += f", dd={a['delta_delta_norm']:.3f}, dd/d={a['delta_delta_norm'] / a['delta_norm']:.2f}"
                print(line)
                sims = [(lbl, a[f]) for lbl, f in (("act_sim", "activation_similarity"), ("delta_sim", "delta_similarity"),
                                                   ("tx_delta_sim", "transmitted_delta_similarity"),
                                                   ("dbf_sim", "delta_before_feedback_similarity"),
                                                   ("dbf_lr_sim", "delta_before_feedback_lowrank_similarity")) if a[f] is not None]
                if sims:
                    print("  " + ", ".join(f"{lbl}: {v:.3f}" for lbl, v in sims))

    def summary_over_steps(self, steps=None, keys=None):
        if not self.stats:
            print("No statistics logged yet.")
            return
        pool = list(self.stats) if keys is None else (keys if isinstance(keys, list) else [keys])
        n = max([len(self.stats[k]) for k in pool if k in self.stats], default=0)
        for step in (range(n) if steps is None else steps):
            if step >= n:
                print(f"Step {step} is out of range")
                continue
            print(f"=== Step {step} ===")
            for k in ([None] if keys is None else pool):
                self.summary_over_keys(step_range=(step, step + 1), key=k)

    def compression_volume(self) -> dict:
        raw, wire = self.total_original_volume, self.total_compressed_volume
        return {"raw_bytes": raw, "wire_bytes": wire, "ratio": (raw / wire) if wire else float("nan")}

    def summary_compression_volume(self):
        v = self.compression_volume()
        if v["raw_bytes"] == 0:
            print("No volume data logged yet.")
            return v
        ratio = f"{v['ratio']:.2f}x" if v["wire_bytes"] else "N/A"
        print(f"Vol: Orig {v['raw_bytes'] / 2 ** 20:.2f} MB, Comp {v['wire_bytes'] / 2 ** 20:.2f} MB, Ratio {ratio}")
        return v

    def total_averages(self) -> dict:
        """The numbers of the reference's closing summary (stats.py:528-608): the activation norm and the error are means of
        per-key means, every other quantity is a mean over all records."""
        every = [r for rows in self.stats.values() for r in rows]
        out = {
            "activation_norm": float(np.mean([np.mean([r["activation_norm"] for r in rows]) for rows in self.stats.values()])),
            "error": float(np.mean([np.mean([r["error"] for r in rows]) for rows in self.stats.values()])),
            "delta_norm": _mean([r["delta_norm"] for r in every if r["residual"] >= 1]),
            "delta_before_feedback_norm": _mean([r["delta_before_feedback_norm"] for r in every]),
            "delta_delta_norm": _mean([r["delta_delta_norm"] for r in every if r["residual"] >= 2]),
            "total_error": _mean([r["total_error"] for r in every]),
        }
        for f in ("activation_similarity", "delta_similarity", "delta_before_feedback_similarity",
                  "delta_before_feedback_lowrank_similarity", "transmitted_delta_similarity"):
            out[f] = _mean([r[f] for r in every])
        out["relative_error"] = out["error"] / out["activation_norm"] if out["activation_norm"] > 1e-8 else float("inf")
        return out

    def summary_total_avg(self):
        t = self.total_averages()
        line = f"avg activation: {t['activation_norm']:.3f}"
        for lbl, f in (("avg delta", "delta_norm"), ("avg dbf", "delta_before_feedback_norm"), ("avg delta-delta", "delta_delta_norm")):
            if t[f] is not None:
                line += f", {lbl}: {t[f]:.3f}"
        print(line)
        sims = [(lbl, t[f]) for lbl, f in (("act_sim", "activation_similarity"), ("delta_sim", "delta_similarity"),
                                           ("dbf_sim", "delta_before_feedback_similarity"),
                                           ("dbf_lr_sim", "delta_before_feedback_lowrank_similarity"),
                                           ("tx_delta_sim", "transmitted_delta_similarity")) if t[f] is not None]
        if sims:
            print("avg similarities: " + ", ".join(f"{lbl}: {v:.3f}" for lbl, v in sims))
        tail = f", avg total err: {t['total_error']:.3f}" if t["total_error"] is not None else ", [total err not logged]"
        print(f"avg comp error: {t['error']:.3f}, avg rel err: {t['relative_error']:.1%}{tail}")
        return t

    def summary_error(self) -> Dict[str, float]:
        """Per key: mean over steps of error / activation norm."""
        return {k: float(np.mean([r["error"] / r["activation_norm"] for r in rows if r["activation_norm"] > 0] or [0.0]))
                for k, rows in self.stats.items()}

    # -- dumps (same file names and dict keys as the reference's plot.py:413-560) --------------------------------------------
    def _per_step(self, field: str) -> List[Optional[float]]:
        n = max((len(rows) for rows in self.stats.values()), default=0)
        return [_mean([rows[s][field] for rows in self.stats.values() if s < len(rows)]) for s in range(n)]

    def dump_average_error_vs_steps(self, save_dir: str):
        assert self.stats, "No statistics logged. Cannot dump data."
        comp, total = self._per_step("error"), self._per_step("total_error")
        data = {"steps": list(range(len(comp))), "avg_comp_errors": comp, "avg_total_errors": total}
        os.makedirs(save_dir, exist_ok=True)
        path = os.path.join(save_dir, "average_error_vs_steps.pt")
        torch.save(data, path)
        print(f"Saved average error data to {path}")
        return data

    def dump_average_norms_and_similarity_vs_steps(self, save_dir: str):
        assert self.stats, "No statistics logged. Cannot dump data."
        act = self._per_step("activation_norm")
        data = {"steps": list(range(len(act))), "avg_act_norms": act, "avg_delta_norms": self._per_step("delta_norm"),
                "avg_act_similarities": self._per_step("activation_similarity")}
        os.makedirs(save_dir, exist_ok=True)
        path = os.path.join(save_dir, "average_norms_and_similarity_vs_steps.pt")
        torch.save(data, path)
        print(f"Saved average norms and similarity data to {path}")
        return data

    def save_eigenvalues(self, save_dir="eigenvalues"):
        """One `<key>_<step>_<kind>.pt` per captured spectrum list (stats.py:610-632)."""
        if not self.eigenvalues:
            print("No eigenvalue data available.")
            return
        os.makedirs(save_dir, exist_ok=True)
        for key, per_step in self.eigenvalues.items():
            for step, kinds in per_step.items():
                for kind, spectra in kinds.items():
                    torch.save(spectra, os.path.join(save_dir, f"{key}_{step}_{kind}.pt"))
        print(f"Saved eigenvalues to {save_dir}")

    # -- figures (stats.py:350-371, :634-647 -> compact/plot.py) ------------------------------------------------------------------
    def plot_eigenvalue_distribution(self, key=None, step=None, data_type="activation", save_dir=None, log_scale=True, top_k=None, num_bins=100):
        from .plot import plot_eigenvalue_distribution
        return plot_eigenvalue_distribution(self.eigenvalues, key, step, data_type, save_dir, log_scale, top_k, num_bins)

    def plot_eigenvalue_cumsum(self, key=None, step=None, data_type="activation", save_dir=None, log_scale=True, top_k=None):
        from .plot import plot_eigenvalue_cumsum
        return plot_eigenvalue_cumsum(self.eigenvalues, key, step, data_type, save_dir, log_scale, top_k)

    def plot_low_rank_factors(self, u, v, key, step, save_dir):
        """The U / V figure for the layers and steps the module constants select (the factors themselves are kept beside it)."""
        assert step is not None, f"Step is None for key {key}, cannot save U/V plot with step index."
        if int(str(key).split("-")[0]) not in UV_PLOT_LAYERS or step not in UV_PLOT_STEPS:
            return None
        os.makedirs(save_dir, exist_ok=True)
        torch.save({"u": u.detach().cpu(), "v": v.detach().cpu()}, os.path.join(save_dir, f"uv_{key}_{step}.pt"))
        from .plot import plot_low_rank_factors
        return plot_low_rank_factors(u, v, key, step, save_dir)

    # kept for callers of the earlier, smaller logger
    @property
    def records(self):
        return self.stats


# ---- module-level singleton API (stats.py:673-771) ---------------------------------------------------------------------
_stats: Optional[StatsLogger] = None


def stats_log() -> StatsLogger:
    global _stats
    if _stats is None:
        _stats = StatsLogger()
    return _stats


def stats_clear():
    global _stats
    _stats = None


def log(key, base, delta_base, real_activation, recv_activation, compressed_tensor, compress_residual):
    stats_log().log(key, base, delta_base, real_activation, recv_activation, compressed_tensor, compress_residual)


def stats_verbose(step_range=None, key=None, summary_keys=True):
    if _stats is None:
        print("No statistics logged.")
        return
    if summary_keys:
        _stats.summary_over_keys(step_range, key)
    _stats.summary_compression_volume()
    _stats.summary_total_avg()


def stats_verbose_steps(steps=None, keys=None):
    if _stats is None:
        print("No statistics logged.")
        return
    _stats.summary_over_steps(steps, keys)


def plot_eigenvalues(key=None, step=None, data_type="activation", save_dir=None, log_scale=True, top_k=None, cum_sum=False):
    if _stats is None:
        print("No statistics logged.")
        return
    if cum_sum:
        return _stats.plot_eigenvalue_cumsum(key, step, data_type, save_dir, log_scale, top_k)
    return _stats.plot_eigenvalue_distribution(key, step, data_type, save_dir, log_scale, top_k)


def save_eigenvalues(save_dir="eigenvalues"):
    if _stats is None:
        print("No statistics logged.")
        return
    _stats.save_eigenvalues(save_dir)


def dump_err_vs_steps(save_dir: str):
    if _stats is None:
        print("No statistics logged. Cannot dump data.")
        return
    _stats.dump_average_error_vs_steps(save_dir)


def dump_norms_sim_vs_steps(save_dir: str):
    if _stats is None:
        print("No statistics logged. Cannot dump data.")
        return
    _stats.dump_average_norms_and_similarity_vs_steps(save_dir)
