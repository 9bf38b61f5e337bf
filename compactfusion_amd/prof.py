"""Named timing scopes on HIP events (or wall clock) for the exchange path.

API-compatible with the reference's profiler (`xfuser/prof.py`: `Profiler.instance()`, `.start/.stop/.scope/
.prof_func/.elapsed_time/.get_all_elapsed_times/.sync/.reset/.enable/.disable`, `prof_summary`,
`set_torch_profiler/torch_profiler_step`) so instrumented callers keep working, but organised differently:
each scope is a small record object holding open/closed intervals, GPU intervals are (start, stop) event pairs
recorded on the stream the caller names, and a host without a GPU silently skips GPU scopes instead of raising.
"""
from __future__ import annotations

import functools
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch


@dataclass
class _ScopeRecord:
    wall_clock: bool
    open_mark: Optional[object] = None
    closed: List[Tuple[object, object]] = field(default_factory=list)
    total_ms: float = 0.0
    intervals: int = 0

    def drain(self) -> None:
        """Fold finished intervals into total_ms."""
        if not self.closed:
            return
        if self.wall_clock:
            self.total_ms += sum((b - a) * 1e3 for a, b in self.closed)
        else:
            torch.cuda.synchronize()
            self.total_ms += sum(a.elapsed_time(b) for a, b in self.closed)
        self.intervals += len(self.closed)
        self.closed.clear()


def _mark(wall_clock: bool, stream):
    if wall_clock:
        return time.time()
    ev = torch.cuda.Event(enable_timing=True)
    if stream is None:
        ev.record()
    else:
        ev.record(stream)
    return ev


class Profiler:
    _singleton: Optional["Profiler"] = None

    def __init__(self):
        self._scopes: Dict[str, _ScopeRecord] = {}
        self.enabled = True

    # -- switches ---------------------------------------------------------------------------------------------
    def enable(self):
        self.enabled = True

    def disable(self):
        self.enabled = False

    def _skip(self, cpu: bool) -> bool:
        return (not self.enabled) or (not cpu and not torch.cuda.is_available())

    # -- recording --------------------------------------------------------------------------------------------
    def start(self, name, stream=None, cpu=False):
        if self._skip(cpu):
            return
        rec = self._scopes.setdefault(name, _ScopeRecord(wall_clock=cpu))
        assert rec.open_mark is None, f"scope '{name}' started twice without a stop"
        rec.open_mark = _mark(rec.wall_clock, stream)

    def stop(self, name, stream=None, cpu=False):
        if self._skip(cpu):
            return
        rec = self._scopes.get(name)
        assert rec is not None and rec.open_mark is not None, f"scope '{name}' stopped without a start"
        rec.closed.append((rec.open_mark, _mark(rec.wall_clock, stream)))
        rec.open_mark = None

    # -- read-out ---------------------------------------------------------------------------------------------
    @property
    def events(self):
        """Names of the recorded scopes (kept for callers that iterate `profiler.events`)."""
        return self._scopes

    def elapsed_time(self, name):
        if name not in self._scopes:
            raise ValueError(f"No events recorded for '{name}'")
        rec = self._scopes[name]
        rec.drain()
        return rec.total_ms, (rec.total_ms / rec.intervals if rec.intervals else 0.0)

    def get_all_elapsed_times(self):
        totals, means = {}, {}
        for name in list(self._scopes):
            totals[name], means[name] = self.elapsed_time(name)
        return totals, means

    def sync(self):
        self.get_all_elapsed_times()

    def reset(self):
        self.sync()
        self._scopes = {}

    @staticmethod
    def instance() -> "Profiler":
        if Profiler._singleton is None:
            Profiler._singleton = Profiler()
        return Profiler._singleton

    # -- sugar ------------------------------------------------------------------------------------------------
    class _Ctx:
        def __init__(self, owner, name, stream, cpu):
            self._args = (owner, name, stream, cpu)

        def __enter__(self):
            owner, name, stream, cpu = self._args
            owner.start(name, stream, cpu=cpu)

        def __exit__(self, *exc):
            owner, name, stream, cpu = self._args
            rec = owner._scopes.get(name)
            if rec is not None and rec.open_mark is not None:
                owner.stop(name, stream, cpu=cpu)

    @staticmethod
    def scope(name, stream=None, cpu=False):
        return Profiler._Ctx(Profiler.instance(), name, stream, cpu)

    @staticmethod
    def prof_func(name, cpu=False):
        def wrap(fn):
            @functools.wraps(fn)
            def timed(*a, **kw):
                with Profiler.scope(name, cpu=cpu):
                    return fn(*a, **kw)
            return timed
        return wrap


def prof_summary(profiler: Profiler, rank=None):
    """Lines of a per-scope breakdown sorted by total time; percentages are relative to the scope named 'total'."""
    who = "N/A" if rank is None else rank
    totals, means = profiler.get_all_elapsed_times()
    whole = totals.get("total", 0.0)
    rule = "-" * 20
    out = [rule, f"Profiling Summary for Rank {who}"]
    for name in sorted(totals, key=totals.get, reverse=True):
        share = totals[name] / whole if whole > 0 else 0.0
        out.append(f"[Rank {who}] [{name}] {totals[name] / 1000:.2f}s {share:.2%} avg={means[name]:.2f}ms")
    out.append(rule)
    return out


_step_hook = None


def set_torch_profiler(profiler):
    global _step_hook
    _step_hook = profiler


def torch_profiler_step():
    if _step_hook is not None:
        _step_hook.step()
