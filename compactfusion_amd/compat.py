"""Drop-in aliasing: make `import xfuser.compact...` resolve to this package.

The reference's compressor is a Python package with module-global state and no FFI: the call sites in xDiT /
CompactFusion import `xfuser.compact.main`, `xfuser.compact.ring`, `xfuser.prof`, ... by name (attn_layer.py:59-64, pipeline_flux.py:447-450,
examples/*.py).  `install_xfuser_alias()` registers this package's modules under those names BEFORE anything imports them, so a CompactFusion
checkout runs on libcfx without a source change (INTEGRATION.md section 1 shows the same lines as a patch to `xfuser/__init__.py`).
"""
from __future__ import annotations

import importlib
import sys

import pkgutil


def _compact_modules() -> list:
    """Every submodule of compactfusion_amd.compact (walked, not listed: a module that is reachable only through the aliased parent's
    __path__ would be imported a SECOND time under the xfuser.* name - duplicate classes, duplicate module-level state)."""
    pkg = importlib.import_module("compactfusion_amd.compact")
    return sorted(m.name[len("compactfusion_amd.compact."):] for m in pkgutil.walk_packages(pkg.__path__, "compactfusion_amd.compact.")
                  if not m.ispkg)


def install_xfuser_alias(overwrite: bool = False) -> list:
    """Register compactfusion_amd.compact.* as xfuser.compact.*, compactfusion_amd.prof as xfuser.prof and the collector as
    xfuser.collector.collector.  Returns the names it registered.  Raises if a real `xfuser.compact` module is already imported
    (call it first, or pass overwrite=True on purpose)."""
    if not overwrite and "xfuser.compact.main" in sys.modules and not sys.modules["xfuser.compact.main"].__name__.startswith("compactfusion_amd"):
        raise RuntimeError("xfuser.compact is already imported: call install_xfuser_alias() before anything imports xfuser.compact")
    done = []

    def reg(alias: str, target: str) -> None:
        sys.modules[alias] = importlib.import_module(target)
        done.append(alias)
    reg("xfuser.compact", "compactfusion_amd.compact")
    reg("xfuser.compact.patchpara", "compactfusion_amd.compact.patchpara")
    for m in _compact_modules():
        reg(f"xfuser.compact.{m}", f"compactfusion_amd.compact.{m}")
    reg("xfuser.prof", "compactfusion_amd.prof")
    reg("xfuser.collector.collector", "compactfusion_amd.collector.collector")
    return done
