"""Build libcfx.so (the HIP kernels + C-ABI) in-tree with hipcc for gfx950.

The shared object is git-ignored but travels to the GPU box with the repo snapshot.
`python -m compactfusion_amd.build` or `__graft_entry__.build()` calls this.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
SRC = [os.path.join(PKG_DIR, "csrc", f) for f in ("cfx_api.hip", "cfx_absmean.hip", "cfx_minmax.hip", "cfx_topk.hip", "cfx_plan.hip", "cfx_lowrank.hip", "cfx_lrgram.hip", "cfx_lrslab.hip")]
INC = os.path.join(REPO, "include")
LIB = os.path.join(PKG_DIR, "libcfx.so")
LIB_DEV = os.path.join(PKG_DIR, "libcfx_dev.so")      # the same sources with -DCFX_DEV_PROBES (include/cfx_dev.h): tools and one test only
ARCH = "gfx950"

HIPCC_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17",
    "-ffp-contract=off",          # one rounding per reference op: no fma contraction of u*v + base
    "-fPIC", "-shared",
]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libcfx.so")


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = SRC + [os.path.join(INC, "cfx.h"), os.path.join(INC, "cfx_dev.h"), os.path.join(PKG_DIR, "csrc", "cfx_internal.h"),
                  os.path.join(PKG_DIR, "csrc", "cfx_lr.h"), os.path.join(PKG_DIR, "csrc", "cfx_device.h"), os.path.join(PKG_DIR, "csrc", "cfx_host.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False, dev_probes: bool = False) -> str:
    """The product library libcfx.so - or, dev_probes, the DEVELOPER library libcfx_dev.so: the same sources with -DCFX_DEV_PROBES
    (per-workgroup phase stamps in the kernels, the launch-tag test hook, early exits of the compress kernel: include/cfx_dev.h).  The
    product library has none of that compiled in and exports none of those symbols."""
    lib = LIB_DEV if dev_probes else LIB
    if not force and not needs_build(lib):
        return lib
    cmd = [hipcc_path()] + HIPCC_FLAGS + (["-DCFX_DEV_PROBES"] if dev_probes else []) + [f"-I{INC}", f"-I{os.path.join(PKG_DIR, 'csrc')}"] + SRC + ["-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(lib + ".tmp", lib)
    return lib


def build_both(force: bool = False) -> tuple:
    """Product and developer library side by side (two hipcc processes)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as ex:
        a, b = ex.submit(build_lib, force, False, False), ex.submit(build_lib, force, False, True)
        return a.result(), b.result()


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv or "--dev-probes" in sys.argv, verbose=True, dev_probes="--dev-probes" in sys.argv))
