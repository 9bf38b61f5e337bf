"""Per-kernel register / scratch / LDS use of every HIP source of libcfx.so, from the compiler's own remarks.

`python tools/resource_usage.py` prints one line per kernel (demangled name, VGPRs, AGPRs, SGPRs, scratch bytes per lane, occupancy,
LDS bytes); tests/test_resource_usage.py fails the CPU suite when a kernel uses scratch. hipcc cross-compiles: no GPU needed.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

_FIELDS = (("name", r"Function Name: (\S+)"), ("sgpr", r" SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r" AGPRs: (\d+)"),
           ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
           ("lds", r"LDS Size \[bytes/block\]: (\d+)"))


def _one(src: str) -> list[dict]:
    from compactfusion_amd import build as B
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared",)]
    cmd = [B.hipcc_path()] + flags + ["-c", f"-I{B.INC}", f"-I{os.path.join(B.PKG_DIR, 'csrc')}", src, "-o", os.devnull,
                                      "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on " + src + ":\n" + r.stderr[-4000:])
    out, cur = [], None
    for line in r.stderr.splitlines():
        for key, pat in _FIELDS:
            m = re.search(pat, line)
            if not m:
                continue
            if key == "name":
                cur = {"name": m.group(1), "file": os.path.basename(src)}
                out.append(cur)
            elif cur is not None:
                cur[key] = int(m.group(1))
    return out


def demangle(names: list[str]) -> list[str]:
    import shutil
    tool = shutil.which("llvm-cxxfilt") or shutil.which("c++filt")
    if not tool:
        return names
    r = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True)
    got = r.stdout.splitlines()
    return got if len(got) == len(names) else names


def collect() -> list[dict]:
    from compactfusion_amd import build as B
    with ThreadPoolExecutor(max_workers=5) as ex:
        rows = [k for ks in ex.map(_one, B.SRC) for k in ks]
    for k, d in zip(rows, demangle([k["name"] for k in rows])):
        k["demangled"] = re.sub(r"^void ", "", re.sub(r"\(.*$", "", d))
    return rows


if __name__ == "__main__":
    rows = collect()
    for k in sorted(rows, key=lambda k: (-k.get("scratch", 0), -k.get("vgpr", 0))):
        print(f'{k.get("scratch", 0):5d} B/lane  v{k.get("vgpr", 0):3d} a{k.get("agpr", 0):3d} s{k.get("sgpr", 0):3d}  occ {k.get("occupancy", 0)}  '
              f'lds {k.get("lds", 0):6d}  {k["file"]}: {k["demangled"]}')
