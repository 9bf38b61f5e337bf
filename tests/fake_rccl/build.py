"""Build tests/fake_rccl/libcfx_fake_rccl.so (test infrastructure: an in-process stand-in for the RCCL entry points
libcfx.so resolves at run time).  `python tests/fake_rccl/build.py`; __graft_entry__.build() calls build()."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl.cpp")
LIB = os.path.join(HERE, "libcfx_fake_rccl.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-x", "hip", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", SRC, "-o", LIB + ".tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the fake RCCL failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
