"""ctypes access to the C oracle (oracle/cfx_oracle.c).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libcfx_oracle.so")
LIB_F16C = os.path.join(HERE, "libcfx_oracle_f16c.so")
CODEC_ID = {"binary": 1, "int2": 2, "int4": 3, "int8": 4, "topk": 5}
_lib = None


def build(force=False):
    src = os.path.join(HERE, "cfx_oracle.c")
    stale = any(not os.path.exists(p) or os.path.getmtime(src) > os.path.getmtime(p) for p in (LIB, LIB_F16C))
    if force or stale:
        subprocess.run(["make", "-C", HERE, "-s", "-B"], check=True)
    return LIB


def cpu_has_f16c() -> bool:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = set(line.split(":", 1)[1].split())
                return "f16c" in flags and "avx2" in flags
    except OSError:
        pass
    return False


def load(variant=None):
    """variant: None = hardware fp16 conversions when the CPU has them, "generic" / "f16c" to force one (tests)."""
    global _lib
    if variant is not None:
        build()
        return _bind(ctypes.CDLL(LIB_F16C if variant == "f16c" else LIB))
    if _lib is None:
        if not os.path.exists(LIB) or not os.path.exists(LIB_F16C):
            build()
        _lib = _bind(ctypes.CDLL(LIB_F16C if cpu_has_f16c() else LIB))
    return _lib


def _bind(L):
    if True:
        L.oracle_packet_bytes.restype = ctypes.c_size_t
        L.oracle_packet_bytes.argtypes = [ctypes.c_int] * 4
        L.oracle_compress.restype = ctypes.c_int
        L.oracle_compress.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4
        L.oracle_decompress.restype = ctypes.c_int
        L.oracle_decompress.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_set_num_threads.argtypes = [ctypes.c_int]
        L.oracle_uses_f16c.restype = ctypes.c_int
        L.oracle_init()
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def compress(codec, x, base, N, C, param=0, update=True, ef=True, packet=None, new_base=None):
    """x/base: uint16 or float16 arrays (N,C).  Returns (packet uint16 words, new_base uint16 | None)."""
    L = load()
    cid = CODEC_ID[codec] if isinstance(codec, str) else codec
    x = np.ascontiguousarray(x).view(np.uint16)
    base = None if base is None else np.ascontiguousarray(base).view(np.uint16)
    nbytes = L.oracle_packet_bytes(cid, N, C, param)
    if packet is None:
        packet = np.zeros(nbytes // 2, dtype=np.uint16)
    if update and new_base is None:
        new_base = np.empty((N, C), dtype=np.uint16)
    flags = (1 if update else 0) | (0 if ef else 2)
    rc = L.oracle_compress(cid, _p(x), _p(base), _p(new_base) if update else None, _p(packet), N, C, param, flags)
    assert rc == 0
    return packet, (new_base if update else None)


def decompress(codec, packet, base, N, C, param=0, out=None):
    L = load()
    cid = CODEC_ID[codec] if isinstance(codec, str) else codec
    packet = np.ascontiguousarray(packet).view(np.uint16)
    base = None if base is None else np.ascontiguousarray(base).view(np.uint16)
    if out is None:
        out = np.empty((N, C), dtype=np.uint16)
    rc = L.oracle_decompress(cid, _p(packet), _p(base), _p(out), N, C, param)
    assert rc == 0
    return out


def num_threads():
    return load().oracle_num_threads()


def set_num_threads(n: int):
    load().oracle_set_num_threads(int(n))
