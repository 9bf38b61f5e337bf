"""ctypes access to the C oracle (oracle/cfx_oracle.c).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libcfx_oracle.so")
CODEC_ID = {"binary": 1, "int2": 2, "int4": 3, "int8": 4, "topk": 5}
_lib = None


def build(force=False):
    src = os.path.join(HERE, "cfx_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(src) > os.path.getmtime(LIB):
        subprocess.run(["make", "-C", HERE, "-s", "-B"], check=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = ctypes.CDLL(LIB)
        L.oracle_packet_bytes.restype = ctypes.c_size_t
        L.oracle_packet_bytes.argtypes = [ctypes.c_int] * 4
        L.oracle_compress.restype = ctypes.c_int
        L.oracle_compress.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4
        L.oracle_decompress.restype = ctypes.c_int
        L.oracle_decompress.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_init()
        _lib = L
    return _lib


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
