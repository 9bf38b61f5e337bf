"""Access to the committed golden vectors (tests/golden/*.npz + MANIFEST.json).

Big arrays of the larger shapes are not stored; they are pinned by sha256 and the inputs are regenerated
with the reference test recipe (tests/compact/compress_fastpath_test.py:57-58) from the seed."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_man = None
_files = {}


def manifest():
    global _man
    if _man is None:
        _man = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    return _man


def npz(fn):
    if fn not in _files:
        _files[fn] = np.load(os.path.join(HERE, fn))
    return _files[fn]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def entry(fn, key):
    return manifest()[fn][key]


def stored(fn, key):
    return manifest()[fn][key]["stored"]


def get(fn, key):
    return npz(fn)[key]


def check(fn, key, arr, what=""):
    """arr must equal the golden array bit for bit (compared by sha256, and elementwise when stored)."""
    e = entry(fn, key)
    arr = np.ascontiguousarray(arr)
    assert list(arr.shape) == e["shape"], (what, key, arr.shape, e["shape"])
    if e["stored"]:
        g = get(fn, key)
        if arr.dtype != g.dtype:
            arr = arr.view(g.dtype)
        bad = int((arr != g).sum())
        assert bad == 0, f"{what} {key}: {bad}/{arr.size} elements differ from the golden vector"
    else:
        assert sha(arr.view(np.dtype(e["dtype"])) if arr.dtype != np.dtype(e["dtype"]) else arr) == e["sha256"], f"{what} {key}: sha256 mismatch"


def gen_inputs(seed, N, C):
    """Regenerate (x, base) as uint16 bit patterns with the reference recipe."""
    import torch
    torch.manual_seed(seed)
    x = torch.randn((N, C), dtype=torch.half).contiguous()
    base = (torch.randn_like(x) * 0.1).contiguous()
    return x.view(torch.int16).numpy().view(np.uint16).copy(), base.view(torch.int16).numpy().view(np.uint16).copy()


def inputs(fn, tag, seed, N, C):
    if stored(fn, f"{tag}/x"):
        return get(fn, f"{tag}/x"), get(fn, f"{tag}/base")
    x, b = gen_inputs(seed, N, C)
    assert sha(x) == entry(fn, f"{tag}/x")["sha256"], "torch CPU RNG produced different inputs than when the golden vectors were captured"
    assert sha(b) == entry(fn, f"{tag}/base")["sha256"]
    return x, b


def rel_err(a16, b16):
    a = a16.view(np.float16).astype(np.float64)
    b = b16.view(np.float16).astype(np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def ulp_diff_count(a_bits, b_bits):
    a = np.asarray(a_bits).view(np.uint16).astype(np.int32).reshape(-1)
    b = np.asarray(b_bits).view(np.uint16).astype(np.int32).reshape(-1)
    d = np.abs(a - b)
    return int((d != 0).sum()), int(d.max()) if d.size else 0
