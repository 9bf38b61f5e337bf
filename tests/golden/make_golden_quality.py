"""Golden group G12 - a quality band without model weights (run in the BUILD container only: imports /root/reference).

BASELINE.json's metric has an "images at matched quality" half; FLUX weights are not available here, so what is pinned instead is
the error-vs-step behaviour of the REFERENCE's own compact_compress / compact_decompress over a 28-step denoise-like drift, for
its shipped presets (examples/configs.py:39-98): BINARY fastpath, INT2 fastpath, LOW_RANK r=8, LOW_RANK_Q r=32, residual 1 +
error feedback, 1 WARMUP step.  Per step: relative reconstruction error of K and V at the receiver, and the PSNR of the attention
output computed from the reconstructed K,V against the attention output from the true K,V.  tests/test_gpu_quality.py runs the
HIP path on the same inputs and must stay within 1e-3 relative of this trace.

Inputs are regenerated from seeds in the test (sha256-pinned here); the low-rank start matrices are drawn as the reference draws
them (torch.randn(C, r) on the CPU generator, compress_lowrank.py:41) after `torch.manual_seed(SEED_Q + 2 * step + kv)`.

usage: TORCHDYNAMO_DISABLE=1 python tests/golden/make_golden_quality.py [--codecs binary,int2,lr8,lrq32] [--steps 28]
"""
import argparse
import hashlib
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
N, HEADS, HD = 128, 24, 128
C = HEADS * HD
SEED_X, SEED_Q = 4242, 900000


def drift(seed, T):
    """K or V of one layer over T denoise steps: x_0 ~ N(0,1), x_t = x_{t-1} + 0.1 N(0,1) (BASELINE.md section 2 recipe)."""
    g = torch.Generator().manual_seed(seed)
    cur = torch.randn(N, C, generator=g).half()
    out = []
    for _ in range(T):
        out.append(cur.contiguous())
        cur = (cur.float() + 0.1 * torch.randn(N, C, generator=g)).half()
    return out


def query(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, N, HEADS, HD, generator=g).half()


def attention(q, k, v):
    """fp32 softmax attention, (1,N,H,D) layout."""
    qt, kt, vt = (t.view(1, N, HEADS, HD).transpose(1, 2).float() for t in (q, k, v))
    s = torch.matmul(qt, kt.transpose(-1, -2)) * HD ** -0.5
    return torch.matmul(torch.softmax(s, dim=-1), vt)


def metrics(q, k_true, v_true, k_rec, v_rec):
    rk = float((k_rec.float() - k_true.float()).norm() / k_true.float().norm())
    rv = float((v_rec.float() - v_true.float()).norm() / v_true.float().norm())
    ref, got = attention(q, k_true, v_true), attention(q, k_rec, v_rec)
    mse = float(((ref - got) ** 2).mean())
    psnr = float(20 * np.log10(float(ref.abs().max())) - 10 * np.log10(max(mse, 1e-30)))
    return rk, rv, psnr


PRESETS = {   # name: (COMPACT_COMPRESS_TYPE member, CompactConfig kwargs)   examples/configs.py:39-98
    "binary": ("BINARY", dict(comp_rank=-1, fastpath=True)),
    "int2": ("INT2", dict(comp_rank=-1, fastpath=True)),
    "lr8": ("LOW_RANK", dict(comp_rank=8, fastpath=False)),
    "lrq32": ("LOW_RANK_Q", dict(comp_rank=32, fastpath=False)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--codecs", default="binary,int2,lr8,lrq32")
    ap.add_argument("--steps", type=int, default=28)
    args = ap.parse_args()
    os.environ.setdefault("TRITON_INTERPRET", "1")
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    m = types.ModuleType("xfuser")
    m.__path__ = [os.path.join(REF, "xfuser")]
    sys.modules["xfuser"] = m
    from xfuser.prof import Profiler
    Profiler.instance().disable()
    from xfuser.collector import collector
    collector.init(collector.Collector("/tmp/cfx_golden_collector", enabled=False))
    import xfuser.compact.main as cm
    from xfuser.compact.utils import CompactConfig, COMPACT_COMPRESS_TYPE as T

    Tn = args.steps
    ks, vs, q = drift(SEED_X, Tn), drift(SEED_X + 1, Tn), query(SEED_X + 2)
    out_path = os.path.join(HERE, "g12_quality.npz")
    res = dict(np.load(out_path)) if os.path.exists(out_path) else {}
    sha = lambda t: hashlib.sha256(t.contiguous().view(torch.int16).numpy().tobytes()).hexdigest()   # noqa: E731
    meta = {"N": N, "C": C, "steps": Tn, "seed_x": SEED_X, "seed_q": SEED_Q,
            "sha_k_last": sha(ks[-1]), "sha_v_last": sha(vs[-1]), "sha_q": sha(q)}
    for name in args.codecs.split(","):
        tname, kw = PRESETS[name]
        ctype = T[tname]
        cm.compact_init(CompactConfig(enabled=True, compress_func=lambda l, s: None, residual=1, ef=True, simulate=False,
                                      log_stats=False, **kw))
        rows = []
        t0 = time.time()
        for t in range(Tn):
            typ = T.WARMUP if t == 0 else ctype
            rec = []
            for kv, x in enumerate((ks[t], vs[t])):
                torch.manual_seed(SEED_Q + 2 * t + kv)           # the low-rank start matrix is the next torch.randn(C, r)
                skey, rkey = f"0-0-{'kv'[kv]}", f"0-1-{'kv'[kv]}"
                pkt = cm.compact_compress(skey, x.view(1, N, HEADS, HD), typ, update_cache=True)
                r = cm.compact_decompress(rkey, pkt.clone(), typ, (1, N, HEADS, HD), update_cache=True)
                rec.append(r.reshape(N, C).clone())
                s_state, r_state = cm.compact_cache().get_base(skey), cm.compact_cache().get_base(rkey)
                assert torch.equal(s_state, r_state), "reference sender / receiver states diverged"
            rows.append(metrics(q, ks[t], vs[t], rec[0], rec[1]))
            print(f"G12 {name} step {t}: rel_k {rows[-1][0]:.5f} rel_v {rows[-1][1]:.5f} psnr {rows[-1][2]:.3f} dB  ({time.time() - t0:.0f} s)", flush=True)
        res[f"{name}/trace"] = np.array(rows, dtype=np.float64)          # [step][rel_err_k, rel_err_v, attention PSNR dB]
        res[f"{name}/final_state_k_sha"] = np.frombuffer(bytes.fromhex(sha(cm.compact_cache().get_base("0-1-k"))), dtype=np.uint8)
        np.savez_compressed(out_path, **res)
    with open(os.path.join(HERE, "g12_quality_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
