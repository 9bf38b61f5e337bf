"""Golden group G14 - LOW_RANK_AWL, the reference's deprecated "attention-aware" low-rank simulate codec (BUILD container only: imports
/root/reference with COMPACT_ALLOW_DEPRECATED=1).  sim_compress(x, LOW_RANK_AWL, rank) for a K key with a per-token scale (what
compact_update_awl_scale sets, ring.py:96-103), a V key with a per-channel scale, and without a scale; torch.manual_seed before every call
pins the random start matrix (compress_lowrank.py:41).
usage: TORCHDYNAMO_DISABLE=1 COMPACT_ALLOW_DEPRECATED=1 python tests/golden/make_golden_awl.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
os.environ.setdefault("TRITON_INTERPRET", "1")
os.environ["COMPACT_ALLOW_DEPRECATED"] = "1"
m = types.ModuleType("xfuser")
m.__path__ = ["/root/reference/xfuser"]
sys.modules["xfuser"] = m
from xfuser.prof import Profiler  # noqa: E402
Profiler.instance().disable()
import xfuser.compact.main as cm  # noqa: E402
import xfuser.compact.slowpath as sp  # noqa: E402
from xfuser.compact.utils import COMPACT_COMPRESS_TYPE as T  # noqa: E402

N, C, R = 96, 256, 8
g = torch.Generator().manual_seed(77)
x = (torch.randn(N, R, generator=g) @ torch.randn(R, C, generator=g) + 0.05 * torch.randn(N, C, generator=g)).half()
tok = (0.5 + torch.rand(N, generator=g)).float()
chan = (0.5 + torch.rand(C, generator=g)).float()
out = {"x": x.view(torch.int16).numpy(), "tok": tok.numpy(), "chan": chan.numpy()}
for name, key, sk, sv in (("k_token_scale", "0-0-k", tok, None), ("v_channel_scale", "0-0-v", None, chan), ("k_no_scale", "0-0-k", None, None)):
    cm._current_cache_key = key
    sp.set_current_lowrank_scale(sk, sv)
    torch.manual_seed(4321)
    y = sp.sim_compress(x, T.LOW_RANK_AWL, rank=R)
    out[name] = y.float().numpy()
    print(name, y.dtype, float((y.float() - x.float()).norm() / x.float().norm()))
np.savez_compressed(os.path.join(HERE, "g14_lowrank_awl.npz"), **out)
