"""Golden group G1b - the 1-bit codec with rank-K scales (run in the BUILD container only: imports /root/reference).

The reference's `binary_quant_fastpath(x, base, rank, True)` / `binary_dequant_fastpath` for rank in {1, 4} - the ranks its own test
parametrises (tests/compact/compress_fastpath_test.py:45-101) - on the G1 input recipe.  The scales are `subspace_iter(|x - base|,
rank, 2)` whose start matrix is the first `torch.randn(C, rank)` after `torch.manual_seed(SEED_Q)` (compress_lowrank.py:41); it is
stored, so the oracle and the HIP path can start from the same span.  Deprecated branch in the reference (main.py:188-189).

`wide` writes g1c_binary_rank_wide.npz: ranks 16 and 32 - the reference's Triton kernels take any power of two (fastpath.py:91
tl.arange(0, RANK)); 32 is this repo's factor-chain limit - on one shape each, so the file stays small.

usage: TORCHDYNAMO_DISABLE=1 TRITON_INTERPRET=1 python tests/golden/make_golden_rank.py [wide]
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SEED_Q = 777


def gen_inputs(seed, N, C):
    """The reference test's recipe (compress_fastpath_test.py:57-58)."""
    torch.manual_seed(seed)
    x = torch.randn((N, C), dtype=torch.half).contiguous()
    base = (torch.randn_like(x) * 0.1).contiguous()
    return x, base


def np16(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16) if t.dtype == torch.half else t.detach().cpu().numpy()


def main():
    os.environ.setdefault("TRITON_INTERPRET", "1")
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    m = types.ModuleType("xfuser")
    m.__path__ = [os.path.join(REF, "xfuser")]
    sys.modules["xfuser"] = m
    from xfuser.prof import Profiler
    Profiler.instance().disable()
    from xfuser.compact.fastpath import binary_dequant_fastpath, binary_quant_fastpath
    res = {}
    wide = len(sys.argv) > 1 and sys.argv[1] == "wide"
    cases = [((64, 256), (42,), (16, 32)), ((128, 1152), (43,), (16,))] if wide else [((64, 256), (42, 43), (1, 4)), ((256, 1152), (42, 43), (1, 4))]
    for (N, C), seeds, ranks in cases:
        for seed in seeds:
            x, base = gen_inputs(seed, N, C)
            for rank in ranks:
                tag = f"{N}x{C}_s{seed}/r{rank}"
                torch.manual_seed(SEED_Q + rank)
                q0 = torch.randn(C, rank, dtype=torch.float)              # what subspace_iter is about to draw
                torch.manual_seed(SEED_Q + rank)
                packed, u, v, nb = binary_quant_fastpath(x, base, rank, True)
                recon = binary_dequant_fastpath(packed, u, v, base)
                assert torch.equal(recon, nb), "reference sender / receiver arithmetic differ"
                res[f"{tag}/q0"] = q0.numpy()
                res[f"{tag}/packed"] = packed.numpy()
                res[f"{tag}/u"] = np16(u)
                res[f"{tag}/v"] = np16(v)
                res[f"{tag}/new_base"] = np16(nb)
                print("G1b", tag, "scale rel. to |d| mean:", float((u.float() @ v.float().t()).mean() / (x - base).abs().float().mean()), flush=True)
    np.savez_compressed(os.path.join(HERE, "g1c_binary_rank_wide.npz" if wide else "g1b_binary_rank.npz"), **res)


if __name__ == "__main__":
    main()
