"""The reference's shipped configurations by name (`/root/reference/examples/configs.py:6-200`, `get_config(model_name, method)`): what
its example scripts and benchmark drivers (`examples/flux_example.py`, `run_BWTest.sh`) select with `--compact_method`.  Same names, same fields.

One table instead of a function per preset.  A codec preset starts with WARMUP steps (uncompressed exchange that seeds the states): 2 for
CogVideoX, 1 otherwise (`configs.py:9`).  `pipe`, `ring`, `ulysses` run the baseline parallelism with compaction disabled; `patch` / `df` are the
uncompressed patch-gather forward (synchronous / DistriFusion-style displaced); `int2patch` compresses the patch gather.  The reference's `patch`
branch passes an argument its `_patch_config()` does not take (`configs.py:32` vs `:153`) and so raises there; here it returns the configuration
that function builds.
"""
from .utils import CompactConfig, COMPACT_COMPRESS_TYPE as T
from .patchpara.state import PatchConfig

MODELS = ("Flux", "Pixart-alpha", "CogVideoX")                 # configs.py:8

# method -> (codec, comp_rank, fastpath)                        configs.py:39-107, 193-203
_CODECS = {
    "binary": (T.BINARY, -1, True),
    "int2": (T.INT2, -1, True),
    "lowrank8": (T.LOW_RANK, 8, False),
    "lowrank12": (T.LOW_RANK, 12, False),
    "lowrank16": (T.LOW_RANK, 16, False),                        # defined upstream, commented out of its dispatcher (:22-23)
    "lowrankq32": (T.LOW_RANK_Q, 32, False),
}
# method -> PatchConfig arguments (use_compact, async_comm, async_warmup = the warm-up steps unless given)     configs.py:110-165
_PATCH = {"int2patch": (True, False, None), "df": (False, True, None), "patch": (False, False, 0)}
_DISABLED = ("pipe", "ring", "ulysses")                         # configs.py:27-33, 145-151

METHODS = tuple(m for m in _CODECS if m != "lowrank16") + tuple(_PATCH) + _DISABLED


def warmup_steps_for(model_name: str) -> int:
    return 2 if model_name == "CogVideoX" else 1


def _schedule(codec, warmup):
    """compress_func(layer_idx, step): WARMUP for the first `warmup` steps, the codec afterwards (configs.py:42 and siblings)."""
    return lambda layer_idx, step: codec if step >= warmup else T.WARMUP


def get_config(model_name: str, method: str) -> CompactConfig:
    if model_name not in MODELS:
        raise ValueError(f"Model {model_name} not supported")
    warmup = warmup_steps_for(model_name)
    if method in _DISABLED:
        return CompactConfig(enabled=False, compress_func=None, simulate=False, log_stats=False)
    if method in _CODECS:
        codec, rank, fast = _CODECS[method]
        return CompactConfig(enabled=True, compress_func=_schedule(codec, warmup), comp_rank=rank, residual=1, ef=True, simulate=False,
                             log_stats=False, fastpath=fast)
    if method in _PATCH:
        use_compact, async_comm, async_warmup = _PATCH[method]
        patch = PatchConfig(use_compact=use_compact, async_comm=async_comm, async_warmup=warmup if async_warmup is None else async_warmup)
        if use_compact:                                          # int2patch: the 2-bit codec on the gathered K,V (configs.py:110-127)
            return CompactConfig(enabled=True, override_with_patch_gather_fwd=True, patch_gather_fwd_config=patch,
                                 compress_func=_schedule(T.INT2, warmup), comp_rank=-1, residual=1, ef=True, simulate=False, log_stats=False,
                                 fastpath=True)
        return CompactConfig(enabled=True, override_with_patch_gather_fwd=True, patch_gather_fwd_config=patch, compress_func=None, ef=False,
                             simulate=False, log_stats=False, fastpath=False)
    raise ValueError(f"compact method {method!r} not known (one of {', '.join(METHODS)})")
