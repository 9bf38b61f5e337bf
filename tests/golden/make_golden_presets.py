"""Golden group G15 - the reference's named configurations (BUILD container only: imports /root/reference/examples/configs.py).  For every
model x method its dispatcher accepts: the CompactConfig's fields, the PatchConfig's, and what compress_func answers for layers 0 / 5 at steps
0 .. 3.  Data only (names and numbers).  A method whose construction raises upstream is recorded with the exception's type.
usage: TORCHDYNAMO_DISABLE=1 python tests/golden/make_golden_presets.py"""
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
os.environ.setdefault("TRITON_INTERPRET", "1")
m = types.ModuleType("xfuser")
m.__path__ = ["/root/reference/xfuser"]
sys.modules["xfuser"] = m
spec = importlib.util.spec_from_file_location("ref_configs", "/root/reference/examples/configs.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

FIELDS = ("enabled", "override_with_patch_gather_fwd", "comp_rank", "compress_residual", "error_feedback", "simulate_compress", "log_compress_stats",
          "fastpath", "quantized_cache", "sparse_ratio", "delta_decay_factor", "check_cache_consistency")
out = {}
for model in ("Flux", "Pixart-alpha", "CogVideoX"):
    for method in ("binary", "int2", "lowrank12", "lowrank8", "lowrankq32", "df", "pipe", "ring", "patch", "ulysses", "int2patch"):
        try:
            c = ref.get_config(model, method)
        except Exception as e:                                   # noqa: BLE001
            out[f"{model}/{method}"] = {"raises": type(e).__name__}
            continue
        rec = {f: getattr(c, f) for f in FIELDS if hasattr(c, f)}
        rec["fields_present"] = sorted(k for k in vars(c) if not k.startswith("_"))
        p = getattr(c, "patch_gather_fwd_config", None)
        rec["patch"] = None if p is None else {"use_compact": p.use_compact, "async_comm": p.async_comm, "async_warmup": p.async_warmup}
        f = getattr(c, "compress_func", None)
        rec["schedule"] = None if f is None else [[f(layer, step).name for step in range(4)] for layer in (0, 5)]
        out[f"{model}/{method}"] = rec
json.dump(out, open(os.path.join(HERE, "g15_presets.json"), "w"), indent=1, sort_keys=True, default=str)
print(len(out), "entries;", sum("raises" in v for v in out.values()), "raise upstream")
