"""compactfusion_amd - MI355X-native residual-compressed activation exchange (CompactFusion's hot path).

Python host code mirroring the reference's `xfuser.compact` plugin API on top of hand-written gfx950 HIP kernels
behind a C-ABI (include/cfx.h, libcfx.so).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"


from .config import configure  # noqa: E402,F401


def _default_hw_queues() -> None:
    """Flag-ordered streams (the exchange lane, the collective form of the one-launch exchange layer) need hardware queues of their own:
    HIP only gives every stream one when GPU_MAX_HW_QUEUES is set before the runtime initialises (include/cfx.h: cfx_hw_queues_ok).
    Importing the package before the first CUDA / HIP call exports a default for the process - said ONCE on stderr, because it reaches
    every stream of the process, the model's included; `compactfusion_amd.configure(hw_queues=n)` or the variable itself choose another
    value; CFX_NO_DEFAULT_HW_QUEUES=1 opts out - libcfx then runs those ops in stream order on one stream
    (same results, two launches per layer).  If HIP is already up the variable is left alone."""
    import os
    import sys
    if "GPU_MAX_HW_QUEUES" in os.environ or os.environ.get("CFX_NO_DEFAULT_HW_QUEUES") == "1":
        return
    torch = sys.modules.get("torch")
    try:
        if torch is not None and torch.cuda.is_initialized():
            return
    except Exception:  # noqa: BLE001
        return
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    if os.environ.get("RANK", "0") != "0" or os.environ.get("CFX_QUIET") == "1":
        return                                 # (said once per job - by rank 0 - not once per rank)
    print("compactfusion_amd: GPU_MAX_HW_QUEUES=8 exported for this process (flag-ordered streams need hardware queues of their own; "
          "set the variable, call compactfusion_amd.configure(hw_queues=n) first, or CFX_NO_DEFAULT_HW_QUEUES=1 to opt out)", file=sys.stderr)


_default_hw_queues()
