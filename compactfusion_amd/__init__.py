"""compactfusion_amd - MI355X-native residual-compressed activation exchange (CompactFusion's hot path).

Python host code mirroring the reference's `xfuser.compact` plugin API on top of hand-written gfx950 HIP kernels
behind a C-ABI (include/cfx.h, libcfx.so).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"


def _default_hw_queues() -> None:
    """Flag-ordered streams (the one-launch exchange layer, the exchange lane) need hardware queues of their own: HIP only gives every
    stream one when GPU_MAX_HW_QUEUES is set before the runtime initialises (include/cfx.h: cfx_hw_queues_ok).
    Importing the package before the first CUDA / HIP call sets a default; if HIP is already up the variable is left alone - libcfx
    then sees it unset and runs those ops in stream order on one stream instead (same results, two launches per layer)."""
    import os
    import sys
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return
    torch = sys.modules.get("torch")
    try:
        if torch is not None and torch.cuda.is_initialized():
            return
    except Exception:  # noqa: BLE001
        return
    os.environ["GPU_MAX_HW_QUEUES"] = "8"


_default_hw_queues()
