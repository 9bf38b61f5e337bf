"""compactfusion_amd - MI355X-native residual-compressed activation exchange (CompactFusion's hot path).

Python host code mirroring the reference's `xfuser.compact` plugin API on top of hand-written gfx950 HIP kernels
behind a C-ABI (include/cfx.h, libcfx.so).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
