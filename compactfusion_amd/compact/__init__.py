"""`compactfusion_amd.compact` - drop-in for the reference's `xfuser.compact` package (same module and function names)."""
from .utils import COMPACT_COMPRESS_TYPE, CompactCache, CompactConfig  # noqa: F401
from .patchpara.df_utils import PatchConfig  # noqa: F401
from .main import (  # noqa: F401
    allgather_cache, compact_all_gather, compact_cache, compact_compress, compact_config, compact_decompress,
    compact_get_step, compact_hello, compact_init, compact_reset, compact_set_step,
)
