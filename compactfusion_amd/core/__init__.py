"""Hook surface of the exchange path: sequence-parallel process groups and the long-context attention layer that binds
`compact_fwd` (mirrors the pieces of `xfuser.core.distributed` / `xfuser.core.long_ctx_attention` the path touches)."""
from .distributed import get_sp_group, init_sequence_parallel, destroy_sequence_parallel  # noqa: F401
from .long_ctx_attention import xFuserLongContextAttention  # noqa: F401
