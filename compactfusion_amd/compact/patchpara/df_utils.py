from .state import PatchConfig  # noqa: F401  (import path of the reference: xfuser.compact.patchpara.df_utils)
