from .state import (AllGatherCache, DummyHandle, ENTRY_VAL_LEN, HANDLES_IDX, RECV_BUF_IDX, SEND_BUF_IDX)  # noqa: F401
