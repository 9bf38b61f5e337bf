import os
import sys

import pytest

# flag-ordered streams need hardware queues of their own (include/cfx.h: cfx_hw_queues_ok); before HIP initialises.  4, not more: this
# process stays alive (with every CU-masked stream its tests created - each a hardware queue of its own) while tests start rank
# processes that share the GPU, and a device whose hardware queues are oversubscribed time-slices them: a polling kernel of one process
# then waits out the quantum of another (seen as a gate time-out in the two-process bench test when this was 8)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
