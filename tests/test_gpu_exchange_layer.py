"""cfx_plan_add_exchange_layer on the GPU (-m gpu): the cases live in tests/xlayer_cases.py and each runs in a process of its own.

Why a process each: the op orders two streams by flag words, i.e. kernels that POLL.  HIP multiplexes streams over a small pool of
hardware queues, and with GPU_MAX_HW_QUEUES unset a stream created after RCCL has initialised in the process can be time-sliced against -
or never scheduled beside - the polling kernel's queue ("Hardware-queue note"): inside the full suite, behind a
test that brings RCCL up, a gate of these cases then sits out its 5 s timeout.  A deployment sets GPU_MAX_HW_QUEUES before HIP starts
(bench.py does); so do these processes, instead of changing the environment of every other test of the suite."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

CASES = ['test_flag_relay_without_a_communicator',
         'test_collective_kernel_on_the_exchange_stream',
         'test_collective_kernels_that_need_room',
         'test_one_launch_form_only_when_the_group_leaves_room_for_a_collective_kernel',
         'test_two_rank_threads_exchange_for_real',
         'test_p2p_exchange_two_processes_one_gpu',
         'test_p2p_sync_chain_two_processes_one_gpu',
         'test_p2p_exchange_layer_of_the_other_codecs_in_one_launch',
         'test_a_layer_launch_whose_peer_never_answers_stores_nothing_and_the_context_recovers']


@pytest.mark.parametrize("case", CASES)
def test_exchange_layer(case):
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "xlayer_cases.py"), "-q", "-x", "-m", "gpu", "-k", case, "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    tail = (r.stdout[-3000:] + r.stderr[-1500:])
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
