"""N1 stand-in (-m gpu): error compounding across layers AND ranks.  Golden group G13 (tests/golden/make_golden_stack.py) ran the
REFERENCE's compact_compress / compact_decompress through a seeded 4-layer attention stack over two gloo ranks for 8 steps and stored
the PSNR of the stack's final output against the same stack with the exact K,V exchanged.  Here the HIP path runs the same stack
through `compact_fwd` (two processes sharing GPU 0) and must reproduce those PSNRs within 0.1 dB, per rank and step."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import _dist_workers as W

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "g13_stack.npz")


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _entry(rank, fn_name, world, port, out, args):
    W.run(getattr(W, fn_name), rank, world, port, out, *args, device="cuda")


@pytest.mark.parametrize("codec,tol,steady", [("BINARY", 0.1, False), ("INT2", 0.1, False), ("lowrank8", 0.1, False), ("lowrankq32", 0.5, False),
                                              ("BINARY", 0.1, True), ("lowrank8", 0.1, True), ("lowrankq32", 0.5, True)])
def test_stack_psnr_matches_the_reference(tmp_path, codec, tol, steady):
    """lowrank8 / lowrankq32: the reference's LOW_RANK r = 8 and LOW_RANK_Q r = 32 presets (slow path), both sides iterating from
    the same pinned start matrix; q32 wider: the int4 re-quantisation of the factors turns last-bit differences between Cholesky-QR
    and Householder QR into whole quantisation levels (as in the single-layer trace G12).  Round 6 ran the reference's OTHER execution mode
    (@torch.compile) on the same stack (tests/golden/g13_stack_lrq32_compiled.npz, make_golden_stack.py with TORCHDYNAMO_DISABLE=0): its own two
    modes differ by up to 0.29 dB on single steps and by 0.02-0.04 dB in the mean over the steps; the HIP path sits 0.34 / 0.43 dB (single
    step) and 0.02 dB (mean) from the eager golden, 0.38 / 0.33 from the compiled trace (profiles/r06_stack_lrq32.txt) - three
    implementations of one arithmetic, pairwise equally far apart on single steps, together in the mean.  Held: per step within 0.5 dB,
    MEAN over the steps within 0.05 dB (the other codecs: per step 0.1, mean 0.05)"""
    if not os.path.exists(GOLD):
        pytest.skip("G13 golden vectors not generated")
    gold = np.load(GOLD)
    if f"{codec.lower()}/r0/psnr" not in gold.files:
        pytest.skip(f"G13 has no {codec} run")
    out = str(tmp_path / "res")
    for attempt in range(2):
        mp.start_processes(_entry, args=("w_stack", 2, _port(), out, (codec, steady)), nprocs=2, join=True, start_method="spawn")
        # The two rank processes share ONE GPU here, and a layer launch waits INSIDE the kernel for the peer's packets: when the scheduler
        # time-slices the two processes coarsely a wait can outlast the 5 s gate timeout (seen as runs of 30+ s instead of 10).  The run
        # recovers and the ranks stay consistent, but a sender whose launch gave up skipped one error-feedback update - its chain is not the
        # golden run's any more.  That is the test rig, not the product (one GPU per rank): such a run is repeated once.
        if all(int(np.load(out + f".r{r}.npz")["timeouts"][0]) == 0 for r in range(2)):
            break
    for r in range(2):
        if codec.startswith("lowrank") and steady:
            # round 6: with the lane at its default (and the profiler's scopes off: the steady layers take the calls) the low-rank layers run
            # their factor chain on the compute lane and the peer's reconstruction on the exchange lane behind the publish-and-wait -
            # here between two REAL rank processes
            assert int(np.load(out + f".r{r}.npz")["lane_ops"][0]) > 0, "the low-rank layers never took the lane form"
        got = np.load(out + f".r{r}.npz")["psnr"]
        want = gold[f"{codec.lower()}/r{r}/psnr"]
        assert got.shape == want.shape
        assert got[0] > 100 and want[0] > 100, "step 0 is WARMUP: the exchange is exact"
        d = np.abs(got[1:] - want[1:])
        assert d.max() < tol, f"{codec} rank {r}: final-output PSNR departs from the reference by {d.max():.3f} dB (step {1 + int(d.argmax())}): {got} vs {want}"
        assert np.all(want[1:] < 80), "the compressed run must differ from the exact one"
        assert abs(float(got[1:].mean() - want[1:].mean())) < 0.05, f"{codec} rank {r}: mean PSNR {got[1:].mean():.3f} vs {want[1:].mean():.3f}"
        comp = os.path.join(HERE, "golden", "g13_stack_lrq32_compiled.npz")
        if codec == "lowrankq32" and os.path.exists(comp):
            # ... and not further from the reference's compiled trace than 0.5 dB either (the reference runs compiled on its own hardware)
            other = np.load(comp)[f"lowrankq32/r{r}/psnr"]
            assert np.abs(got[1:] - other[1:]).max() < tol and abs(float(got[1:].mean() - other[1:].mean())) < 0.08
