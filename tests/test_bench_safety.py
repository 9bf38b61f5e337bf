"""bench.py's N > 1 safety net (tools/benchlib/safety.py) on CPU: two gloo processes, CPU tensors, no libcfx.

VERDICT round 5, task 7: `validate` / `fall_back` / `states_consistent` - what a first multi-GPU run depends on - importable and tested
against a deliberately poisoned state on the gloo path (the same functions run on one GPU with two processes in tests/test_gpu_bench.py).
Cases, each on BOTH ranks: a consistent run validates clean; a reconstruction poisoned on ONE rank fails the validation on EVERY rank with
the same reason; a gate time-out counted on one rank does too and outranks a state mismatch; the ladder then takes every rank down the
same rungs - p2p -> native (when a communicator exists) -> torch -> SystemExit - and records the story."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
from benchlib import safety          # noqa: E402

L, WL, N, C = 5, 8, 16, 1024


def _states(rank, live, seed=0):
    """own_base of `rank` and its peer_base laid out as bench.py does: peer p < live - 1 is rank (rank + 1 + p) mod live, the rest loop-back"""
    def own(r):
        g = torch.Generator().manual_seed(100 * seed + r)
        return torch.randn(L, 2, N, C, generator=g).half()
    mine = own(rank)
    peers = torch.empty(L, WL - 1, 2, N, C, dtype=torch.float16)
    for p in range(WL - 1):
        peers[:, p] = own((rank + 1 + p) % live) if (live > 1 and p < live - 1) else mine
    return mine, peers


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    dev = torch.device("cpu")

    def run(label, own, peers, gate_errors):
        return safety.validate(torch, dist, label, True, world, gate_errors,
                               lambda: safety.states_consistent(torch, dist, own, peers, rank, world, world, 1, WL - 1), dev)
    own, peers = _states(rank, world)
    res["clean"] = run("after the warm-up steps", own, peers, 0)
    # a stale line's worth of wrong bits in what RANK 1 reconstructed for rank 0's shard (layer 0, K): every rank must hear of it
    own, peers = _states(rank, world)
    if rank == 1:
        peers[0, 0, 0].view(torch.int16)[0, :8] += 1
    res["poisoned"] = run("after the timed region", own, peers, 0)
    # outside the sampled window: the check compares the first SAMPLE_HALVES halves of the sampled tensors - documented, not an accident
    own, peers = _states(rank, world)
    if rank == 1:
        peers[0, 0, 0].reshape(-1).view(torch.int16)[safety.SAMPLE_HALVES + 5] += 1
    res["poisoned_outside_the_sample"] = run("after the timed region", own, peers, 0)
    # a gate time-out on rank 0 only (its count is read locally): all ranks agree, and it outranks a state mismatch on another rank
    own, peers = _states(rank, world)
    if rank == 1:
        peers[L - 1, 0, 1].view(torch.int16)[0, 0] += 1
    res["timeout"] = run("after the warm-up steps", own, peers, 3 if rank == 0 else 0)
    # the ladder: every rank walks the same rungs for the same (all-reduced) reason
    lad = safety.Ladder("p2p", None)
    rungs = [lad.down(res["poisoned"], True, world)]
    rungs.append(lad.down("again", True, world))
    try:
        lad.down("and again", True, world)
        rungs.append("no exit")
    except SystemExit as e:
        rungs.append("exit: " + str(e)[:40])
    res["rungs"], res["story"] = rungs, lad.text
    lad2 = safety.Ladder("p2p", "set-up story")
    res["rungs_no_comm"] = [lad2.down("x", False, world)]
    res["story_no_comm"] = lad2.text
    torch.save(res, f"{out}.r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_validate_and_ladder_agree_on_every_rank(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "res")
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(out + ".r0.pt"), torch.load(out + ".r1.pt")
    assert r0 == r1, "the ranks disagree about a validation"
    assert r0["clean"] is None
    assert r0["poisoned"] is not None and "a reconstructed state differs from its owner's" in r0["poisoned"] and r0["poisoned"].startswith("after the timed region")
    assert r0["poisoned_outside_the_sample"] is None
    assert r0["timeout"] is not None and "timed out" in r0["timeout"]
    assert r0["rungs"][:2] == ["native", "torch"] and r0["rungs"][2].startswith("exit: ")
    assert "peer-to-peer exchange layer" in r0["story"] and " ; then " in r0["story"] and "torch.distributed.all_gather_into_tensor" in r0["story"]
    assert r0["rungs_no_comm"] == ["torch"] and r0["story_no_comm"].startswith("set-up story ; then ")


def test_single_rank_consistency_and_sampling():
    """N = 1: every looped-back peer state must equal the sender's; sample_keys covers first / second / middle / last layer, K and V."""
    own, peers = _states(0, 1)
    ok, _ = safety.states_consistent(torch, None, own, peers, 0, 1, 1, 1, WL - 1)
    assert ok
    peers[L // 2, 3, 1].view(torch.int16)[2, 7] ^= 1
    ok, why = safety.states_consistent(torch, None, own, peers, 0, 1, 1, 1, WL - 1)
    assert not ok and "looped-back peer" in why
    assert safety.sample_keys(57, 1) == sorted({(l, kv) for l in (0, 1, 28, 56) for kv in (0, 1)})
    assert safety.sample_keys(57, 7) == sorted({(l, kv) for l in (0, 1, 6, 28, 56) for kv in (0, 1)})
    assert safety.sample_keys(1, 1) == [(0, 0), (0, 1)]
    assert safety.validate(torch, None, "x", False, 1, 5, lambda: (False, "never called"), None) is None      # no collective in the path: nothing to validate


def test_ladder_refuses_when_nothing_is_left():
    with pytest.raises(SystemExit):
        safety.Ladder("torch").down("r", True, 2)
    with pytest.raises(SystemExit):
        safety.Ladder("native").down("r", True, 1)              # one rank: torch.distributed is not a rung
    assert safety.Ladder("xgate").down("r", True, 1) == "native"
    assert safety.Ladder("native").down("r", True, 2, stream_mode=2) == "native"      # the exchange stream of the pipelined replay counts as a rung
    assert np.isclose(1, 1)


def test_a_refused_step_waits_for_the_validation_instead_of_ending_the_run(monkeypatch, capsys):
    """runner.guarded_step (round 6): a native call that refuses to launch because an earlier wait of this rank timed out (CFX_ERR_GATE ->
    workload.GateTripped) stops the rank's step loop until the next validation, where check_run reports it like a counted gate error -
    with --warmup > 1 the second step's refusal used to end a multi-rank run with a RuntimeError.  Any other error still raises."""
    from benchlib import runner, workload

    class FakeLib:
        def cfx_last_error_string(self, ctx):
            return b"an earlier gate / flag wait on this context timed out"

        def cfx_gate_errors(self, ctx):
            return 0

    class S:
        gate_tripped, rank, use_dist, world, dev, ctx, lib = False, 0, True, 1, None, None, FakeLib()
        check = workload.Run.check
    s = S()
    with pytest.raises(workload.GateTripped):
        s.check(-8, "plan_run")
    with pytest.raises(RuntimeError) as ei:
        s.check(-3, "plan_run")
    assert not isinstance(ei.value, workload.GateTripped)
    issued = []

    def fake_step(S_, i):
        issued.append(i)
        if i == 1:
            S_.check(-8, "plan_run(exchange)")
    monkeypatch.setattr(runner, "one_step", fake_step)
    for i in range(5):
        runner.guarded_step(s, i)
    assert issued == [0, 1] and s.gate_tripped, "steps behind the refused one must not be issued"
    assert "refused" in capsys.readouterr().err
    # the validation counts the refusal as a gate time-out (and clears it)
    seen = {}
    monkeypatch.setattr(runner.safety, "validate", lambda torch, dist, label, use_dist, world, ge, consistent, dev: seen.setdefault("ge", ge) and "tripped")

    class T:
        class cuda:
            @staticmethod
            def synchronize(dev):
                pass
    s.torch, s.dist = T, None
    assert runner.check_run(s, "after the warm-up steps") == "tripped" and seen["ge"] == 1 and not s.gate_tripped
