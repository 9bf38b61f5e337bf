#!/usr/bin/env python3
"""Developer soak of the peer-to-peer exchange layer (cfx_plan_add_exchange_layer_p2p): two rank processes on one GPU, hundreds of steps,
final states compared across the ranks (the worker is tests/xlayer_cases.py::_p2p_worker).  Packets written by one process's workgroups on
some XCDs are read in place by the other process's workgroups on others - the cross-XCD half of the protocol's memory ordering."""
import os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch.multiprocessing as mp
import xlayer_cases as T
if __name__ == "__main__":
    for (N, C, L, steps) in ((544, 3072, 8, 300), (96, 1024, 6, 500)):
        with tempfile.TemporaryDirectory() as td:
            mp.start_processes(T._p2p_worker, args=(2, td, L, N, C, steps, True), nprocs=2, join=True, start_method="spawn")
            ok = all(np.array_equal(np.load(os.path.join(td, f"peer{1-r}_{r}.npy")), np.load(os.path.join(td, f"own{r}.npy"))) for r in range(2))
            print((N, C, L, steps), "consistent" if ok else "MISMATCH", flush=True)
