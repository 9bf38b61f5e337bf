#!/usr/bin/env python3
"""Developer measurement: the LOW_RANK_Q-32 run of the G13 stack (tests/test_gpu_stack.py) - per-step |PSNR difference| of the HIP path to the
reference's eager trace (the committed golden) and to its @torch.compile trace (tests/golden/g13_stack_lrq32_compiled.npz), beside the
reference's own eager-to-compiled difference."""
import os, sys, numpy as np, socket
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch.multiprocessing as mp
import test_gpu_stack as T

if __name__ == "__main__":
    out = "/tmp/stackres"
    mp.start_processes(T._entry, args=("w_stack", 2, T._port(), out, ("lowrankq32",)), nprocs=2, join=True, start_method="spawn")
    g = np.load("tests/golden/g13_stack.npz"); c = np.load("tests/golden/g13_stack_lrq32_compiled.npz")
    for r in range(2):
        got = np.load(out + f".r{r}.npz")["psnr"]; e = g[f"lowrankq32/r{r}/psnr"]; k = c[f"lowrankq32/r{r}/psnr"]
        print(r, "hip-eager", np.round(np.abs(got - e)[1:], 3), "max", round(float(np.abs(got - e)[1:].max()), 3), "| hip-compiled max", round(float(np.abs(got - k)[1:].max()), 3), "| eager-compiled max", round(float(np.abs(e - k)[1:].max()), 3), "| mean diff", round(float(abs(got[1:].mean() - e[1:].mean())), 3))

