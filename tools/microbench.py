#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box (developer tool, not part of the product or the tests).

Sweeps rows-per-tile for each codec at a given shape/batch and prints per-kernel average durations measured by the
native hipEvent hooks of libcfx.so, next to the float4 copy probe (achievable HBM bandwidth in the same units)."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from compactfusion_amd import _lib, codecs as K

ALG = {1: (6.125, 4.125), 2: (6.25, 4.25), 3: (6.5, 4.5), 4: (7.0, 5.0), 5: (None, None)}


def read_prof(lib, ctx, cap=100000):
    ids = (ctypes.c_int * cap)()
    ms = (ctypes.c_float * cap)()
    n = lib.cfx_profile_read(ctx, ids, ms, cap)
    agg = {}
    for i in range(n):
        agg.setdefault(ids[i], []).append(ms[i])
    return {lib.cfx_kernel_name(k).decode(): (sum(v) / len(v) * 1e3, len(v)) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=544)
    ap.add_argument("--C", type=int, default=3072)
    ap.add_argument("--codecs", default="1,2,3,4,5")
    ap.add_argument("--rows", default="0,16,32,64,128")
    ap.add_argument("--cbatch", type=int, default=2)
    ap.add_argument("--dbatch", type=int, default=14)
    ap.add_argument("--sets", type=int, default=24, help="distinct tensor sets cycled through (cold HBM)")
    ap.add_argument("--iters", type=int, default=96)
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    ctx = K.context(0)
    N, C = args.N, args.C
    s = torch.cuda.current_stream().cuda_stream
    # copy probe
    nb = 96 * 1024 * 1024
    srcs = [torch.empty(nb, dtype=torch.uint8, device=dev).random_(0, 255) for _ in range(8)]
    dsts = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(8)]
    lib.cfx_profile_enable(ctx, 4096, 0xffffffff, 1)
    for i in range(32):
        lib.cfx_copy_probe(ctx, dsts[i % 8].data_ptr(), srcs[i % 8].data_ptr(), nb, s)
    torch.cuda.synchronize()
    for name, (us, n) in read_prof(lib, ctx).items():
        print(f"copy probe {nb/1e6:.0f} MB: {us:.2f} us -> {2*nb/us/1e3:.0f} GB/s (read+write, event-bracketed)")
    del srcs, dsts
    g = torch.Generator(device=dev).manual_seed(0)
    S = args.sets
    xb = [torch.randn(args.dbatch, N, C, generator=g, device=dev).half() for _ in range(S)]
    xx = [(b[:args.cbatch].float() + 0.1 * torch.randn(args.cbatch, N, C, generator=g, device=dev)).half() for b in xb]
    for cid in [int(c) for c in args.codecs.split(",")]:
        param = 8 if cid == 5 else 0
        pbytes = K.packet_bytes(cid, N, C, param)
        slot = (pbytes + 255) // 256 * 256
        pk = [torch.zeros(args.cbatch, slot, dtype=torch.uint8, device=dev) for _ in range(S)]
        wsb = lib.cfx_workspace_bytes(cid, N, C, param, args.cbatch)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        citems, ditems = [], []
        for i in range(S):
            a = (_lib.CompItem * args.cbatch)()
            for j in range(args.cbatch):
                a[j] = _lib.CompItem(xx[i][j].data_ptr(), xb[i][j].data_ptr(), xb[i][j].data_ptr(), pk[i][j].data_ptr())
            citems.append(a)
            d = (_lib.DecompItem * args.dbatch)()
            for j in range(args.dbatch):
                d[j] = _lib.DecompItem(pk[i][j % args.cbatch].data_ptr(), xb[i][j].data_ptr(), xb[i][j].data_ptr())
            ditems.append(d)
        for rows in [int(r) for r in args.rows.split(",")]:
            lib.cfx_set_rows_per_tile(ctx, rows)
            for i in range(S):   # warm
                lib.cfx_compress_batch(ctx, cid, N, C, param, 1, args.cbatch, citems[i], ws.data_ptr(), wsb, s)
            torch.cuda.synchronize()
            lib.cfx_profile_enable(ctx, 8 * args.iters + 64, 0xffffffff, 1)
            for i in range(args.iters):
                rc = lib.cfx_compress_batch(ctx, cid, N, C, param, 1, args.cbatch, citems[i % S], ws.data_ptr(), wsb, s)
                assert rc == 0, lib.cfx_last_error_string(ctx)
            torch.cuda.synchronize()
            pc = read_prof(lib, ctx)
            lib.cfx_profile_enable(ctx, 8 * args.iters + 64, 0xffffffff, 1)
            for i in range(args.iters):
                rc = lib.cfx_decompress_batch(ctx, cid, N, C, param, args.dbatch, ditems[i % S], s)
                assert rc == 0, lib.cfx_last_error_string(ctx)
            torch.cuda.synchronize()
            pd = read_prof(lib, ctx)
            lib.cfx_profile_enable(ctx, 0, 0, 1)
            # wall time of whole compress sequence without events
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for i in range(args.iters):
                lib.cfx_compress_batch(ctx, cid, N, C, param, 1, args.cbatch, citems[i % S], ws.data_ptr(), wsb, s)
            t1.record(); torch.cuda.synchronize()
            cw = t0.elapsed_time(t1) / args.iters * 1e3
            t0.record()
            for i in range(args.iters):
                lib.cfx_decompress_batch(ctx, cid, N, C, param, args.dbatch, ditems[i % S], s)
            t1.record(); torch.cuda.synchronize()
            dw = t0.elapsed_time(t1) / args.iters * 1e3
            el_c, el_d = args.cbatch * N * C, args.dbatch * N * C
            ac, ad = ALG[cid]
            line = f"codec {cid} rows {rows:3d} | compress seq {cw:7.2f} us"
            if ac:
                line += f" ({ac*el_c/cw/1e3:6.0f} GB/s alg)"
            line += " [" + ", ".join(f"{k} {v[0]:.2f}" for k, v in pc.items()) + f"] | decompress {dw:7.2f} us"
            if ad:
                line += f" ({ad*el_d/dw/1e3:6.0f} GB/s alg)"
            line += " [" + ", ".join(f"{k} {v[0]:.2f}" for k, v in pd.items()) + "]"
            print(line, flush=True)
        lib.cfx_set_rows_per_tile(ctx, 0)


if __name__ == "__main__":
    main()
