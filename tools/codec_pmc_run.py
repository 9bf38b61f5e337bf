#!/usr/bin/env python3
"""PMC driver (developer tool): one BASELINE configuration's codec launches, a few repetitions over distinct (cold) tensors, plus the 96 MiB copy
probe for calibration - run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes; tools/collect_profiles.sh), summarised by
tools/pmc_summary.py + tools/codec_pmc_table.py into profiles/r04_pmc_traffic_codecs.json.
usage: python tools/codec_pmc_run.py <config 1|2|3|3b|4|5> <form layer|launches>
  layer     what the library runs by default (min/max codecs on small shards: k_minmax_layer; 1-bit / 2-bit: the gated layer launch)
  launches  cfx_set_gated_launch(ctx, 0): statistics + finalize ; quantise (+ error feedback) ; reconstruction as separate launches
Counter passes SERIALISE dispatches, so everything here is loop-back on ONE stream (no flag kernel on another stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compactfusion_amd import _lib, codecs as K

CFG = {  # codec id, param, (N, C), tensors compressed per layer, tensors reconstructed per layer
    "1": (4, 0, (4096, 1152), 1, 1), "2": (3, 0, (1024, 1152), 2, 4), "3": (1, 0, (544, 3072), 2, 16), "3b": (2, 0, (544, 3072), 2, 16),
    "4": (3, 0, (4448, 3072), 2, 8), "5": (5, 8, (512, 1536), 2, 16)}
cfg, form = sys.argv[1], sys.argv[2]
cid, param, (N, C), ncomp, nrec = CFG[cfg]
lib, ctx = _lib.load(), K.context(0)
if form == "launches":
    assert lib.cfx_set_gated_launch(ctx, 0) == 0
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
L = 6
own = torch.randn(L, ncomp, N, C, generator=g, device=dev).half()
x = (own.float() + 0.1 * torch.randn(L, ncomp, N, C, generator=g, device=dev)).half()
peers = torch.randn(L, nrec, N, C, generator=g, device=dev).half()
slot = (K.packet_bytes(cid, N, C, param) + 255) // 256 * 256
pk = torch.zeros(L, ncomp, slot, dtype=torch.uint8, device=dev)
wsb = lib.cfx_workspace_bytes(cid, N, C, param, ncomp)
ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
sh = torch.cuda.current_stream().cuda_stream
nb = 96 * 1024 * 1024
src, dst = torch.empty(nb, dtype=torch.uint8, device=dev).random_(0, 255), torch.empty(nb, dtype=torch.uint8, device=dev)
for _ in range(4):
    assert lib.cfx_copy_probe(ctx, dst.data_ptr(), src.data_ptr(), nb, sh) == 0
torch.cuda.synchronize()
for l in range(L):
    c = (_lib.CompItem * ncomp)(*[_lib.CompItem(x[l, i].data_ptr(), own[l, i].data_ptr(), own[l, i].data_ptr(), pk[l, i].data_ptr()) for i in range(ncomp)])
    # own tensors: compress + error feedback (6.x B/el); the other nrec - ncomp tensors: reconstruction from the (looped-back) packets
    rest = nrec - ncomp
    d = (_lib.DecompItem * max(rest, 1))(*[_lib.DecompItem(pk[l, j % ncomp].data_ptr(), peers[l, j].data_ptr(), peers[l, j].data_ptr()) for j in range(max(rest, 1))])
    if cid == 5 or rest == 0:
        assert lib.cfx_compress_batch(ctx, cid, N, C, param, 1, ncomp, c, ws.data_ptr(), wsb, sh) == 0
        if cid == 5:
            assert lib.cfx_decompress_batch(ctx, cid, N, C, param, rest, d, sh) == 0
        else:
            assert lib.cfx_decompress_batch(ctx, cid, N, C, param, 1, d, sh) == 0          # config 1: the round trip's reconstruction
    else:
        assert lib.cfx_compress_batch_gated(ctx, cid, N, C, param, 1, ncomp, c, 0, None, rest, d, ws.data_ptr(), wsb, sh) == 0, lib.cfx_last_error_string(ctx)
    torch.cuda.synchronize()
assert lib.cfx_gate_errors(ctx) == 0
print("ok", cfg, form)
