#!/usr/bin/env python3
"""Per-codec, per-config-shape kernel timing table (developer tool; output committed under profiles/).

For every codec and every BASELINE.json / SURVEY.md §8d shape: wall time per call of the compress sequence (batch of 2
tensors = K and V of a layer) and of the batched reconstruction (14 tensors = 7 peers x K,V), measured back-to-back on one
stream with hipEvents around >= 64 launches, rotating over enough distinct tensor sets to defeat the 256 MB Infinity
Cache; algorithmic GB/s = SURVEY.md §8d bytes per element x elements / time; fraction of the 8 TB/s HBM3E spec peak."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from compactfusion_amd import _lib, codecs as K

ALG = {1: (6.125, 4.125), 2: (6.25, 4.25), 3: (6.5, 4.5), 4: (7.0, 5.0), 5: (6 + 2.5 / 8, 4 + 2.5 / 8)}
NAMES = {1: "1-bit", 2: "2-bit", 3: "int4", 4: "int8", 5: "top-k 1:8"}
SHAPES = [("S1 config 1: [1,4096,1152]", 4096, 1152), ("S2 PixArt-a 512^2 SP2", 1024, 1152), ("S3 FLUX 1024^2 ring 8", 544, 3072),
          ("S4 CogVideoX-5B SP4", 4448, 3072), ("S5 SD3 1024^2 SP8", 512, 1536)]


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    ctx = K.context(0)
    s = torch.cuda.current_stream().cuda_stream
    rows = []
    for label, N, C in SHAPES:
        per_set = 16 * N * C * 2
        S = max(4, min(48, int(1.2e9 // per_set)))
        g = torch.Generator(device=dev).manual_seed(0)
        xb = [torch.randn(14, N, C, generator=g, device=dev).half() for _ in range(S)]
        xx = [(b[:2].float() + 0.1 * torch.randn(2, N, C, generator=g, device=dev)).half() for b in xb]
        for cid in (1, 2, 3, 4, 5):
            param = 8 if cid == 5 else 0
            if cid == 5 and (N * C) % 1024:
                continue
            slot = (K.packet_bytes(cid, N, C, param) + 255) // 256 * 256
            pk = [torch.zeros(14, slot, dtype=torch.uint8, device=dev) for _ in range(S)]     # 14 DISTINCT packets per set
            wsb = lib.cfx_workspace_bytes(cid, N, C, param, 2)
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
            cit, dit = [], []
            for i in range(S):
                a = (_lib.CompItem * 2)()
                for j in range(2):
                    a[j] = _lib.CompItem(xx[i][j].data_ptr(), xb[i][j].data_ptr(), xb[i][j].data_ptr(), pk[i][j].data_ptr())
                cit.append(a)
                d = (_lib.DecompItem * 14)()
                for j in range(14):
                    d[j] = _lib.DecompItem(pk[i][j].data_ptr(), xb[i][j].data_ptr(), xb[i][j].data_ptr())
                dit.append(d)
            iters = max(64, 2 * S)
            filled = [False]

            def timed(fn):
                for i in range(S):
                    fn(i)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(iters):
                    fn(i % S)
                b.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b) / iters * 1e3

            cw = timed(lambda i: lib.cfx_compress_batch(ctx, cid, N, C, param, 1, 2, cit[i], ws.data_ptr(), wsb, s))
            for i in range(S):      # every peer slot gets a real packet (copies of the two just produced)
                for j in range(2, 14):
                    pk[i][j].copy_(pk[i][j % 2])
            dw = timed(lambda i: lib.cfx_decompress_batch(ctx, cid, N, C, param, 14, dit[i], s))
            ac, ad = ALG[cid]
            rows.append({"shape": label, "N": N, "C": C, "codec": NAMES[cid], "compress_us": round(cw, 2),
                         "compress_GBps": round(ac * 2 * N * C / cw / 1e3, 0), "decompress14_us": round(dw, 2),
                         "decompress_GBps": round(ad * 14 * N * C / dw / 1e3, 0),
                         "decompress_frac_of_8TBps": round(ad * 14 * N * C / dw / 1e3 / 8000, 3)})
            print(rows[-1], flush=True)
        del xb, xx
        torch.cuda.empty_cache()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "codec_table.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rows, open(out, "w"), indent=1)
    print("| shape | codec | compress K,V (us) | alg GB/s | reconstruct 14 (us) | alg GB/s | frac of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['shape']} ({r['N']},{r['C']}) | {r['codec']} | {r['compress_us']} | {int(r['compress_GBps'])} | {r['decompress14_us']} | "
              f"{int(r['decompress_GBps'])} | {r['decompress_frac_of_8TBps']} |")


if __name__ == "__main__":
    main()
