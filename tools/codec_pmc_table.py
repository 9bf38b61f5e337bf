#!/usr/bin/env python3
"""Merge the per-configuration PMC summaries (tools/pmc_summary.py outputs, one per configuration and form) into ONE file: per kernel HBM bytes
per launch (FETCH x2 gfx950 correction + calibrated WRITE), the layer's total against its ALGORITHMIC bytes and the ratio.
usage: python tools/codec_pmc_table.py <dir with pmc_<cfg>_<form>.json> <out.json>"""
import glob, json, os, sys
ALG = {1: (6.125, 4.125), 2: (6.25, 4.25), 3: (6.5, 4.5), 4: (7.0, 5.0), 5: (6 + 2.5 / 8, 4 + 2.5 / 8)}
CFG = {"1": (4, (4096, 1152), 1, 1, "config 1: int8 residual round trip (4096,1152)"), "2": (3, (1024, 1152), 2, 4, "config 2: PixArt-a INT4 (1024,1152), K,V + 2 peers' tensors"),
       "3": (1, (544, 3072), 2, 16, "config 3: FLUX 1-bit (544,3072), K,V + 14 peers' tensors"), "3b": (2, (544, 3072), 2, 16, "config 3 shard, 2-bit preset"),
       "4": (3, (4448, 3072), 2, 8, "config 4: CogVideoX INT4 (4448,3072), K,V + 6 peers' tensors"), "5": (5, (512, 1536), 2, 16, "config 5: SD3 top-k 1:8 (512,1536), K,V + 14 peers' tensors")}
d, out = sys.argv[1], sys.argv[2]
res = {"unit": "bytes per launch / per layer", "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of tools/codec_pmc_run.py; FETCH x2 (gfx950 wide reads), WRITE "
       "calibrated on the 96 MiB copy probe of the same run; loop-back on one stream (counter passes serialise dispatches)", "layers": {}}
for fn in sorted(glob.glob(os.path.join(d, "pmc_*_*.json"))):
    _, cfg, form = os.path.basename(fn)[:-5].split("_")
    j = json.load(open(fn))
    cid, (N, C), ncomp, nrec, what = CFG[cfg]
    c, dq = ALG[cid]
    rest = (nrec - ncomp) if cfg != "1" else 1
    alg = int(N * C * (ncomp * c + rest * dq))
    kern = {k: {"launches_per_layer": v["launches"] / 6.0, "hbm_bytes_per_launch": int(v["hbm_bytes"]), "fetch": int(v["fetch_bytes"]), "write": int(v["write_bytes"])}
            for k, v in j["kernels"].items() if k.startswith("k_") and "copy_probe" not in k and v.get("hbm_bytes")}
    tot = int(sum(v["hbm_bytes_per_launch"] * v["launches_per_layer"] for v in kern.values()))
    res["layers"][f"{cfg} / {form}"] = {"what": what, "form": form, "algorithmic_bytes_per_layer": alg, "hbm_bytes_per_layer": tot, "ratio": round(tot / alg, 3),
                                       "kernels": kern, "copy_probe_fetch_check": j["calibration"]["fetch_check_ratio"]}
json.dump(res, open(out, "w"), indent=1)
for k, v in res["layers"].items():
    print(f"{k:14s} alg {v['algorithmic_bytes_per_layer'] / 1e6:8.1f} MB  hbm {v['hbm_bytes_per_layer'] / 1e6:8.1f} MB  ratio {v['ratio']}")
