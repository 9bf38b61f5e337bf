#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs into per-kernel HBM traffic per launch (developer tool).

Usage: tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [config-key JSON of the profiled bench command] [steps profiled] [note]
Each dir holds the *_counter_collection.csv of ONE --pmc pass (FETCH_SIZE and WRITE_SIZE do not fit one pass,
MI355X_MICROARCH.md §rocprofv3 PMC slots).  Units and corrections follow MI355X_MICROARCH.md §HBM:
  * FETCH_SIZE / WRITE_SIZE are reported in KiB;
  * on gfx950 FETCH_SIZE counts exactly half of the bytes of a wide coalesced streaming read -> x2;
  * WRITE_SIZE is calibrated on the float4 copy probe (k_copy_probe: 96 MiB written per launch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(list)
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") != counter:
                continue
            key = r["Dispatch_Id"]
            per_dispatch[key] += float(r["Counter_Value"])
            names[key] = r["Kernel_Name"]
        for k, v in per_dispatch.items():
            acc[names[k]].append(v)
    # k_binary_pipe: keep the full three-group launches only (the pipeline's prologue / epilogue launches of every step
    # carry one or two groups and a fraction of the traffic)
    for name in list(acc):
        if "k_binary_pipe" in name and acc[name]:
            top = max(acc[name])
            acc[name] = [v for v in acc[name] if v >= 0.9 * top]
    return acc


def main():
    fdir, wdir, out = sys.argv[1:4]
    cfg = json.loads(sys.argv[4]) if len(sys.argv) > 4 else None
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else None
    note = sys.argv[6] if len(sys.argv) > 6 else None
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    probe_bytes = 96 * 1024 * 1024
    res = {"unit": "bytes per launch", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)", "kernels": {}}

    def avg(v):
        return sum(v) / len(v) if v else None

    pf = next((avg(v) for k, v in fetch.items() if "k_copy_probe" in k), None)
    pw = next((avg(v) for k, v in write.items() if "k_copy_probe" in k), None)
    fetch_scale = 2.0 * 1024                        # KiB -> bytes, x2 gfx950 wide-read correction (guide)
    write_scale = (probe_bytes / pw) if pw else 1024.0
    res["calibration"] = {"copy_probe_bytes_each_way": probe_bytes, "FETCH_SIZE_raw_KiB": pf, "WRITE_SIZE_raw_KiB": pw,
                          "fetch_bytes_per_unit": fetch_scale, "write_bytes_per_unit": write_scale,
                          "fetch_check_ratio": (pf * fetch_scale / probe_bytes) if pf else None}
    for k in sorted(set(fetch) | set(write)):
        f, w = avg(fetch.get(k, [])), avg(write.get(k, []))
        short = k.split("(")[0].replace("void ", "")
        res["kernels"][short] = {"launches": len(fetch.get(k, []) or write.get(k, [])),
                                 "fetch_bytes": None if f is None else f * fetch_scale,
                                 "write_bytes": None if w is None else w * write_scale,
                                 "hbm_bytes": None if (f is None or w is None) else f * fetch_scale + w * write_scale}
    deq = res["kernels"].get("k_binary_dequant")
    if deq and deq["hbm_bytes"]:
        res["k_binary_dequant_bytes_per_launch"] = int(deq["hbm_bytes"])
    pipe = res["kernels"].get("k_binary_pipe<true>") or res["kernels"].get("k_binary_pipe")
    if pipe and pipe["hbm_bytes"]:
        res["k_binary_pipe_bytes_per_launch"] = int(pipe["hbm_bytes"])
    # what bench.py matches against: the configuration the counters were taken with, bytes per launch of every libcfx kernel
    # and the bytes one denoise step moves (all libcfx launches of the profiled run / steps profiled)
    res["config"] = cfg
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from provenance import source_sha
    res["source_sha"] = source_sha()
    if note:
        res["measured_with"] = note
    res["bytes_per_launch"] = {k: int(v["hbm_bytes"]) for k, v in res["kernels"].items() if k.startswith("k_") and v["hbm_bytes"]}
    if steps:
        tot = sum(v["hbm_bytes"] * v["launches"] for k, v in res["kernels"].items()
                  if k.startswith("k_") and "copy_probe" not in k and v["hbm_bytes"])
        res["bytes_per_step"] = int(tot / steps)
        res["steps_profiled"] = steps
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
