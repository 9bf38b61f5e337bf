"""Developer probe: the kernels of one LOW_RANK layer of tools/overlap_bench.py --preset lowrank8 --legs default from a rocprofv3
--kernel-trace CSV (queue, start / end relative to the layer's factor chain, name): the chain on the compute queue, the publish-and-wait
and the per-peer reconstructions on the exchange queue, the attention blocks and merges.  usage: python tools/lowrank_lane_dump.py trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), r["Kernel_Name"].replace("void ","")[:60]) for r in rows)
idx = [i for i, e in enumerate(ev) if e[3].startswith("k_lrs")]
i0 = idx[int(len(idx) * 0.8)]
t0 = ev[i0][0]
for a, b, q, name in ev[max(0, i0 - 6): i0 + 60]:
    print(f"q{q} {(a - t0) / 1e3:9.2f} {(b - t0) / 1e3:9.2f}  {(b - a) / 1e3:7.2f} us  {name}")
