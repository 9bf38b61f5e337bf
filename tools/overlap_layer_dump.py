"""Developer probe: print the kernels of a few consecutive layers of tools/overlap_bench.py's native leg from a rocprofv3
--kernel-trace CSV (queue, start / end relative to the first, name).  usage: python tools/overlap_layer_dump.py trace.csv [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), r["Kernel_Name"][:48]) for r in rows)
# find a compress kernel on a non-main queue somewhere in the middle and print around it
from collections import Counter
main_q = Counter(e[2] for e in ev if "attn_fwd" in e[3]).most_common(1)[0][0]
idx = [i for i, e in enumerate(ev) if "k_absmean_compress" in e[3] and e[2] != main_q]       # the chain leg: compress on the exchange queue
i0 = idx[len(idx) // 2] if idx else len(ev) // 2
t0 = ev[i0][0]
for a, b, q, name in ev[max(0, i0 - 4): i0 + n]:
    print(f"q{q} {(a - t0) / 1e3:9.2f} {(b - t0) / 1e3:9.2f}  {(b - a) / 1e3:7.2f} us  {name}")
