"""Summarise a rocprofv3 --kernel-trace CSV of tools/overlap_bench.py: per exchange kernel, how much of its duration lies
under an attention kernel running at the same time (another queue).  usage: python tools/overlap_trace.py trace.csv [out.json]"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
from collections import Counter
main_q = Counter(r.get("Queue_Id") for r in rows if "attn_fwd" in r["Kernel_Name"]).most_common(1)
main_q = main_q[0][0] if main_q else None
for r in rows:
    n = r["Kernel_Name"]
    kind = "attn" if ("attn_fwd" in n or "attention" in n.lower() or "fmha" in n.lower() or "flash" in n.lower()) else \
           ("xchg" if n.startswith("void k_absmean") or n.startswith("k_binary") or "k_replicate" in n or n.startswith("void k_binary") else
            ("merge" if "k_attn_merge" in n else "other"))
    if kind == "xchg" and r.get("Queue_Id") == main_q:
        kind = "xchg_main"          # legs that keep compress / reconstruction on the compute queue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, n[:40], r.get("Queue_Id", "")))
ev.sort()
attn = [(a, b) for a, b, k, _, _ in ev if k == "attn"]
out = {}
import bisect
starts = [a for a, _ in attn]
for kname in sorted({n for _, _, k, n, _ in ev if k == "xchg"}):
    tot = ov = cnt = 0
    for a, b, k, n, _ in ev:
        if n != kname: continue
        cnt += 1; tot += b - a
        i = max(0, bisect.bisect_left(starts, a) - 2)
        while i < len(attn) and attn[i][0] < b:
            lo, hi = max(a, attn[i][0]), min(b, attn[i][1])
            if hi > lo: ov += hi - lo
            i += 1
    out[kname + " [exchange queue]"] = {"calls": cnt, "avg_us": round(tot / cnt / 1e3, 2), "fraction_under_attention_kernels": round(ov / tot, 3)}
queues = sorted({q for *_, q in ev})
res = {"what": "rocprofv3 --kernel-trace of tools/overlap_bench.py: exchange kernels dispatched on the EXCHANGE queue (the chain leg) and the "
               "share of their run time that lies under an attention kernel executing on the compute queue at the same time",
       "queues": queues, "exchange_kernels": out,
       "attention_kernel_avg_us": round(sum(b - a for a, b in attn) / max(1, len(attn)) / 1e3, 2), "attention_kernels": len(attn)}
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
