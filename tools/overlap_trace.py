"""Summarise a rocprofv3 --kernel-trace CSV of tools/overlap_bench.py: per exchange kernel, how much of its duration lies
under an attention kernel (and under ANY kernel of the compute queue) running at the same time on the other queue.
usage: python tools/overlap_trace.py trace.csv [out.json]"""
import bisect
import csv
import json
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))


def is_attn(n):
    return "attn_fwd" in n or "attention" in n.lower() or "fmha" in n.lower() or "flash" in n.lower()


def is_xchg(n):
    n = n.replace("void ", "")
    return n.startswith("k_absmean") or n.startswith("k_binary") or "k_replicate" in n or n.startswith("k_int2") or n.startswith("k_flag")


main_q = Counter(r.get("Queue_Id") for r in rows if is_attn(r["Kernel_Name"])).most_common(1)
main_q = main_q[0][0] if main_q else None
ev = []
for r in rows:
    n = r["Kernel_Name"]
    kind = "attn" if is_attn(n) else ("xchg" if is_xchg(n) else ("merge" if "k_attn_merge" in n else "other"))
    if kind == "xchg" and r.get("Queue_Id") == main_q:
        kind = "xchg_main"          # work the leg keeps on the compute queue (the ready-flag launch; round-1 style legs: everything)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, n.replace("void ", "")[:48], r.get("Queue_Id", "")))
ev.sort()
attn = [(a, b) for a, b, k, _, _ in ev if k == "attn"]
comp = sorted((a, b) for a, b, k, _, q in ev if q == main_q)


def overlap(a, b, spans, starts):
    ov = 0
    i = max(0, bisect.bisect_left(starts, a) - 2)
    while i < len(spans) and spans[i][0] < b:
        lo, hi = max(a, spans[i][0]), min(b, spans[i][1])
        if hi > lo:
            ov += hi - lo
        i += 1
    return ov


a_starts, c_starts = [a for a, _ in attn], [a for a, _ in comp]
out = {}
tot_all = ov_attn_all = ov_comp_all = 0
for kname in sorted({n for _, _, k, n, _ in ev if k == "xchg"}):
    tot = ova = ovc = cnt = 0
    for a, b, k, n, _ in ev:
        if n != kname or k != "xchg":
            continue
        cnt += 1
        tot += b - a
        ova += overlap(a, b, attn, a_starts)
        ovc += overlap(a, b, comp, c_starts)
    out[kname + " [exchange queue]"] = {"calls": cnt, "avg_us": round(tot / cnt / 1e3, 2), "fraction_under_attention_kernels": round(ova / tot, 3),
                                        "fraction_under_any_compute_queue_kernel": round(ovc / tot, 3)}
    if not kname.startswith("k_flag"):          # a waiting flag kernel idles by design: not exchange work
        tot_all += tot; ov_attn_all += ova; ov_comp_all += ovc
on_main = Counter(n for _, _, k, n, _ in ev if k == "xchg_main")
merges = [(b - a) for a, b, k, _, _ in ev if k == "merge"]
res = {"what": "rocprofv3 --kernel-trace of tools/overlap_bench.py: exchange kernels dispatched on the EXCHANGE queue and the share of their "
               "run time that lies under an attention kernel / under any kernel executing on the compute queue at the same time",
       "queues": sorted({q for *_, q in ev}), "compute_queue": main_q, "exchange_kernels": out,
       "exchange_work_total": {"fraction_under_attention_kernels": round(ov_attn_all / max(1, tot_all), 3),
                               "fraction_under_any_compute_queue_kernel": round(ov_comp_all / max(1, tot_all), 3),
                               "note": "all exchange-queue kernels except the waiting flag kernels, weighted by duration"},
       "exchange_kernels_on_the_compute_queue": dict(on_main),
       "attention_kernel_avg_us": round(sum(b - a for a, b in attn) / max(1, len(attn)) / 1e3, 2), "attention_kernels": len(attn),
       "merge_kernel_avg_us": round(sum(merges) / max(1, len(merges)) / 1e3, 2), "merge_kernels": len(merges)}
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
