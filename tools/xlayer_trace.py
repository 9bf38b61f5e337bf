"""Developer tool: from a rocprofv3 kernel-trace CSV of a bench step, the layer launches' durations and the gaps between them; where the
exchange runs as a kernel of its own on the exchange stream (`--p2p off`, or the collective form) also how that flag kernel sits relative to
the layer kernel: start / end after the layer kernel's start, layer kernel end after the flag kernel's end (us).  With the peer-to-peer
exchange INSIDE the layer launch (the default since mid round 4) there is no flag kernel: the step is one kernel per layer.
usage: python tools/xlayer_trace.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
lay = sorted([r for r in rows if "compress" in r["Kernel_Name"] or "minmax_layer" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
flg = sorted([r for r in rows if "k_flag" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
print(len(lay), "layer launches,", len(flg), "flag kernels;", "layer kernel:", lay[0]["Kernel_Name"][:60] if lay else None)
j = 0
ds, de, tail, dur, gap = [], [], [], [], []
for i, r in enumerate(lay):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur.append((e - s) / 1e3)
    if i:
        gap.append((s - int(lay[i - 1]["End_Timestamp"])) / 1e3)
    while j < len(flg) and int(flg[j]["End_Timestamp"]) < s:
        j += 1
    if j < len(flg) and int(flg[j]["Start_Timestamp"]) < e:
        fs, fe = int(flg[j]["Start_Timestamp"]), int(flg[j]["End_Timestamp"])
        ds.append((fs - s) / 1e3); de.append((fe - s) / 1e3); tail.append((e - fe) / 1e3)
def q(v): return "n=%d p10 %.1f p50 %.1f p90 %.1f" % (len(v), st.quantiles(v, n=10)[0], st.median(v), st.quantiles(v, n=10)[-1]) if len(v) > 10 else str(v)
print("layer kernel duration      ", q(dur))
print("gap between layer kernels  ", q(gap))
if flg:
    print("flag kernel start - layer start", q(ds))
    print("flag kernel end   - layer start", q(de))
    print("layer end - flag kernel end    ", q(tail))
else:
    print("no exchange-stream kernel in the trace: the exchange runs inside the layer launch (one kernel per layer)")
