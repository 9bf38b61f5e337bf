"""Developer tool: per-queue busy / idle summary of the last bench step in a rocprofv3 kernel trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r["Kernel_Name"].startswith(("k_binary", "void k_absmean", "k_absmean"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
L = int(sys.argv[2]) if len(sys.argv) > 2 else 57
rows = rows[-3 * L:]
t0 = int(rows[0]["Start_Timestamp"])
span = (max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e3
print(f"step span {span:.1f} us")
byq = {}
for r in rows:
    byq.setdefault(r["Queue_Id"], []).append(((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Kernel_Name"][:20]))
for q, ev in byq.items():
    busy = sum(e - s for s, e, _ in ev)
    gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
    big = [g for g in gaps if g > 3]
    print(f"queue {q}: {len(ev)} kernels, busy {busy:.1f} us, first start {ev[0][0]:.1f}, last end {ev[-1][1]:.1f}, gaps>3us: {len(big)} totalling {sum(big):.1f} us")
    names = {}
    for s, e, n in ev:
        names.setdefault(n, []).append(e - s)
    for n, d in names.items():
        print(f"    {n}: n={len(d)} avg {sum(d) / len(d):.2f} us")
if "-v" in sys.argv:
    for q, ev in byq.items():
        for s, e, n in ev:
            print(q, n, f"{s:.1f} {e:.1f}")
