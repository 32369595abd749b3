"""Post-process a rocprofv3 --kernel-trace CSV of tools/update_prof_plain.py: the kernel sequence of the LAST sub-update of
each kind with every kernel's duration and the idle gap before it (where a step's wall time goes beyond kernel time)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
def short(n):
    n = n.split("(")[0]
    return n[-70:]
# a sub-update starts at its gather_rows_kernel
starts = [i for i, n in enumerate(names) if "gather_rows_kernel" in n.split("(")[0]]
segs = [(a, b) for a, b in zip(starts, starts[1:] + [len(rows)])]
kinds = {}
for a, b in segs:
    seq = tuple(short(n) for n in names[a:b])
    kinds.setdefault(len(seq), []).append((a, b))
for ln, lst in sorted(kinds.items()):
    if len(lst) < 5:
        continue
    a, b = lst[-2]
    t0 = int(rows[a]["Start_Timestamp"]); prev = t0
    busy = 0
    print(f"=== sub-update with {ln} kernels ({len(lst)} seen)")
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {short(r['Kernel_Name'])}")
        busy += e - s; prev = e
    nxt = int(rows[b]["Start_Timestamp"]) if b < len(rows) else prev
    print(f"   span {(prev - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, to next start {(nxt - t0) / 1e3:.1f} us")
