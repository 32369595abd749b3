"""Post-process a rocprofv3 --kernel-trace CSV of tools/update_prof_plain.py with EVENTS=n: the kernel sequence of the LAST
whole update event (from one flex_rollout_burst / first gather to the next), every launch with its duration, the idle gap
before it and its grid — where an event's wall time goes, small launches (fills, copies) included."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.split("(")[0][-64:]
names = [short(r["Kernel_Name"]) for r in rows]
# an update event of replay_event: 10 value + 1 policy sub-updates; policy sub-updates contain gru_backward_fused_kernel
pol = [i for i, n in enumerate(names) if "gru_backward_fused_kernel" in n]
if len(pol) < 3:
    print("fewer than 3 policy sub-updates in the trace"); sys.exit(0)
a, b = pol[-3] + 1, pol[-2] + 1
# extend to the end of that policy sub-update: up to and including its clip_rmsprop
def end_of(i):
    while i < len(names) and not "clip_rmsprop_kernel" in names[i]:
        i += 1
    return i + 1
a, b = end_of(a), end_of(b)
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
agg = collections.OrderedDict()
busy = 0
print(f"=== one update event: launches {a}..{b} ({b - a} launches)")
for r, n in zip(rows[a:b], names[a:b]):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
    if "--full" in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  grid {grid:>9} wg {wg:>5}  {n}")
    d = agg.setdefault(n, [0, 0.0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3; d[2] += max(0, s - prev) / 1e3
    busy += e - s; prev = e
span = (prev - t0) / 1e3
print(f"span {span:.1f} us, kernel time {busy / 1e3:.1f} us, idle {span - busy / 1e3:.1f} us")
for n, (c, dur, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {dur:9.1f} us  {100 * dur / span:5.1f} %  x{c:<4d} gap-before {gap:7.1f} us  {n}")
