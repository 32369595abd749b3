"""Kernel sequence between two consecutive actor launches of a training run's kernel trace (one fused rollout step of any
algorithm): names, durations, gaps.  usage: rollout_seq.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = [r["Kernel_Name"].split("(")[0][-70:] for r in rows]
S = [int(r["Start_Timestamp"]) for r in rows]; E = [int(r["End_Timestamp"]) for r in rows]
act = [i for i, n in enumerate(nm) if "actor_forward" in n or "actor_rollout16" in n]
# steps = consecutive actor launches with a flex_step in between and few kernels
steps = [(a, b) for a, b in zip(act, act[1:]) if 1 < b - a <= 12 and any("flex_step" in nm[k] for k in range(a, b))]
print(len(steps), "rollout steps found")
a, b = steps[len(steps) // 2]
for k in range(a, b):
    print("%7.1f us  gap %5.1f  dur %6.1f  %s" % ((S[k] - S[a]) / 1e3, (S[k] - E[k - 1]) / 1e3 if k > a else 0.0, (E[k] - S[k]) / 1e3, nm[k]))
print("period %.1f us" % ((S[b] - S[a]) / 1e3))
per = sorted((S[y] - S[x]) / 1e3 for x, y in steps)
print("median period %.1f us" % per[len(per) // 2])
