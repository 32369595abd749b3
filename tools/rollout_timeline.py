"""Post-process a rocprofv3 --kernel-trace CSV of a training run: start-to-start period of the fused rollout step (actor,
env) inside graph replays, each kernel's duration and the idle gaps between them."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = [r["Kernel_Name"].split("(")[0] for r in rows]
S = [int(r["Start_Timestamp"]) for r in rows]
E = [int(r["End_Timestamp"]) for r in rows]
runs, i = [], 0
while i + 1 < len(rows):
    j = i
    while j + 1 < len(rows) and ("actor_forward" in nm[j] or "actor_rollout16" in nm[j]) and "flex_step_kernel" in nm[j + 1]:
        j += 2
    if j - i >= 16:
        runs.append((i, j))
    i = max(j, i + 1)
print(len(runs), "runs of >= 8 fused steps")
tot = {"actor": 0, "env": 0, "gap_ae": 0, "gap_ea": 0, "n": 0}
for a, b in runs:
    for k in range(a, b - 2, 2):
        tot["actor"] += E[k] - S[k]; tot["env"] += E[k + 1] - S[k + 1]
        tot["gap_ae"] += S[k + 1] - E[k]; tot["gap_ea"] += S[k + 2] - E[k + 1]; tot["n"] += 1
n = max(tot["n"], 1)
print("per step: actor %.1f us, gap %.1f, env %.1f us, gap %.1f  => period %.1f us over %d steps" % (
    tot["actor"] / n / 1e3, tot["gap_ae"] / n / 1e3, tot["env"] / n / 1e3, tot["gap_ea"] / n / 1e3,
    (tot["actor"] + tot["env"] + tot["gap_ae"] + tot["gap_ea"]) / n / 1e3, n))
# gaps between graph replays (runs)
seams = [(S[b2] - E[b1 - 1]) / 1e3 for (a1, b1), (b2, _) in zip(runs, runs[1:]) if b2 == b1]
between = [(S[a2] - E[b1 - 1]) / 1e3 for (a1, b1), (a2, _) in zip(runs, runs[1:])]
between.sort()
if between:
    print("gap between consecutive runs: median %.1f us, p10 %.1f, p90 %.1f (n=%d)" % (
        between[len(between) // 2], between[len(between) // 10], between[9 * len(between) // 10], len(between)))
a, b = runs[len(runs) // 2]
t0 = S[a]
for k in range(a, min(a + 8, b)):
    print("%8.1f us  gap %5.1f  dur %6.1f  %s" % ((S[k] - t0) / 1e3, (S[k] - E[k - 1]) / 1e3 if k > a else 0, (E[k] - S[k]) / 1e3, nm[k][-60:]))
