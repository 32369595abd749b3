"""Post-processor of tools/env_counters.sh: per-launch averages of every counter for the env-only leg's kernel
(flex_step_many_kernel: `steps_per_launch` steps per launch; flex_step_kernel with --launch-form single: one) ->
gpurun_out/<tag>_pmc_traffic.json, stamped with the digest of the sources the benched library was built from."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, envs = sys.argv[1], int(sys.argv[2])
spl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
form = sys.argv[4] if len(sys.argv) > 4 else "single"
prefix = "void flex_step_many_kernel" if form == "many" else "void flex_step_kernel"
if form != "many":
    spl = 1
agg = collections.defaultdict(list)
names = set()
for fn in glob.glob(os.path.join(ROOT, "gpurun_out", f"ctr_{tag}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Kernel_Name"].startswith(prefix):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            names.add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
c = {k: sum(v) / len(v) for k, v in agg.items()}
n = {k: len(v) for k, v in agg.items()}
import safe_marl_amd  # noqa: F401,E402
from safe_marl_amd import build  # noqa: E402
out = {"source_digest": build.built_digest(), "envs_per_launch": envs, "steps_per_launch": spl, "kernel": sorted(names), "launches_averaged": n,
       "counters_per_launch": {k: round(v, 1) for k, v in sorted(c.items())}}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # KB units; gfx950 counts half of the fetched bytes (MI355X_MICROARCH.md §HBM; profiles/r01_fetch_calibration.txt)
    out["flex_step_kernel_bytes_per_launch"] = int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    out["bytes_per_env_step"] = out["flex_step_kernel_bytes_per_launch"] / float(spl * envs)
    out["source"] = (f"tools/env_counters.sh {tag} {envs}: (2 x FETCH_SIZE {c['FETCH_SIZE']:.0f} KB + WRITE_SIZE {c['WRITE_SIZE']:.0f} KB) x 1024 per launch of {spl} step(s); "
                     "separate --pmc passes, eager launches of bench.py's env-only leg")
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
