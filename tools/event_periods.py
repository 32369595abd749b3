"""Period of the value sub-updates inside one update event, from tools/event_sequence.py --full listings: the time from one
sub-update's optimiser step (clip_rmsprop) to the next one's, i.e. a whole sub-update including whatever sits between two of
them (hand-over between graph launches, the batch refresh).  usage: tools/event_periods.py <label>=<sequence file> ..."""
import sys
for arg in sys.argv[1:]:
    label, fn = arg.rsplit("=", 1)
    t, span = [], ""
    for ln in open(fn):
        if ln.startswith("span "):
            span = ln.strip()
        f = ln.split()
        if len(f) > 6 and f[1] == "us" and "clip_rmsprop_kernel" in ln and "%" not in ln:
            t.append(float(f[0]))
    v = t[:10]                                        # ten value sub-updates, then the policy one
    per = [b - a for a, b in zip(v, v[1:])]
    print(f"{label}: event {span}")
    print("   value sub-update periods (us): " + " ".join(f"{p:.1f}" for p in per) + f"   mean {sum(per) / len(per):.1f}")
