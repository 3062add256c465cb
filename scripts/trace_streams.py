"""Per-queue busy time and per-kernel totals of the steady-state part of a rocprofv3 --kernel-trace CSV."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
n = len(ev); ev = ev[n // 2:]                      # steady state
t0, t1 = ev[0][0], max(e[1] for e in ev)
frac = nsteps * 0.5
print("span %.3f ms (~%.1f steps -> %.3f ms/step)" % ((t1 - t0) / 1e6, frac, (t1 - t0) / 1e6 / frac))
byq = collections.defaultdict(list)
for s, e, k, q in ev: byq[q].append((s, e, k))
for q, lst in byq.items():
    busy = sum(e - s for s, e, _ in lst)
    print("queue %s: %d kernels, busy %.3f ms/step" % (q, len(lst), busy / 1e6 / frac))
    tot = collections.Counter()
    for s, e, k in lst: tot[k.split("(")[0].split("::")[-1][:44]] += e - s
    for k, v in tot.most_common(10): print("    %-46s %.3f ms/step" % (k, v / 1e6 / frac))
