"""Busy time vs gaps on the GPU timeline from a rocprofv3 --kernel-trace CSV (one process, one queue)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# take the last third of the trace (steady-state steps)
n = len(ev); ev = ev[2 * n // 3:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
gaps = []
for s, e, k in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, k)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
print("kernels %d span %.3f ms busy %.3f ms (%.1f%%) gaps %.3f ms, mean gap %.2f us, median %.2f us" % (
    len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, (span - busy) / max(1, len(gaps)) / 1e3,
    sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3))
big = collections.Counter()
for g, k in gaps:
    big[k.split("(")[0][-40:]] += g
for k, v in big.most_common(8):
    print("  gap before %-42s %.3f ms" % (k, v / 1e6))
