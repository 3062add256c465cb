"""Per (kernel, grid) average duration from a rocprofv3 kernel-trace CSV (steady-state half), sorted by total time."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2])
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"], r["Queue_Id"]) for r in rows))
ev = ev[len(ev) // 2:]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k, g, w, q in ev:
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    a = agg[(q, name, int(g) // max(1, int(w)))]
    a[0] += 1; a[1] += e - s
for (q, name, blocks), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("q%s %-72s blocks %6d  n/step %5.1f  avg %7.1f us  total %.3f ms/step" % (q, name, blocks, n / nsteps, t / n / 1e3, t / 1e6 / nsteps))
