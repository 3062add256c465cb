"""Chronological dump of one steady-state step of a rocprofv3 --kernel-trace CSV: start (us from the previous adam_kernel's end), duration,
queue, gap to the previous kernel of the same queue, kernel.    usage: step_dump.py <kernel_trace.csv> [step_index_from_end=2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))


def nm(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0][:60]


back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows)
adam = [i for i, e in enumerate(ev) if e[2].startswith("adam_kernel")]
lo, hi = adam[-back - 1], adam[-back]
t0 = ev[lo][1]
last_end = {}
for s, e, k, q in ev[lo + 1:hi + 1]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print("%9.1f %7.1f q%-2s gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap, k))
