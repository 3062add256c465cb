"""The last kernels of one steady-state step from a rocprofv3 --kernel-trace CSV, both queues, times relative to the step's end (the Adam
launch): what the exposed tail of the backward consists of.   usage: step_tail.py <kernel_trace.csv> [n=40] [step_index_from_end=2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2


def nm(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("Grid_Size", "")) for r in rows)
adam = [i for i, e in enumerate(ev) if e[2].startswith("adam_kernel")]
hi = adam[-back]
t_adam = ev[hi][0]
for s, e, k, q, g in ev[hi - n:hi + 1]:
    print("q%-2s start %9.1f us  dur %7.1f us  end %9.1f   %s" % (q, (s - t_adam) / 1e3, (e - s) / 1e3, (e - t_adam) / 1e3, k))
