"""Average PMC counters per launch for the conv kernels in rocprofv3 counter_collection CSVs."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv" not in k and "wgrad" not in k: continue
        key = (k.split("<")[0].split("::")[-1][:28], r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:30s} {c:34s} {v / n:16.1f}  (n={n})")
