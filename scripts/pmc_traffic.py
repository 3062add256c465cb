"""Aggregate FETCH_SIZE / WRITE_SIZE (KB) per kernel family from rocprofv3 counter_collection CSVs -> JSON.
FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads; MI355X_MICROARCH.md, HBM/rocprofv3 section)."""
import csv, glob, json, os, sys, collections
root, out = sys.argv[1], sys.argv[2]
def collect(sub, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0].split("::")[-1].strip()
            acc[k][0] += float(r["Counter_Value"]) * 1024.0
            acc[k][1] += 1
    return acc
fe, wr = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
kern = {}
for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[0])):
    n = max(fe.get(k, [0, 0])[1], wr.get(k, [0, 0])[1])
    if n == 0: continue
    kern[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * fe.get(k, [0, 1])[0] / n, "write_bytes_per_launch": wr.get(k, [0, 1])[0] / n}
fam = [k for k in kern if k.startswith("conv_glds") or k.startswith("mdcv_conv3x3_shift") or k.startswith("conv_igemm")]
tot_f = sum(kern[k]["fetch_bytes_per_launch"] * kern[k]["launches"] for k in fam)
tot_w = sum(kern[k]["write_bytes_per_launch"] * kern[k]["launches"] for k in fam)
steps = 4        # --steps 2 --warmup 1 + the final loss step
json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, MDCV_WGRAD_STREAM=0) of `python bench.py --workload yolo --steps 2 "
                    "--warmup 1 --no-breakdown` (4 training steps); KB counters converted to bytes; FETCH_SIZE doubled (gfx950 reports half of wide "
                    "coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
           "steps": steps,
           "conv2d_family": {"kernels": fam, "fetch_bytes_per_step": tot_f / steps, "write_bytes_per_step": tot_w / steps},
           "kernels": kern}, open(out, "w"), indent=1)
print(json.dumps({"conv2d_family_bytes_per_step": (tot_f + tot_w) / steps, "kernels": len(kern)}))
