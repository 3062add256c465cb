"""rocprofv3 counter_collection CSVs of scripts/profile_round.sh -> <tag>_pmc_hbm_traffic.json, <tag>_pmc_mfma_busy.json.

HBM bytes: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read,
/opt/skills/guides/MI355X_MICROARCH.md "HBM"); WRITE_SIZE as reported.  Per kernel symbol (template arguments kept, so the keys
are the `kernel` strings of bench.py's roofline_kernels) and per launch.  Both files carry the kernel fingerprint of the tree they
were measured with (mdcv/_fingerprint.py); bench.py quotes `traffic` only when it matches the running tree."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import short_symbol                      # noqa: E402
from mdcv._fingerprint import kernel_fingerprint    # noqa: E402

tmp, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
wl = sys.argv[4] if len(sys.argv) > 4 else "yolo"
pre = tag if wl == "yolo" else f"{tag}_{wl}"               # r05_pmc_hbm_traffic.json (yolo: the name bench.py has always looked for) / r05_rektnet_pmc_...
# STEADY-STATE steps only (round 6): the first step of a process also pays one-off initialisation copies / fills (parameter flattening, plan buffers:
# ~2.5 GB that rounds 2-5 divided into "per step").  A step ends with the optimizer's `adam_kernel`; records up to and including the FIRST one are
# dropped, and the window closes with the LAST one: STEPS = adam launches - 1.
STEPS = None


def steady(rows):
    """rows of ONE process in dispatch order -> (rows of the steady-state window, steps in it)"""
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    marks = [int(r["Dispatch_Id"]) for r in rows if "adam_kernel" in r["Kernel_Name"]]
    marks = sorted(set(marks))
    if len(marks) < 2:
        raise SystemExit("pmc_round.py: fewer than two optimizer steps in the trace -- run bench.py with --steps >= 2")
    lo, hi = marks[0], marks[-1]
    return [r for r in rows if lo < int(r["Dispatch_Id"]) <= hi], len(marks) - 1


def collect(sub):
    global STEPS
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(tmp, sub, "**", "*counter_collection.csv"), recursive=True):
        rows, n = steady(list(csv.DictReader(open(f))))
        STEPS = n if STEPS is None else min(STEPS, n)
        for r in rows:
            a = acc[short_symbol(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


def durations(sub):
    d = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(tmp, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = d[short_symbol(r["Kernel_Name"])]
            a[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            a[1] += 1
    return d


fp = kernel_fingerprint()
fe, wr = collect("fetch"), collect("write")
kern = {}
for k in set(fe) | set(wr):
    f, w = fe.get(k, {}).get("FETCH_SIZE", [0.0, 0]), wr.get(k, {}).get("WRITE_SIZE", [0.0, 0])
    n = max(f[1], w[1])
    if n:
        kern[k] = {"launches_per_step": n / STEPS, "fetch_bytes_per_launch": 2.0 * 1024.0 * f[0] / max(f[1], 1),
                   "write_bytes_per_launch": 1024.0 * w[0] / max(w[1], 1)}
tot = {k: (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches_per_step"] for k, v in kern.items()}
json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only, MDCV_WGRAD_STREAM=0, of `python bench.py --workload "
                    + wl + " --steps 4 --warmup 1 --no-breakdown --no-ref-loop --no-classes1`; STEADY-STATE steps only (the records up to the first adam_kernel -- one-off initialisation copies / fills -- are dropped; `steps` = optimizer steps in the window); KB -> bytes; FETCH_SIZE doubled (gfx950), WRITE_SIZE as reported",
           "fingerprint": fp, "tag": tag, "workload": wl, "steps": STEPS, "total_bytes_per_step": sum(tot.values()),
           "bytes_per_step_by_kernel": dict(sorted(((k, round(v)) for k, v in tot.items()), key=lambda kv: -kv[1])),
           "kernels": dict(sorted(kern.items(), key=lambda kv: -tot[kv[0]]))},
          open(os.path.join(out, f"{pre}_pmc_hbm_traffic.json"), "w"), indent=1)

mf, du = collect("mfma"), durations("mfma")
rows = {}
for k, c in mf.items():
    busy, sq, grbm = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0]), c.get("SQ_BUSY_CYCLES", [0, 0]), c.get("GRBM_GUI_ACTIVE", [0, 0])
    n = max(busy[1], 1)
    ns = du[k][0] / max(du[k][1], 1)
    rows[k] = {"launches": busy[1], "avg_ns_under_pmc": ns, "SQ_VALU_MFMA_BUSY_CYCLES": busy[0] / n, "SQ_BUSY_CYCLES": sq[0] / max(sq[1], 1),
               "GRBM_GUI_ACTIVE": grbm[0] / max(grbm[1], 1),
               # fraction of the chip's SIMD-cycles (256 CUs x 4 SIMDs at the 2.4 GHz peak clock, over the kernel's own duration) with the MFMA pipe busy
               "mfma_busy_frac_of_simd_cycles": (busy[0] / n) / (ns * 2.4 * 1024) if ns else None,
               "mfma_busy_over_4x_sq_busy": (busy[0] / n) / (4.0 * sq[0] / max(sq[1], 1)) if sq[0] else None}
json.dump({"_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass, --kernel-trace only, MDCV_WGRAD_STREAM=0) of the same short "
                    "bench command; per-launch averages per kernel symbol",
           "fingerprint": fp, "tag": tag, "workload": wl,
           "kernels": dict(sorted(rows.items(), key=lambda kv: -(kv[1]["avg_ns_under_pmc"] * kv[1]["launches"])))},
          open(os.path.join(out, f"{pre}_pmc_mfma_busy.json"), "w"), indent=1)
print(json.dumps({"fingerprint": fp, "total_hbm_GB_per_step": sum(tot.values()) / 1e9, "kernels": len(kern)}))
