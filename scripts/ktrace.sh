#!/bin/bash
# per-kernel average duration of a command: ktrace.sh <script.py> args...
cd /tmp && export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/ktrace
rm -rf $OUT; mkdir -p $OUT
S=$1; shift
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/scripts/$S "$@" > /dev/null 2>&1 || echo "failed"
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-60s calls %5s avg %8.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
