#!/bin/bash
# usage (GPU box): tools/trace_exact.sh <tag> : rocprofv3 kernel trace of tools/prof_exact.py under the current EX_* env
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
OUT=gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/prof_exact.py > $OUT/run.log 2>&1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %5s  avg %10.1f us  total %9.3f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
