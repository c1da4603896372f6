"""Kernel table of a rocprofv3 kernel trace between two marks: every launch in time order, or summed by kernel.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/query_bench.py
    python tools/kernel_table.py OUT [--last N]      # the last N launches summed by kernel (default: all)"""
import csv, glob, os, sys
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from epoch_trace import short

d = sys.argv[1]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if last:
    rows = rows[-last:]
tot, cnt = OrderedDict(), {}
for s, e, n in rows:
    k = short(n)
    tot[k] = tot.get(k, 0.0) + (e - s) / 1e6
    cnt[k] = cnt.get(k, 0) + 1
print("%d launches, wall %.3f ms, kernels %.3f ms" % (len(rows), (rows[-1][1] - rows[0][0]) / 1e6, sum(tot.values())))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %-64s x%-4d %9.3f ms" % (k, cnt[k], v))
