"""Per-epoch kernel table from a rocprofv3 kernel trace of a training run (epochs are delimited by the merge kernel).

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/resid_probe.py
    python tools/epoch_trace.py OUT [epoch ...]        # default: every epoch, one line each + a table for the listed ones
"""
import csv
import glob
import os
import re
import sys
from collections import OrderedDict, defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"^somhip::", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", name)
    base = m.group(1) if m else name
    tmpl = m.group(2) or "" if m else ""
    if "rocprim" in name:
        return "rocprim::*"
    if base.startswith("_ZN6somhip"):
        base = re.sub(r"^_ZN6somhip\d+", "", base)
        base = re.sub(r"E[A-Z].*$", "", base)
    if len(tmpl) > 28:
        tmpl = tmpl[:28] + ".."
    return base + tmpl


def main():
    d = sys.argv[1]
    want = [int(v) for v in sys.argv[2:]]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    epochs, cur = [], []
    for s, e, n in rows:
        cur.append((s, e, n))
        if "merge_kernel" in n or "merge_prep" in n:
            epochs.append(cur)
            cur = []
    for i, ep in enumerate(epochs):
        tot = defaultdict(float)
        for s, e, n in ep:
            tot[short(n)] += (e - s) / 1e6
        busy = sum(tot.values())
        wall = (ep[-1][1] - ep[0][0]) / 1e6
        top = sorted(tot.items(), key=lambda kv: -kv[1])[:7]
        print("epoch %2d  wall %7.3f ms  kernels %7.3f ms  " % (i, wall, busy) + "  ".join("%s %.3f" % (k[:26], v) for k, v in top))
    for i in want:
        if i >= len(epochs):
            continue
        tot, cnt = OrderedDict(), defaultdict(int)
        for s, e, n in epochs[i]:
            k = short(n)
            tot[k] = tot.get(k, 0.0) + (e - s) / 1e6
            cnt[k] += 1
        print("\n--- epoch %d ---" % i)
        for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
            print("  %-60s x%-3d %8.3f ms" % (k, cnt[k], v))


if __name__ == "__main__":
    main()
