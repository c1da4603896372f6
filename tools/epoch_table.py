"""The per-epoch kernel table of DESIGN.md 3.0 from a rocprofv3 kernel trace of a training run (epochs end with the merge kernel).
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/schedule_trace.py
    python tools/epoch_table.py OUT [epoch ...]          # markdown rows for the listed epochs + the mean of epochs 5.."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from epoch_trace import short

COLS = [("scout", ("bmu_bf16_k16_kernel<4, F16, false", "exact_scout", "exact_groupkey", "exact_sample_tiles", "bmu_finalize")),
        ("sort+gather", ("exact_sortkey", "exact_gather_sorted", "SORT")),
        ("plan L1", ("exact_plan_kernel<4, F16, false",)),
        ("L2", ("exact_plan_kernel<4, F16, true",)),
        ("screen", ("bmu_bf16_k16_kernel<4, F16, true",)),
        ("select", ("exact_select_kernel",)),
        ("refine+select2", ("exact_refine", "exact_select2")),
        ("re-score", ("exact_rescore",))]
UPDATE = ("runsum", "leftmul", "strided_gemm", "neigh_tables", "merge_kernel", "band_ranges", "cs_", "UPD_SORT")

d = sys.argv[1]
want = [int(v) for v in sys.argv[2:]] or [0, 1, 2, 3, 5, 8, 12, 15, 18, 21, 24]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
epochs, cur = [], []
for s, e, n in rows:
    cur.append((s, e, short(n)))
    if "merge_kernel" in n or "merge_prep" in n:
        epochs.append(cur)
        cur = []


def line(ep):
    # (the radix sort serves the resident order -- before the screen -- and the update -- after the finalize kernel)
    fin = max((i for i, (_, _, k) in enumerate(ep) if k.startswith("exact_finalize")), default=len(ep))
    tot = dict((c, 0.0) for c, _ in COLS)
    upd = bmu = 0.0
    for i, (s, e, k) in enumerate(ep):
        ms = (e - s) / 1e6
        kk = k
        if k.startswith("rs_"):
            kk = "SORT" if i < fin else "UPD_SORT"
        hit = False
        for c, pats in COLS:
            if any(kk.startswith(p) for p in pats):
                tot[c] += ms; hit = True
                break
        if any(kk.startswith(p) for p in UPDATE):
            upd += ms
        else:
            bmu += ms
    return tot, bmu, bmu + upd


print("| epoch | " + " | ".join(c for c, _ in COLS) + " | BMU search + preparation | epoch (kernels) |")
print("|---|" + "---|" * (len(COLS) + 2))
acc = None
n_acc = 0
for i, ep in enumerate(epochs):
    tot, bmu, allk = line(ep)
    if i >= 5:
        acc = [a + b for a, b in zip(acc, list(tot.values()) + [bmu, allk])] if acc else list(tot.values()) + [bmu, allk]
        n_acc += 1
    if i in want:
        print("| %d | " % i + " | ".join(("%.2f" % tot[c]) if tot[c] >= 0.005 else "–" for c, _ in COLS) + " | %.2f | %.2f |" % (bmu, allk))
if acc:
    print("| mean 5…%d | " % (len(epochs) - 1) + " | ".join("%.2f" % (v / n_acc) for v in acc[:-2]) + " | %.2f | %.2f |" % (acc[-2] / n_acc, acc[-1] / n_acc))
