#!/usr/bin/env python3
"""Fold the passes of tools/pmc_all.sh into one JSON per (tag, rows): per kernel the mean counter values per dispatch
(over every launch of the process, and `timed_*`: over the launches of the bench line's timed epochs, epochs delimited by the
merge kernel), its mean duration from the kernel trace, and the derived figures DESIGN.md quotes:
  mfma_busy      SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)
  clock_ghz      GRBM_GUI_ACTIVE / 8 / duration   (MI355X_MICROARCH.md, DVFS give-back)
  valu_per_mfma  (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA
  fabric_bytes_corrected  2 * FETCH_SIZE(KB) * 1024 + WRITE_SIZE(KB) * 1024   (gfx950 counts a 128-B read as 64 B)
  l2_hit         TCC_HIT / (TCC_HIT + TCC_MISS)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out_dir, tag, rows = sys.argv[1], sys.argv[2], int(sys.argv[3])
extra = sys.argv[4:]


def demangle_somhip(name):
    """rocprofv3 leaves a kernel name mangled when its template arguments hold a 16-bit float type (DF16b = __bf16,
    DF16_ = _Float16): _ZN6somhip<len><name>I<args>E... -> name<ints, type>."""
    m = re.match(r"^_ZN6somhip(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    ident, rest = name[m.end():m.end() + n], name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        t = rest[1:]
        while t and not t.startswith("E"):
            mi = re.match(r"^L[ib](\d+)E", t)
            if mi:
                args.append(mi.group(1)); t = t[mi.end():]
            elif t.startswith("DF16b"):
                args.append("bf16"); t = t[5:]
            elif t.startswith("DF16_"):
                args.append("f16"); t = t[5:]
            else:
                break
    return ident + ("<" + ", ".join(args) + ">" if args else "")


def short(name):
    name = demangle_somhip(name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    name = re.sub(r"^somhip::", "", name)
    if "rocprim" in name or "ROCPRIM" in name:
        name = "rocprim::" + ("merge_sort" if "merge" in name else "radix_sort" if "radix" in name or "onesweep" in name else "other")
    return name


steps, warm = int(os.environ.get("PMC_STEPS", "20")), int(os.environ.get("PMC_WARMUP", "5"))


def by_epoch(rows_of_pass):
    """(order key, kernel, payload) of one profiler pass -> [(epoch, kernel, payload)]: an epoch ends with its merge kernel
    (one per epoch in every mode), so a dispatch belongs to the TIMED region of the bench line iff warm <= epoch < warm + steps
    whichever kernels the epoch happened to launch (the first epoch of an exact-mode run has no plan and no tile-list screen)."""
    out, ep = [], 0
    for _, k, payload in sorted(rows_of_pass, key=lambda r: r[0]):
        out.append((ep, k, payload))
        if k == "merge_kernel":
            ep += 1
    return out


counters = collections.defaultdict(lambda: collections.defaultdict(list))
counters_timed = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq_a", "sq_b", "fetch", "write", "tcc"):
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            per_dispatch.setdefault(int(r["Dispatch_Id"]), (short(r["Kernel_Name"]), []))[1].append((r["Counter_Name"], float(r["Counter_Value"])))
        for ep, k, vals in by_epoch([(i, k, v) for i, (k, v) in per_dispatch.items()]):
            for c, v in vals:
                counters[k][c].append(v)
                if warm <= ep < warm + steps:
                    counters_timed[k][c].append(v)
dur = {}
for f in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                 "max_us": float(r["MaxNs"]) / 1e3}
    stats_copy = os.path.join("gpurun_out", "%s_kernel_stats_rows%d.csv" % (tag, rows))
    open(stats_copy, "w").write(open(f).read())

# the kernels dispatch by dispatch (kernel trace): under block skipping a launch's duration depends on the epoch it serves, so
# the average over the bench line's TIMED epochs is what agrees with the line's avg_launch_ms, not the average over every
# launch of the process (the profiler's own --stats table, copied beside this file, averages over all of them)
timed = {}
for f in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows_of_pass = [(int(r["Start_Timestamp"]), short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                    for r in csv.DictReader(open(f))]
    per = collections.defaultdict(list)
    for ep, k, d in by_epoch(rows_of_pass):
        per[k].append((ep, d))
    with open(os.path.join("gpurun_out", "%s_dispatches_rows%d.csv" % (tag, rows)), "w") as g:
        g.write("kernel,epoch,duration_us,timed_region\n")
        for k, v in sorted(per.items()):
            if not (k.startswith("bmu_") or k.startswith("exact_") or k.startswith("runsum") or k.startswith("rs_") or k.startswith("leftmul")):
                continue
            for ep, d in v:
                g.write("%s,%d,%.1f,%d\n" % (k.replace(",", ";"), ep, d, int(warm <= ep < warm + steps)))
    for k, v in per.items():
        sel = [d for ep, d in v if warm <= ep < warm + steps]
        if sel:
            timed[k] = {"timed_dispatches": len(sel), "timed_avg_us": sum(sel) / len(sel), "timed_epochs_with_a_launch": len({ep for ep, _ in v if warm <= ep < warm + steps}),
                        "timed_us_per_epoch": sum(sel) / steps}

from xpysom_dask_amd import build as B  # noqa: E402
lib = os.environ.get("SOM_LIB_PATH")
res = {"note": "rocprofv3 --pmc passes (one counter group per pass) of `python3 bench.py --rows %d --steps %d --warmup %d "
               "--no-cpu-baseline --no-batch65536 --no-throughput-mode --no-modes %s`; per-dispatch means; durations from a separate --kernel-trace pass. "
               "FETCH_SIZE / WRITE_SIZE are KB; on gfx950 a wide coalesced read is tallied at half its bytes "
               "(MI355X_MICROARCH.md, HBM): fabric bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024; Infinity-Cache hits "
               "are included, so this bounds HBM traffic from above." % (rows, steps, warm, " ".join(extra)),
       "workload": "c3" if "--workload" not in extra else extra[extra.index("--workload") + 1],
       "rows_per_launch": rows,
       "precision": extra[extra.index("--precision") + 1] if "--precision" in extra else
                    {"c3": "exact", "c2": "f32", "c5": "exact"}["c3" if "--workload" not in extra else extra[extra.index("--workload") + 1]],
       "build": B.built_hash(lib) if lib else B.built_hash(), "library": os.path.basename(lib) if lib else "libsomhip.so"}
for k, cs in sorted(counters.items()):
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["dispatches"] = max(len(v) for v in cs.values())
    if k in dur:
        d["duration_us"] = dur[k]["avg_us"]
    if k in timed:
        d.update(timed[k])
    if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d and d["GRBM_GUI_ACTIVE"] > 0:
        d["mfma_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if "GRBM_GUI_ACTIVE" in d and k in dur and dur[k]["avg_us"] > 0:
        d["clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / (dur[k]["avg_us"] * 1e3)
    if d.get("SQ_INSTS_MFMA", 0) > 0:
        d["valu_per_mfma"] = (d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / d["SQ_INSTS_MFMA"]
    if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in d:
                d[c.lower() + "_share"] = d[c] / d["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        d["fabric_bytes_corrected"] = 2.0 * d.get("FETCH_SIZE", 0.0) * 1024.0 + d.get("WRITE_SIZE", 0.0) * 1024.0
    t = {c: sum(v) / len(v) for c, v in counters_timed.get(k, {}).items() if v}
    if "FETCH_SIZE" in t or "WRITE_SIZE" in t:
        d["timed_fabric_bytes_corrected"] = 2.0 * t.get("FETCH_SIZE", 0.0) * 1024.0 + t.get("WRITE_SIZE", 0.0) * 1024.0
    if t.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in t:
        d["timed_mfma_busy"] = t["SQ_VALU_MFMA_BUSY_CYCLES"] / (t["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0) > 0:
        d["timed_l2_hit"] = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
    if d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0) > 0:
        d["l2_hit"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    res[k] = d
path = os.path.join("gpurun_out", "%s_pmc_traffic_rows%d.json" % (tag, rows))
json.dump(res, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path)
for k in sorted(res):
    if isinstance(res[k], dict) and ("mfma_busy" in res[k] or "fabric_bytes_corrected" in res[k]):
        d = res[k]
        print("%-34s dur %9.1f us  mfma_busy %s clock %s valu/mfma %s fabric %s MB l2hit %s wait %s" % (
            k, d.get("duration_us", float("nan")), "%.3f" % d["mfma_busy"] if "mfma_busy" in d else "-",
            "%.2f" % d["clock_ghz"] if "clock_ghz" in d else "-", "%.2f" % d["valu_per_mfma"] if "valu_per_mfma" in d else "-",
            "%.1f" % (d["fabric_bytes_corrected"] / 1e6) if "fabric_bytes_corrected" in d else "-",
            "%.3f" % d["l2_hit"] if "l2_hit" in d else "-", "%.3f" % d["sq_wait_any_share"] if "sq_wait_any_share" in d else "-"))
