#!/bin/bash
# round 5, first call: the PMC summaries of the c3 launches on THIS build (copied into profiles/ on the box so that the bench
# lines that follow cite a summary of their own build), then the c3 / c2 / c3_bf16 bench lines (the driver's own flags for c3)
set -e
TAG=r05
tools/pmc_all.sh $TAG 1048576 > gpurun_out/${TAG}_pmc_c3.log 2>&1
cp gpurun_out/${TAG}_pmc_traffic_rows1048576.json gpurun_out/${TAG}_kernel_stats_rows1048576.csv gpurun_out/${TAG}_dispatches_rows1048576.csv profiles/
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c3.json
python3 bench.py --workload c2 > gpurun_out/${TAG}_bench_c2.json
python3 bench.py --precision bf16 --no-modes > gpurun_out/${TAG}_bench_c3_bf16.json
for w in c3 c2 c3_bf16; do python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_$w.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$w", round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 2), "M/s frac", round(r["frac"], 4), r.get("frac_full_scan"), "traffic", r.get("traffic"), (r.get("traffic_source") or {}).get("same_build"), (r.get("batch65536") or {}).get("epoch_ms"), (d.get("whole_schedule") or {}).get("ms_per_epoch"))
PY
done
