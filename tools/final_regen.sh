#!/bin/bash
# usage (one gpurun call, from the repo root): tools/final_regen.sh <tag>
# PMC summaries of the three profiled launches on THIS build, copied into profiles/ on the box so that the bench lines
# that follow cite a summary of their own build (traffic_source.same_build), then the bench lines themselves.
set -e
TAG=$1
tools/pmc_all.sh $TAG 1048576 > gpurun_out/${TAG}_pmc_c3.log 2>&1
tools/pmc_all.sh $TAG 65536 > gpurun_out/${TAG}_pmc_b64k.log 2>&1
tools/pmc_all.sh ${TAG}_bf16 1048576 --precision bf16 > gpurun_out/${TAG}_pmc_c3_bf16.log 2>&1
tools/pmc_all.sh ${TAG}_c5 250000 --workload c5 > gpurun_out/${TAG}_pmc_c5.log 2>&1
cp gpurun_out/${TAG}_pmc_traffic_rows1048576.json gpurun_out/${TAG}_pmc_traffic_rows65536.json profiles/
cp gpurun_out/${TAG}_bf16_pmc_traffic_rows1048576.json gpurun_out/${TAG}_c5_pmc_traffic_rows250000.json profiles/
cp gpurun_out/${TAG}_kernel_stats_rows1048576.csv gpurun_out/${TAG}_kernel_stats_rows65536.csv gpurun_out/${TAG}_c5_kernel_stats_rows250000.csv profiles/
cp gpurun_out/${TAG}_dispatches_rows1048576.csv gpurun_out/${TAG}_dispatches_rows65536.csv profiles/
python3 bench.py > gpurun_out/${TAG}_bench_c3.json
python3 bench.py --precision bf16 --no-modes > gpurun_out/${TAG}_bench_c3_bf16.json
python3 bench.py --workload c5 > gpurun_out/${TAG}_bench_c5.json
python3 bench.py --workload c2 > gpurun_out/${TAG}_bench_c2.json
python3 bench.py --scaling strong --no-modes > gpurun_out/${TAG}_bench_strong_n1.json
for w in c3 c3_bf16 c5 c2 strong_n1; do python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_$w.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$w", round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 2), "M/s frac", round(r["frac"], 4), "traffic", r.get("traffic"), (r.get("traffic_source") or {}).get("same_build"), (r.get("batch65536") or {}).get("epoch_ms"))
PY
done
