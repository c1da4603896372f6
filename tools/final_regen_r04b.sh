#!/bin/bash
# round 4, second call: PMC of the bf16 launch and of the configs[4] shard, then the bf16 / c5 / strong-scaling bench lines
set -e
TAG=r04
tools/pmc_all.sh ${TAG}_bf16 1048576 --precision bf16 > gpurun_out/${TAG}_pmc_c3_bf16.log 2>&1
PMC_STEPS=3 PMC_WARMUP=1 tools/pmc_all.sh ${TAG}_c5 250000 --workload c5 > gpurun_out/${TAG}_pmc_c5.log 2>&1
cp gpurun_out/${TAG}_bf16_pmc_traffic_rows1048576.json gpurun_out/${TAG}_c5_pmc_traffic_rows250000.json profiles/
cp gpurun_out/${TAG}_c5_kernel_stats_rows250000.csv profiles/
python3 bench.py --steps 20 --warmup 5 --precision bf16 --no-modes > gpurun_out/${TAG}_bench_c3_bf16.json
python3 bench.py --workload c5 > gpurun_out/${TAG}_bench_c5.json
python3 bench.py --scaling strong --no-modes --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_strong_n1.json
for w in c3_bf16 c5 strong_n1; do python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_$w.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$w", round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 2), "M/s frac", round(r["frac"], 4), "traffic", r.get("traffic"), (r.get("traffic_source") or {}).get("same_build"))
PY
done
