#!/bin/bash
# round 5, second call: the wide workloads (c5: configs[4]'s shard; c5e: the same map with euclidean + gaussian, where block skipping
# beyond 128 features engages), the strong-scaling N = 1 leg, the one-rank RCCL leg (the collective's latency floor), the four-rank
# one-card rehearsal with one structureless shard, the query path and the schedule traces of the data variants
set -e
TAG=r05
python3 bench.py --workload c5 --no-modes > gpurun_out/${TAG}_bench_c5.json
python3 bench.py --workload c5e --steps 9 --warmup 3 > gpurun_out/${TAG}_bench_c5e.json
python3 bench.py --scaling strong --no-modes --no-variants > gpurun_out/${TAG}_bench_strong_n1.json
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 SOM_FORCE_ALLREDUCE=1 python3 bench.py --no-modes --no-variants --no-schedule --no-cpu-baseline --no-batch65536 --no-throughput-mode > gpurun_out/${TAG}_allreduce_floor.json
SOM_DIST_BACKEND=gloo python3 bench.py --gpus 4 --steps 6 --warmup 3 --rows 65536 --no-cpu-baseline --no-throughput-mode --no-batch65536 --no-modes --unstructured-ranks 1 > gpurun_out/${TAG}_rank_spread.json
python3 tools/query_bench.py > gpurun_out/${TAG}_query_bench.txt 2>&1
for v in blobs heavy overlap manifold normal; do
  echo "== $v: planned (default switches), then SOM_EXACT_SKIP=0" >> gpurun_out/${TAG}_schedule_variants.txt
  python3 tools/schedule_trace.py $v 2>/dev/null | grep -E "^epoch|whole" | awk '{printf "%s ", $5} END {print ""}' >> gpurun_out/${TAG}_schedule_variants.txt
  python3 tools/schedule_trace.py $v 2>/dev/null | tail -1 >> gpurun_out/${TAG}_schedule_variants.txt
  SOM_EXACT_SKIP=0 python3 tools/schedule_trace.py $v 2>/dev/null | grep -E "^epoch|whole" | awk '{printf "%s ", $5} END {print ""}' >> gpurun_out/${TAG}_schedule_variants.txt
done
for w in c5 c5e strong_n1; do python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_$w.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$w", round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 2), "M/s frac", round(r["frac"], 4), r.get("frac_full_scan"))
PY
done
