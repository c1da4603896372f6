#!/bin/bash
# One gpurun call: the three bench lines, the batch-65536 line, and rocprofv3 kernel stats of the default bench command.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err &&
timeout -k 10 400 python3 bench.py --workload c5 --steps 5 --warmup 1 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err &&
timeout -k 10 200 python3 bench.py --workload c2 --steps 50 --warmup 5 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err &&
timeout -k 10 200 python3 bench.py --rows 65536 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_batch65536.json 2> gpurun_out/bench_batch65536.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_c5.log 2>&1
find gpurun_out/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_c3.csv \;
find gpurun_out/prof_c5 -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_c5.csv \;
python3 - <<'PY'
import json
for w in ("c3","c5","c2","batch65536"):
    d=json.load(open(f"gpurun_out/bench_{w}.json"))
    print(w, round(d["value"]), "samples/s", round(d["ms_per_step"],3), "ms", "frac", round(d["roofline"]["frac"],3), "bmu ms", round(d["roofline"]["avg_launch_ms"],3))
PY
head -5 gpurun_out/kernel_stats_c5.csv | cut -c1-160
