#!/bin/bash
# One gpurun call: bench line, batch-65536 line, rocprofv3 kernel stats of the bench command.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err &&
timeout -k 10 200 python3 bench.py --rows 65536 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_batch65536.json 2> gpurun_out/bench_batch65536.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
find gpurun_out/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats.csv \;
cat gpurun_out/bench_default.json gpurun_out/bench_batch65536.json
head -12 gpurun_out/kernel_stats.csv
