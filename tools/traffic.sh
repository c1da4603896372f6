#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py $ARGS > gpurun_out/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py $ARGS > gpurun_out/pmc_write.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/pmc_tcc -- python3 bench.py $ARGS > gpurun_out/pmc_tcc.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_fetch","gpurun_out/pmc_write","gpurun_out/pmc_tcc"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-36:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in acc:
            for c, v in acc[k].items():
                print("%-38s %-22s n=%d mean=%.1f" % (k, c, len(v), sum(v)/len(v)))
PY
