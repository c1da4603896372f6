#!/bin/bash
export SOM_TEST_HOOKS=1   # (the library reads its developer switches only under this one)
# round 5's fuzz run: the exact mode against float32 (IDENTICAL ids demanded) with the scout (queries, streamed chunks, first
# epochs: forced on every map by SOM_EXACT_SKIP=2), without it, the wide screen under a plan, stale orders, several passes, the
# canary on -- then the oracle-differential fuzzers.  One summary line each.
out=gpurun_out/r05_fuzz_summary.txt
python - <<'PY' > $out
from xpysom_dask_amd import build as B
print("# round 5 fuzz run, build", B.built_hash())
PY
run() { name=$1; shift; line=$(timeout -k 10 ${FUZZ_TIMEOUT:-240} "$@" 2>/dev/null | tail -1); echo "$name: $line" | tee -a $out; }
N=${FUZZ_CASES:-250}
for s in 241 242 243; do SOM_EXACT_SKIP=2 run "fuzz_exact SOM_EXACT_SKIP=2 seed $s ($N cases: scout + plan + sub-blocks + refinement on every map, queries and streamed chunks too)" python tests/fuzz/fuzz_exact.py $s $N; done
SOM_EXACT_SKIP=2 SOM_EXACT_SCOUT=0 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_SCOUT=0 seed 244 ($N cases: plans from last epoch's BMUs only)" python tests/fuzz/fuzz_exact.py 244 $N
SOM_EXACT_SKIP=2 SOM_EXACT_RESORT=1000 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_RESORT=1000 seed 245 ($N cases: the first order kept for good)" python tests/fuzz/fuzz_exact.py 245 $N
SOM_EXACT_SKIP=2 SOM_EXACT_RESORT=1 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_RESORT=1 seed 246 ($N cases: a sort in every planned epoch)" python tests/fuzz/fuzz_exact.py 246 $N
SOM_EXACT_SKIP=2 SOM_EXACT_SUBBLOCKS=0 SOM_EXACT_REFINE=0 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_SUBBLOCKS=0 SOM_EXACT_REFINE=0 seed 247 ($N cases)" python tests/fuzz/fuzz_exact.py 247 $N
SOM_EXACT_SKIP=2 SOM_EXACT_PASS_ROWS=1024 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_PASS_ROWS=1024 seed 248 ($N cases: several passes)" python tests/fuzz/fuzz_exact.py 248 $N
SOM_EXACT_SKIP=2 FUZZ_MAXSIDE=260 run "fuzz_exact SOM_EXACT_SKIP=2 FUZZ_MAXSIDE=260 seed 249 ($N cases)" python tests/fuzz/fuzz_exact.py 249 $N
SOM_EXACT_SKIP=2 SOM_VERIFY=64 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_VERIFY=64 seed 250 ($N cases, canary on)" python tests/fuzz/fuzz_exact.py 250 $N
run "fuzz_exact default switches seed 251 ($N cases)" python tests/fuzz/fuzz_exact.py 251 $N
SOM_EXACT_SKIP=2 SOM_EXACT_QUEUE=0 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_QUEUE=0 seed 255 ($N cases: the listed screen as one workgroup per tile)" python tests/fuzz/fuzz_exact.py 255 $N
SOM_EXACT_SKIP=2 SOM_EXACT_QUEUE=25 FUZZ_MAXSIDE=260 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_QUEUE=25 FUZZ_MAXSIDE=260 seed 256 ($N cases: work items a quarter of the mean list)" python tests/fuzz/fuzz_exact.py 256 $N
SOM_EXACT_SKIP=2 FUZZ_WIDE=1 run "fuzz_exact SOM_EXACT_SKIP=2 FUZZ_WIDE=1 seed 252 (120 cases, wide screen under a plan)" python tests/fuzz/fuzz_exact.py 252 120
SOM_EXACT_SKIP=2 FUZZ_WIDE=1 SOM_EXACT_PASS_ROWS=1024 SOM_EXACT_RESORT=1000 run "fuzz_exact SOM_EXACT_SKIP=2 FUZZ_WIDE=1 SOM_EXACT_PASS_ROWS=1024 SOM_EXACT_RESORT=1000 seed 253 (120 cases)" python tests/fuzz/fuzz_exact.py 253 120
FUZZ_WIDE=1 SOM_VERIFY=64 run "fuzz_exact FUZZ_WIDE=1 SOM_VERIFY=64 seed 254 (120 cases, default switches, canary on)" python tests/fuzz/fuzz_exact.py 254 120
SOM_EXACT_SKIP=2 FUZZ_WIDE=1 SOM_EXACT_QUEUE=0 run "fuzz_exact SOM_EXACT_SKIP=2 FUZZ_WIDE=1 SOM_EXACT_QUEUE=0 seed 257 (120 cases: the wide listed screen as one workgroup per tile and part)" python tests/fuzz/fuzz_exact.py 257 120
run "fuzz_shapes seed 205 ($N cases)" python tests/fuzz/fuzz_shapes.py 205 $N
run "fuzz_paths seed 305 (150 cases)" python tests/fuzz/fuzz_paths.py 305 150
run "fuzz_train seed 405 (150 cases)" python tests/fuzz/fuzz_train.py 405 150
run "fuzz_infer seed 505 (150 cases)" python tests/fuzz/fuzz_infer.py 505 150
