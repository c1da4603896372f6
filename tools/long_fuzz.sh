#!/bin/bash
export SOM_TEST_HOOKS=1   # (the library reads its developer switches only under this one)
# long fuzz run on the final build: every fuzzer, several seeds; one summary line each
out=gpurun_out/r03_fuzz_summary.txt
python - <<'PY' > $out
from xpysom_dask_amd import build as B
print("# long fuzz run, build", B.built_hash())
PY
run() { name=$1; shift; line=$(timeout -k 10 280 "$@" 2>/dev/null | tail -1); echo "$name: $line" | tee -a $out; }
for s in 101 102 103 104; do run "fuzz_exact seed $s (500 cases, resident screen, one round + seed)" python tests/fuzz/fuzz_exact.py $s 500; done
for s in 111 112; do FUZZ_WIDE=1 run "fuzz_exact FUZZ_WIDE=1 seed $s (300 cases, wide screen, two rounds)" python tests/fuzz/fuzz_exact.py $s 300; done
SOM_EXACT_TWO_ROUND=1 run "fuzz_exact SOM_EXACT_TWO_ROUND=1 seed 121 (500 cases)" python tests/fuzz/fuzz_exact.py 121 500
SOM_EXACT_TWO_ROUND=0 FUZZ_WIDE=1 run "fuzz_exact FUZZ_WIDE=1 SOM_EXACT_TWO_ROUND=0 seed 122 (200 cases)" python tests/fuzz/fuzz_exact.py 122 200
SOM_EXACT_PASS_ROWS=1024 run "fuzz_exact SOM_EXACT_PASS_ROWS=1024 seed 123 (300 cases, several passes)" python tests/fuzz/fuzz_exact.py 123 300
for s in 131 132 133 134; do SOM_EXACT_SKIP=2 run "fuzz_exact SOM_EXACT_SKIP=2 seed $s (500 cases, block skipping on every map)" python tests/fuzz/fuzz_exact.py $s 500; done
SOM_EXACT_SKIP=2 SOM_EXACT_PASS_ROWS=1024 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_PASS_ROWS=1024 seed 135 (500 cases)" python tests/fuzz/fuzz_exact.py 135 500
SOM_EXACT_SKIP=2 FUZZ_MAXSIDE=260 run "fuzz_exact SOM_EXACT_SKIP=2 FUZZ_MAXSIDE=260 seed 136 (400 cases)" python tests/fuzz/fuzz_exact.py 136 400
SOM_EXACT_SKIP=2 SOM_VERIFY=64 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_VERIFY=64 seed 137 (400 cases, canary on)" python tests/fuzz/fuzz_exact.py 137 400
SOM_EXACT_SKIP=0 run "fuzz_exact SOM_EXACT_SKIP=0 seed 138 (300 cases)" python tests/fuzz/fuzz_exact.py 138 300
for s in 201 202; do run "fuzz_shapes seed $s (600 cases)" python tests/fuzz/fuzz_shapes.py $s 600; done
run "fuzz_paths seed 301 (300 cases)" python tests/fuzz/fuzz_paths.py 301 300
run "fuzz_train seed 401 (300 cases)" python tests/fuzz/fuzz_train.py 401 300
run "fuzz_infer seed 501 (300 cases)" python tests/fuzz/fuzz_infer.py 501 300
