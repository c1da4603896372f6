#!/bin/bash
export SOM_TEST_HOOKS=1   # (the library reads its developer switches only under this one)
# round 4's fuzz run: the exact mode against float32 (IDENTICAL ids demanded) on the resident sorted pass, the two-level plan,
# the refinement pass -- forced on every map (SOM_EXACT_SKIP=2), with stale orders, without sub-blocks / refinement, in
# several passes, with the canary on -- then the oracle-differential fuzzers.  One summary line each.
out=gpurun_out/r04_fuzz_summary.txt
python - <<'PY' > $out
from xpysom_dask_amd import build as B
print("# round 4 fuzz run, build", B.built_hash())
PY
run() { name=$1; shift; line=$(timeout -k 10 ${FUZZ_TIMEOUT:-170} "$@" 2>/dev/null | tail -1); echo "$name: $line" | tee -a $out; }
N=${FUZZ_CASES:-250}
for s in 141 142 143; do SOM_EXACT_SKIP=2 run "fuzz_exact SOM_EXACT_SKIP=2 seed $s ($N cases: plan + sub-blocks + refinement on every map)" python tests/fuzz/fuzz_exact.py $s $N; done
SOM_EXACT_SKIP=2 SOM_EXACT_RESORT=1000 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_RESORT=1000 seed 144 ($N cases: the order of the first plan kept for good)" python tests/fuzz/fuzz_exact.py 144 $N
SOM_EXACT_SKIP=2 SOM_EXACT_RESORT=1 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_RESORT=1 seed 145 ($N cases: a sort in every planned epoch)" python tests/fuzz/fuzz_exact.py 145 $N
SOM_EXACT_SKIP=2 SOM_EXACT_SUBBLOCKS=0 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_SUBBLOCKS=0 seed 146 ($N cases)" python tests/fuzz/fuzz_exact.py 146 $N
SOM_EXACT_SKIP=2 SOM_EXACT_REFINE=0 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_REFINE=0 seed 147 ($N cases)" python tests/fuzz/fuzz_exact.py 147 $N
SOM_EXACT_SKIP=2 SOM_EXACT_PASS_ROWS=1024 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_PASS_ROWS=1024 seed 148 ($N cases: several passes)" python tests/fuzz/fuzz_exact.py 148 $N
SOM_EXACT_SKIP=2 FUZZ_MAXSIDE=260 run "fuzz_exact SOM_EXACT_SKIP=2 FUZZ_MAXSIDE=260 seed 149 ($N cases)" python tests/fuzz/fuzz_exact.py 149 $N
SOM_EXACT_SKIP=2 SOM_VERIFY=64 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_VERIFY=64 seed 150 ($N cases, canary on)" python tests/fuzz/fuzz_exact.py 150 $N
SOM_EXACT_SKIP=2 SOM_EXACT_SUB44=0 run "fuzz_exact SOM_EXACT_SKIP=2 SOM_EXACT_SUB44=0 seed 153 ($N cases: 2 x 8 sub-blocks, units ascending in every group)" python tests/fuzz/fuzz_exact.py 153 $N
run "fuzz_exact default switches seed 151 ($N cases)" python tests/fuzz/fuzz_exact.py 151 $N
FUZZ_WIDE=1 run "fuzz_exact FUZZ_WIDE=1 seed 152 (120 cases, wide screen)" python tests/fuzz/fuzz_exact.py 152 120
run "fuzz_shapes seed 203 ($N cases)" python tests/fuzz/fuzz_shapes.py 203 $N
run "fuzz_paths seed 302 (150 cases)" python tests/fuzz/fuzz_paths.py 302 150
run "fuzz_train seed 402 (150 cases)" python tests/fuzz/fuzz_train.py 402 150
run "fuzz_infer seed 502 (150 cases)" python tests/fuzz/fuzz_infer.py 502 150
