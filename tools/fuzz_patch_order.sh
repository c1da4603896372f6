#!/bin/bash
export SOM_TEST_HOOKS=1   # (the library reads its developer switches only under this one)
out=gpurun_out/r03_fuzz_patch_order.txt   # (kept as profiles/r03_fuzz_patch_order.txt)
python - <<'PY' > $out
from xpysom_dask_amd import build as B
print("# extended exact-vs-float32 fuzz of the patch-order build", B.built_hash())
PY
run() { name=$1; shift; line=$(timeout -k 10 280 "$@" 2>/dev/null | tail -1); echo "$name: $line" | tee -a $out; }
for s in 601 602 603 604 605 606; do run "fuzz_exact seed $s (1000 cases)" python tests/fuzz/fuzz_exact.py $s 1000; done
for s in 611 612 613; do FUZZ_MAXSIDE=260 run "fuzz_exact FUZZ_MAXSIDE=260 seed $s (300 cases)" python tests/fuzz/fuzz_exact.py $s 300; done
for s in 621 622 623 624; do FUZZ_WIDE=1 run "fuzz_exact FUZZ_WIDE=1 seed $s (300 cases)" python tests/fuzz/fuzz_exact.py $s 300; done
SOM_EXACT_PASS_ROWS=1024 run "fuzz_exact SOM_EXACT_PASS_ROWS=1024 seed 631 (600 cases)" python tests/fuzz/fuzz_exact.py 631 600
SOM_EXACT_PATCH=0 run "fuzz_exact SOM_EXACT_PATCH=0 seed 641 (500 cases)" python tests/fuzz/fuzz_exact.py 641 500
SOM_VERIFY=64 run "fuzz_exact SOM_VERIFY=64 seed 651 (500 cases)" python tests/fuzz/fuzz_exact.py 651 500
