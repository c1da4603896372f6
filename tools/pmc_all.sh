#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_all.sh <tag> <rows> [extra bench args]
# rocprofv3 passes of `python3 bench.py --rows <rows> ...`: kernel trace + stats, two SQ counter groups, FETCH_SIZE,
# WRITE_SIZE, TCC hit/miss -- one group per pass (gpurun refuses --pmc combined with trace domains), then
# tools/pmc_report.py folds them into gpurun_out/<tag>_pmc_rows<rows>.json (+ the kernel stats csv).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; ROWS=$2; shift 2
# (the bench line's own steps and warm-up: under block skipping a launch's duration depends on the epoch it serves)
STEPS=${PMC_STEPS:-20}; WARM=${PMC_WARMUP:-5}
ARGS="--rows $ROWS --steps $STEPS --warmup $WARM --no-cpu-baseline --no-batch65536 --no-throughput-mode --no-modes --no-schedule $@"
OUT=gpurun_out/pmc_${TAG}_${ROWS}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq_a -- python3 bench.py $ARGS > $OUT/sq_a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq_b -- python3 bench.py $ARGS > $OUT/sq_b.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/tcc -- python3 bench.py $ARGS > $OUT/tcc.log 2>&1 &&
PMC_STEPS=$STEPS PMC_WARMUP=$WARM python3 tools/pmc_report.py $OUT $TAG $ROWS "$@"
