#!/bin/bash
# usage: tools/pmc_c5.sh <tag>   -- PMC passes over tools/c5_probe.py (tiled bf16 kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${TAG}_a -- python3 tools/c5_probe.py > gpurun_out/pmc_${TAG}_a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_${TAG}_b -- python3 tools/c5_probe.py > gpurun_out/pmc_${TAG}_b.log 2>&1 &&
# (a third pass with TCC_HIT/MISS/EA0_RDREQ + FETCH_SIZE aborted inside rocprofv3 on this workload -- signal 6 -- and then hung; not run)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_${TAG}_k -- python3 tools/c5_probe.py > gpurun_out/pmc_${TAG}_k.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_a gpurun_out/pmc_${TAG}_b
find gpurun_out/pmc_${TAG}_k -name "*kernel_stats.csv" -exec head -4 {} \;
