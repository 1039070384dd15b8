#!/bin/bash
# rocprofv3 PMC passes over the kernel micro-bench (separate passes per counter group; gpurun forbids mixing --pmc
# with trace domains other than --kernel-trace).  Every pass has its own timeout: a TA_* pass once hung for the whole
# gpurun limit.  Results: gpurun_out/pmc_<tag>/pass*/ + summary.txt.  KBENCH_ARGS="16 3 256 320 32 32" profiles BASELINE config 4.
TAG=${1:-r01}
shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
GROUPS_DEFAULT=("FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum")
if [ $# -gt 0 ]; then GROUPS_DEFAULT=("$@"); fi
for grp in "${GROUPS_DEFAULT[@]}"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pass$i -o p -- python $R/scripts/kbench.py ${KBENCH_ARGS:-64 3} > $OUT/pass$i.log 2>&1 || echo "pass $i ($grp) failed or timed out" >> $OUT/failed.txt
done
python $R/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
