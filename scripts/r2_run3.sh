set -x
cd $GRAFT_REPO_ROOT
for x in 0 1 0 1; do SAGE_XCD_ORDER=$x python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r2e_bench_xcd$x.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2e_bench_xcd$x.json'));print('xcd',$x,round(d['ms_per_step'],4),round(d['roofline']['avg_launch_ms'],4),round(d['roofline']['geo_kernel']['avg_launch_ms'],4),d['roofline']['error_pass_ms'])"; done
python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r2e_parity.log 2>&1; tail -2 gpurun_out/r2e_parity.log
for v in w2; do SAGE_BA_LIB=$GRAFT_REPO_ROOT/sage_slam_amd/_variants/libsage_$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r2e_bench_$v.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2e_bench_$v.json'));print('variant $v',round(d['ms_per_step'],4),round(d['roofline']['avg_launch_ms'],4))"; done
cd /tmp && export TMPDIR=/tmp
bash $GRAFT_REPO_ROOT/scripts/pmc_run.sh r2e "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"
tail -60 $GRAFT_REPO_ROOT/gpurun_out/pmc_r2e/summary.txt
