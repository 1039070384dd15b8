set -x
cd $GRAFT_REPO_ROOT
for x in 0 1 2 0 1 2; do SAGE_WORK_ORDER=$x python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r2f_bench_order$x.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2f_bench_order$x.json'));print('order',$x,round(d['ms_per_step'],4),round(d['roofline']['avg_launch_ms'],4),round(d['roofline']['geo_kernel']['avg_launch_ms'],4),d['roofline']['error_pass_ms'])"; done
SAGE_WORK_ORDER=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded_lm.py -q -m gpu -x > gpurun_out/r2f_parity_o1.log 2>&1; tail -2 gpurun_out/r2f_parity_o1.log
SAGE_WORK_ORDER=2 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r2f_parity_o2.log 2>&1; tail -2 gpurun_out/r2f_parity_o2.log
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do SAGE_WORK_ORDER=$x bash $GRAFT_REPO_ROOT/scripts/pmc_run.sh r2f_o$x "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE FETCH_SIZE"; grep -A 8 "^photo_kernel<32, 16, true, 1>" $GRAFT_REPO_ROOT/gpurun_out/pmc_r2f_o$x/summary.txt; done
