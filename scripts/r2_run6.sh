set -x
cd $GRAFT_REPO_ROOT
./scripts/micro/l1_cost > gpurun_out/r2h_l1_cost.txt 2>&1; cat gpurun_out/r2h_l1_cost.txt
for v in tl1 tl2 tl1 tl2; do SAGE_PHOTO_FLUSH=8 SAGE_BA_LIB=$GRAFT_REPO_ROOT/sage_slam_amd/_variants/libsage_$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r2h_bench_$v.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2h_bench_$v.json'));print('variant $v flush8',round(d['ms_per_step'],4),round(d['roofline']['avg_launch_ms'],4))"; done
SAGE_PHOTO_FLUSH=8 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r2h_bench_base8.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2h_bench_base8.json'));print('base flush8',round(d['ms_per_step'],4),round(d['roofline']['avg_launch_ms'],4))"
SAGE_BA_LIB=$GRAFT_REPO_ROOT/sage_slam_amd/_variants/libsage_tl1.so python scripts/tpb_noise_probe.py 16 > gpurun_out/r2h_noise_tl1.log 2>&1; tail -7 gpurun_out/r2h_noise_tl1.log
SAGE_PHOTO_FLUSH=8 SAGE_BA_LIB=$GRAFT_REPO_ROOT/sage_slam_amd/_variants/libsage_tl1.so python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r2h_parity_tl1.log 2>&1; tail -2 gpurun_out/r2h_parity_tl1.log
