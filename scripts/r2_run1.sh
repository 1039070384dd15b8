set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_rccl.py tests/test_gpu_tracker.py -q -x -m gpu > gpurun_out/r2c_tests.log 2>&1; tail -3 gpurun_out/r2c_tests.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded_lm.py -q -m gpu > gpurun_out/r2c_parity.log 2>&1; tail -3 gpurun_out/r2c_parity.log
for t in 8 2 1; do SAGE_PHOTO_TPB=$t python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r2c_bench_tpb$t.json 2>gpurun_out/r2c_bench_tpb$t.err; python -c "
import json;d=json.load(open('gpurun_out/r2c_bench_tpb$t.json'));print('tpb',$t,d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['geo_kernel']['avg_launch_ms'],d['roofline']['error_pass_ms'])"; done
SAGE_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2c_bench_rccl1.json 2>gpurun_out/r2c_bench_rccl1.err; python -c "
import json;d=json.load(open('gpurun_out/r2c_bench_rccl1.json'));print('rccl 1-rank',d['ms_per_step'],d['config']['collective'])"; tail -3 gpurun_out/r2c_bench_rccl1.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_gfx950.txt 2>&1; wc -l $GRAFT_REPO_ROOT/gpurun_out/counters_gfx950.txt
