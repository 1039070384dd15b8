set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded_lm.py -q -m gpu -x > gpurun_out/r2g_parity.log 2>&1; tail -2 gpurun_out/r2g_parity.log
SAGE_PIPELINE=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "pipelined or window" > gpurun_out/r2g_parity_pipe.log 2>&1; tail -2 gpurun_out/r2g_parity_pipe.log
for f in 8 2 1 8 2 1; do SAGE_PHOTO_FLUSH=$f python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r2g_bench_flush$f.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2g_bench_flush$f.json'));print('flush',$f,round(d['ms_per_step'],4),round(d['roofline']['avg_launch_ms'],4),round(d['roofline']['geo_kernel']['avg_launch_ms'],4),d['roofline']['error_pass_ms'])"; done
python scripts/tpb_noise_probe.py 16 > gpurun_out/r2g_noise_k16.log 2>&1; tail -8 gpurun_out/r2g_noise_k16.log
