set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_configs.py::test_config3_window_k64 > gpurun_out/r2d_gpu_suite.log 2>&1; tail -5 gpurun_out/r2d_gpu_suite.log
python bench.py --mode edge > gpurun_out/r2d_bench_edge.json 2> gpurun_out/r2d_bench_edge.err; tail -c 1500 gpurun_out/r2d_bench_edge.json; tail -5 gpurun_out/r2d_bench_edge.err
for c in 2 4 5; do python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2d_bench_config$c.json 2> gpurun_out/r2d_bench_config$c.err; python -c "
import json;d=json.load(open('gpurun_out/r2d_bench_config$c.json'));print('config',$c,d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['geo_kernel']['avg_launch_ms'],d['roofline']['error_pass_ms'],d['config']['accepted_steps'])"; tail -2 gpurun_out/r2d_bench_config$c.err; done
python bench.py --steps 40 --warmup 5 > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err; python -c "
import json;d=json.load(open('gpurun_out/r2d_bench.json'));print('headline',d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['cpu_baseline']['value'])"
