cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sharded_lm.py -q -m gpu -x > gpurun_out/r2k_sharded.log 2>&1; tail -15 gpurun_out/r2k_sharded.log
SAGE_BENCH_NO_LOOPS=1 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config5 noloops 1 gpu',round(d['ms_per_step'],3))"
