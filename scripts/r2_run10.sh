cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r2l_parity.log 2>&1; tail -2 gpurun_out/r2l_parity.log
bash scripts/profile_round.sh r02_v1
python -c "
import json;d=json.load(open('gpurun_out/prof_r02_v1/bench.json'));print('bench',round(d['ms_per_step'],4),round(d['roofline']['avg_launch_ms'],4),round(d['roofline']['frac'],4),d['cpu_baseline']['value'])"
head -12 gpurun_out/prof_r02_v1/kernel_stats.csv
