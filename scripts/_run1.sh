cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_tune.py -q -m gpu -x 2>&1 | tail -6
for c in 4 0 2 5; do
  if [ $c = 0 ]; then A=""; else A="--config $c"; fi
  timeout 600 python bench.py $A --steps 20 --warmup 5 --no-cpu-baseline --emulate-shard off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', round(d['ms_per_step'],4), d['config']['photo_runs'], {k:round(v,3) for k,v in d['phase_ms'].items() if isinstance(v,float)})"
done
