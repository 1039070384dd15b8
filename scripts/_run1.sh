cd $GRAFT_REPO_ROOT
for t in 6 3 4 12 2; do
  SAGE_PHOTO_TPB=$t timeout 600 python bench.py --config 5 --steps 12 --warmup 3 --no-cpu-baseline --emulate-shard off --no-tune 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('tpb $t', round(d['ms_per_step'],4), 'photo', round(r['avg_launch_ms'],4), 'err', r.get('error_pass_ms'), {k:round(v,3) for k,v in d['phase_ms'].items() if isinstance(v,float)})"
done
