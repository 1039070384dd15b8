cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
 for v in off on; do
  if [ $v = on ]; then export SAGE_PLACEMENT_MONITOR=1; else unset SAGE_PLACEMENT_MONITOR; fi
  echo "monitor $v: $(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --emulate-shard off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['phase_ms']['solve'],4), d['host_placement'].get('placement_monitor_moves'), d['host_placement'].get('loadavg_1min'))")"
 done
done
