cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
 for v in "20:def" "20:27" "20:60" "40:def"; do
  st=${v%%:*}; ins=${v#*:}
  if [ $ins = def ]; then unset SAGE_BENCH_INSTR_STEPS; else export SAGE_BENCH_INSTR_STEPS=$ins; fi
  echo "steps $st instr $ins: $(timeout 300 python bench.py --gpus 1 --steps $st --warmup 5 --no-cpu-baseline --emulate-shard off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['phase_ms']['solve'], d['roofline']['avg_launch_ms'])")"
 done
done
unset SAGE_BENCH_INSTR_STEPS
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final_bench.json 2>/dev/null; tail -c 300 gpurun_out/r06_final_bench.json
