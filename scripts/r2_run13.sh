cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "linearize_at_candidate" 2>&1 | tail -25
B="python bench.py --steps 40 --warmup 3 --no-cpu-baseline"
echo classic; $B 2>/dev/null | tee gpurun_out/r2m_classic.json | python scripts/bench_brief.py
echo candidate; $B --lm-variant candidate 2>/dev/null | tee gpurun_out/r2m_candidate.json | python scripts/bench_brief.py
python - <<'PY'
import json
for n in ("classic","candidate"):
    d=json.load(open(f"gpurun_out/r2m_{n}.json")); print(n, d["config"]["accepted_steps"], d["config"]["error_first_last"], d["lm_iters_per_sec"])
PY
