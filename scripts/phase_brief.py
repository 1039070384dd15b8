"""dev tool: ms_per_step + phase_ms of a bench.py line on stdin"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in (d.get("phase_ms") or {}).items()})
