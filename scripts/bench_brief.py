import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line); r=d["roofline"]
        print("ms/step %.3f photo %.3f geo %.3f err %s"%(d["ms_per_step"], r["avg_launch_ms"], r["geo_kernel"]["avg_launch_ms"], r["error_pass_ms"]))
