cd $GRAFT_REPO_ROOT
for p in 0 1 0 1; do SAGE_PIPELINE=$p python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r2j_pipe$p.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r2j_pipe$p.json'));print('pipeline',$p,round(d['ms_per_step'],4),round(d['roofline']['avg_launch_ms'],4))"; done
for nl in 1 0; do SAGE_BENCH_NO_LOOPS=$nl SAGE_DEBUG_TIMING=1 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2j_c5_noloops$nl.json 2> gpurun_out/r2j_c5_noloops$nl.err; python -c "
import json;d=json.load(open('gpurun_out/r2j_c5_noloops$nl.json'));print('config5 no_loops',$nl,round(d['ms_per_step'],3))"; grep "sage solve\|sage cholesky\|device solve" gpurun_out/r2j_c5_noloops$nl.err | tail -3; done
