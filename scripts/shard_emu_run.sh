#!/bin/bash
# dev tool (GPU box): emulated rank of an 8-rank job -- JSON of scripts/shard_emu_probe.py + the kernel timeline of its
# last steady LM iteration.  usage: bash scripts/shard_emu_run.sh TAG [config] [R/N]
TAG=${1:-emu}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/emu_$TAG
python $R/scripts/shard_emu_probe.py ${2:-3} ${3:-0/8} 30 > $R/gpurun_out/emu_$TAG.json 2> $R/gpurun_out/emu_$TAG.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/emu_$TAG -- python $R/scripts/shard_emu_probe.py ${2:-3} ${3:-0/8} 9 > $R/gpurun_out/emu_${TAG}_prof.json 2> $R/gpurun_out/emu_${TAG}_prof.err
f=$(find $R/gpurun_out/emu_$TAG -name "*kernel_trace.csv" | head -1)
python $R/scripts/timeline.py $f solve_scatter > $R/gpurun_out/emu_timeline_$TAG.txt 2>&1
rm -rf $R/gpurun_out/emu_$TAG
echo "== $TAG"; python -c "
import json,sys
d=json.load(open('$R/gpurun_out/emu_$TAG.json'))
print({k:d[k] for k in ('ms_per_step','ms_first_step_after_restart','one_gpu_ms_per_step_classic','speedup_vs_one_gpu_classic','kernel_ms','phase_ms') if k in d})"
cat $R/gpurun_out/emu_timeline_$TAG.txt
