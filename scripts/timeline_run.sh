#!/bin/bash
# dev tool (GPU box): kernel timeline of one steady-state LM step of the headline bench -> gpurun_out/timeline_$1.txt
TAG=${1:-tl}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/tl_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-shard off > $R/gpurun_out/tl_$TAG.log 2>&1
f=$(find $R/gpurun_out/tl_$TAG -name "*kernel_trace.csv" | head -1)
python $R/scripts/timeline.py $f > $R/gpurun_out/timeline_$TAG.txt 2>&1
tail -30 $R/gpurun_out/timeline_$TAG.txt
