#!/bin/bash
# dev tool (GPU box): per-kernel average durations of the kernel micro-bench (scripts/kbench.py, K=64 headline window)
# under rocprofv3 --kernel-trace --stats.   usage: [ENV=...] bash scripts/kstats.sh TAG
TAG=${1:-k}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- python $R/scripts/kbench.py 64 6 > $OUT/kbench.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $DB $OUT/kernel_stats.csv > /dev/null
echo "== $TAG: $(cat $OUT/kbench.json)"
grep -E "photo_kernel|geo_kernel" $OUT/kernel_stats.csv | cut -c1-160
