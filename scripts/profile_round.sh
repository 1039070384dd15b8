#!/bin/bash
# Round profile: bench JSON, bench under rocprofv3 --kernel-trace --stats (rocpd db -> kernel stats CSV), and the HBM
# counters of the dominant kernels (separate --pmc passes).  usage (on the GPU box): bash scripts/profile_round.sh r01_v10
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
timeout 600 python bench.py --steps 40 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $DB $OUT/kernel_stats.csv
bash $R/scripts/pmc_run.sh $TAG "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"
cp $R/gpurun_out/pmc_$TAG/summary.txt $OUT/pmc_summary.txt
