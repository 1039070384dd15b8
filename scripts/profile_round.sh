#!/bin/bash
# Round profile: bench JSON, bench under rocprofv3 --kernel-trace --stats (rocpd db -> kernel stats CSV), and the
# counters of the dominant kernels (separate --pmc passes: HBM traffic, L2, SQ issue / wait).
# usage (on the GPU box): bash scripts/profile_round.sh r04_v1            # headline window
#                         bash scripts/profile_round.sh r04_config4 4     # another BASELINE configuration (2, 4)
TAG=${1:-r04}
CFG=${2:-}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
BARGS=""
if [ -n "$CFG" ]; then BARGS="--config $CFG"; fi
if [ "$CFG" = "4" ]; then export KBENCH_ARGS="16 3 256 320 32 32"; fi
if [ "$CFG" = "2" ]; then export KBENCH_ARGS="16 3 128 160 16 32"; fi
timeout 900 python bench.py $BARGS --steps 40 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
# r06: bench.py tunes the photometric run length on the window (sage_window_tune_runs); the trace and counter passes below pin the
# run length it found instead of tuning again, so that their per-kernel averages hold the timed plan's launches only
TPB=$(python -c "import json,sys; print(json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['config']['photo_runs'].get('tpb', 0))" 2>/dev/null)
if [ -n "$TPB" ] && [ "$TPB" != "0" ]; then export SAGE_PHOTO_TPB=$TPB; echo "photometric run length pinned to $TPB for the trace / counter passes" > $OUT/pinned_runs.txt; fi
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- python $R/bench.py $BARGS --steps 20 --warmup 3 --no-cpu-baseline --emulate-shard off > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $DB $OUT/kernel_stats.csv
bash $R/scripts/pmc_run.sh $TAG "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD" \
  "SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS"
cp $R/gpurun_out/pmc_$TAG/summary.txt $OUT/pmc_summary.txt
rm -rf $OUT/trace
