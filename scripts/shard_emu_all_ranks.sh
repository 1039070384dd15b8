#!/bin/bash
# dev tool (GPU box): every rank of an N-rank job emulated one after the other on this device (scripts/shard_emu_probe.py);
# the job's LM iteration is the slowest rank's.   usage: bash scripts/shard_emu_all_ranks.sh [config] [N]
CFG=${1:-3}; N=${2:-8}
R=$GRAFT_REPO_ROOT
cd $R
for r in $(seq 0 $((N-1))); do
  EMU_COMPARE=0 python scripts/shard_emu_probe.py $CFG $r/$N 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config $CFG rank %d/%d: directed edges %3d  ms_per_step %.4f (p10-p90 %.4f-%.4f)  one-GPU classic %.4f  speedup %.2f  photo %.4f geo %.4f' % (d['rank'], d['world'], d['local_directed_edges'], d['ms_per_step'], d['ms_per_step_p10_p90'][0], d['ms_per_step_p10_p90'][1], d['one_gpu_ms_per_step_classic'], d['speedup_vs_one_gpu_classic'], d['kernel_ms']['photo_linearize'], d['kernel_ms']['geo_linearize']))"
done
