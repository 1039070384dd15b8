#!/bin/bash
# dev tool (GPU box): A/B of environment switches on the kernel micro-bench.
#   ab_env_kbench.sh "name:VAR=val,VAR2=val name2:" "K n [H W FS CS]" [reps]
R=$GRAFT_REPO_ROOT
cd $R
REPS=${3:-2}
for i in $(seq 1 $REPS); do
for nv in $1; do
  name=${nv%%:*}; envs=${nv#*:}
  env $(echo $envs | tr ',' ' ') timeout 300 python scripts/kbench.py $2 2>/dev/null | tail -1 | sed "s/^/$name /"
done
done
