#!/bin/bash
# A/B of variant builds (scripts/build_variant.sh) on the kernel micro-bench: ab_kbench.sh "name:lib ..." [reps]
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab
mkdir -p $OUT
cd $R
REPS=${2:-2}
for i in $(seq 1 $REPS); do
for nv in $1; do
  name=${nv%%:*}; lib=${nv#*:}
  if [ "$lib" = "main" ]; then L="$R/sage_slam_amd/libsage_ba.so"; else L="$R/sage_slam_amd/_variants/$lib"; fi
  SAGE_BA_LIB=$L timeout 300 python scripts/kbench.py 64 5 > $OUT/kb_${name}_$i.json 2> $OUT/kb_${name}.err || echo "$name FAILED rc $?"
  echo "$name $(tail -1 $OUT/kb_${name}_$i.json)"
done
done
