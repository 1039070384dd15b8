cd $GRAFT_REPO_ROOT
D=$GRAFT_REPO_ROOT/sage_slam_amd/_variants
M=$GRAFT_REPO_ROOT/sage_slam_amd/libsage_ba.so
for i in 1 2 3; do
  echo "k64 main    $(SAGE_BA_LIB=$M timeout 300 python scripts/kbench.py 64 5 2>/dev/null | tail -1)"
  echo "k64 geolock $(SAGE_BA_LIB=$D/geolock.so timeout 300 python scripts/kbench.py 64 5 2>/dev/null | tail -1)"
  echo "c4 main    $(SAGE_BA_LIB=$M timeout 300 python scripts/kbench.py 16 3 256 320 32 32 2>/dev/null | tail -1)"
  echo "c4 geolock $(SAGE_BA_LIB=$D/geolock.so timeout 300 python scripts/kbench.py 16 3 256 320 32 32 2>/dev/null | tail -1)"
done 2>&1 | tee gpurun_out/lock/ab4.txt
