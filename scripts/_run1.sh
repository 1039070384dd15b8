cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lock
D=$GRAFT_REPO_ROOT/sage_slam_amd/_variants
M=$GRAFT_REPO_ROOT/sage_slam_amd/libsage_ba.so
for i in 1 2 3; do
  echo "k64 base      $(SAGE_BA_LIB=$D/pitch0.so SAGE_SAMPLE_PERM=0 timeout 300 python scripts/kbench.py 64 5 2>/dev/null | tail -1)"
  echo "k64 p12+perm  $(SAGE_BA_LIB=$M SAGE_SAMPLE_PERM=1 timeout 300 python scripts/kbench.py 64 5 2>/dev/null | tail -1)"
  echo "k64 p12 only  $(SAGE_BA_LIB=$M SAGE_SAMPLE_PERM=0 timeout 300 python scripts/kbench.py 64 5 2>/dev/null | tail -1)"
  echo "k64 perm only $(SAGE_BA_LIB=$D/pitch0.so SAGE_SAMPLE_PERM=1 timeout 300 python scripts/kbench.py 64 5 2>/dev/null | tail -1)"
  echo "c4 base      $(SAGE_BA_LIB=$D/pitch0.so SAGE_SAMPLE_PERM=0 timeout 300 python scripts/kbench.py 16 3 256 320 32 32 2>/dev/null | tail -1)"
  echo "c4 p12+perm  $(SAGE_BA_LIB=$M SAGE_SAMPLE_PERM=1 timeout 300 python scripts/kbench.py 16 3 256 320 32 32 2>/dev/null | tail -1)"
done 2>&1 | tee gpurun_out/lock/ab3.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/lock/parity3.txt
bash scripts/pmc_run.sh p12 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU"
grep -A6 "^photo_kernel<32, 16" gpurun_out/pmc_p12/summary.txt
