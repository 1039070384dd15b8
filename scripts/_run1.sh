cd $GRAFT_REPO_ROOT
SAGE_EDGE_ORDER=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
echo k64
bash scripts/ab_env_kbench.sh "base: order:SAGE_EDGE_ORDER=1" "64 5" 3
echo c4
bash scripts/ab_env_kbench.sh "base:SAGE_PHOTO_TPB=12 order:SAGE_PHOTO_TPB=12,SAGE_EDGE_ORDER=1 o9:SAGE_PHOTO_TPB=9,SAGE_EDGE_ORDER=1 o8:SAGE_PHOTO_TPB=8,SAGE_EDGE_ORDER=1" "16 3 256 320 32 32" 2
