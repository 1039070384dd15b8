cd $GRAFT_REPO_ROOT
SAGE_PHOTO_FLUSH=2 python scripts/kbench.py 64 20
for v in probe_l2_w3 probe_l2_w2 probe_l1_w2 probe_l1_w3; do echo $v; SAGE_PHOTO_FLUSH=2 SAGE_BA_LIB=$GRAFT_REPO_ROOT/sage_slam_amd/_variants/libsage_$v.so python scripts/kbench.py 64 20 2>&1 | tail -1; done
