cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sharded_lm.py -q -m gpu -x 2>&1 | tail -8
