cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_factor_cache.py -q -m gpu -x 2>&1 | tail -25
