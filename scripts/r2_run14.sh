cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r2n_full_gpu.log 2>&1; tail -15 gpurun_out/r2n_full_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
