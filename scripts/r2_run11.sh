cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline"
echo base; $B 2>/dev/null | python scripts/bench_brief.py
echo base+2streams; SAGE_TWO_STREAMS=1 $B 2>/dev/null | python scripts/bench_brief.py
echo geo3; SAGE_BA_LIB=sage_slam_amd/_variants/libsage_geo3.so $B 2>/dev/null | python scripts/bench_brief.py
echo geo3+2streams; SAGE_TWO_STREAMS=1 SAGE_BA_LIB=sage_slam_amd/_variants/libsage_geo3.so $B 2>/dev/null | python scripts/bench_brief.py
