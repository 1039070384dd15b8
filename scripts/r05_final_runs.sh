cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r05_v3 > gpurun_out/prof_r05_v3.log 2>&1
for c in 2 4 5; do python bench.py --config $c --steps 20 --warmup 5 > gpurun_out/r05_bench_config$c.json 2> gpurun_out/r05_bench_config$c.err; done
python bench.py --mode edge > gpurun_out/r05_bench_edge.json 2> gpurun_out/r05_bench_edge.err
bash scripts/shard_emu_run.sh r05k64 3 0/8 > gpurun_out/r05_emu_k64.txt 2>&1
bash scripts/shard_emu_run.sh r05c4 4 0/8 > gpurun_out/r05_emu_c4.txt 2>&1
EMU_COMPARE=1 python scripts/shard_emu_probe.py 3 3/8 30 > gpurun_out/r05_emu_k64_rank3.json 2>/dev/null
EMU_COMPARE=1 python scripts/shard_emu_probe.py 4 3/8 30 > gpurun_out/r05_emu_c4_rank3.json 2>/dev/null
bash scripts/profile_round.sh r05_config4 4 > gpurun_out/prof_r05_config4.log 2>&1
echo done
