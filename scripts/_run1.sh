cd $GRAFT_REPO_ROOT
bash scripts/ab_env_kbench.sh "flush4: flush3:SAGE_PHOTO_FLUSH=3 flush0:SAGE_PHOTO_FLUSH=0 flush5:SAGE_PHOTO_FLUSH=5 flush6:SAGE_PHOTO_FLUSH=6 flush1:SAGE_PHOTO_FLUSH=1" "64 5" 2
for f in 3; do echo "== FLUSH $f"; SAGE_PHOTO_FLUSH=$f timeout 600 python tests/tools/delta_probe.py 64 2>&1 | tail -3 | cut -c1-420; done
echo "== config4"; bash scripts/ab_env_kbench.sh "flush4: flush3:SAGE_PHOTO_FLUSH=3" "16 3 256 320 32 32" 2
echo "== config2"; bash scripts/ab_env_kbench.sh "flush4: flush3:SAGE_PHOTO_FLUSH=3" "16 5 128 160 16 32" 2
