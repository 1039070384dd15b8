cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r06_v5
bash scripts/profile_round.sh r06_v5_config4 4
