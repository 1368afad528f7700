R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu_round2_profiles.sh v4 2>&1 | tail -12
bash tools/gpu_round2_final.sh 2>&1 | tail -14
