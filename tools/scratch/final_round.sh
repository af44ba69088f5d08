cd $GRAFT_REPO_ROOT
TAG=${1:-r03s}
bash tools/gpu_round.sh $TAG tests 2>&1 | tail -40
bash tools/profile_round.sh $TAG 45 2>&1 | tail -12
