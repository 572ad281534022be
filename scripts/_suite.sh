set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
(timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -24) > gpurun_out/r6a/gputest.log
tail -20 gpurun_out/r6a/gputest.log
