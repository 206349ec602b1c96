cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r05_gputests.log 2>&1; tail -3 gpurun_out/r05_gputests.log
bash tools/experiments/r05/run_nvs4.sh 2>&1 | tail -14
