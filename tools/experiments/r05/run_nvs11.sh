cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "sampler" 2>&1 | tail -1
bash tools/experiments/r05/run_nvs9.sh
bash tools/experiments/r05/run_nvs8.sh | tail -5
