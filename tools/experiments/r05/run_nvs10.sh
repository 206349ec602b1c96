cd $GRAFT_REPO_ROOT
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -5
bash tools/experiments/r05/run_nvs8.sh 2>&1 | tail -5
bash tools/experiments/r05/run_nvs9.sh 2>&1 | tail -6
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -5
