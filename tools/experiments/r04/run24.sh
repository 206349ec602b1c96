#!/bin/bash
# Round 4, GPU run 24: the new streamed-mode / grid-ahead tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run24; mkdir -p $O; cd ..
python -m pytest tests/test_gpu_round4.py -m gpu -q -k "streamed or fetched" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -n "FAILED\|passed\|failed\|rc \|Error" $O/pytest.log | tail -20
