#!/bin/bash
# Round 4, GPU run 11: queue model with its cumulative counters in registers -- parity (tests + fuzz), rate and HBM traffic
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run11; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_queues.py tests/test_gpu_round3.py -m gpu -x -q > $O/pytest_q.log 2>&1; echo "pytest rc $?" >> $O/pytest_q.log; tail -3 $O/pytest_q.log
python tools/fuzz_queues.py 9000 40 > $O/fuzz_q.log 2>&1; tail -2 $O/fuzz_q.log
bash tools/profile_queue_mode.sh r04q > $O/profile_q.log 2>&1; tail -8 $O/profile_q.log
