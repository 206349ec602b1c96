#!/bin/bash
# round 6, first lease: the new tests, then the drop-in call alone and K at a time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
nproc > gpurun_out/r06/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" >> gpurun_out/r06/nproc.txt
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_abi.py -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r06/run1_tests.log
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_adapter.py -m gpu -x -q -k "selfcheck or checkpoint or autotune or adapter" 2>&1 | tail -15 >> gpurun_out/r06/run1_tests.log
( ./tools/dropin_latency 2000; RS_DROPIN_TIMING=1 ./tools/dropin_latency 1000 2>&1 | grep "rs_schedule_tti x" ) > gpurun_out/r06/run1_dropin_latency.log 2>&1
for K in 1 3 9; do
  timeout 300 ./tools/dropin_concurrency threads $K 500x25 2000
  timeout 300 ./tools/dropin_concurrency procs $K 500x25 2000
done > gpurun_out/r06/run1_concurrency.log 2>&1
tail -5 gpurun_out/r06/run1_tests.log
