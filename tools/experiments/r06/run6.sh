#!/bin/bash
# round 6, sixth lease: the drop-in fuzz
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round6.py -m gpu -x -q -k "fuzz_seeds or hardware_queues" 2>&1 | tail -15 > gpurun_out/r06/run6_tests.log
timeout 1500 python tools/fuzz_dropin.py 1000 60 30 > gpurun_out/r06/run6_fuzz_dropin.log 2>&1
tail -3 gpurun_out/r06/run6_tests.log; tail -4 gpurun_out/r06/run6_fuzz_dropin.log
