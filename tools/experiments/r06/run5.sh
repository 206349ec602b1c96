#!/bin/bash
# round 6, fifth lease: the default bench line with its new fields; the tests added since the last full suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
python3 bench.py > gpurun_out/r06/run5_bench_default.log 2> gpurun_out/r06/run5_bench_default.err; echo "bench rc $?" >> gpurun_out/r06/run5_bench_default.err
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -x -q -k "hardware_queues or bench_line or eight_threads or never_trusts" 2>&1 | tail -15 > gpurun_out/r06/run5_tests.log
python3 bench.py > gpurun_out/r06/run5_bench_second.log 2>> gpurun_out/r06/run5_bench_default.err
tail -3 gpurun_out/r06/run5_tests.log; tail -3 gpurun_out/r06/run5_bench_default.err; cut -c1-300 gpurun_out/r06/run5_bench_default.log
