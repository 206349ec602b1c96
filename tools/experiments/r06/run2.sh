#!/bin/bash
# round 6, second lease: the whole GPU suite, then the concurrency matrix and the kernel durations under K = 1 / 9
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r06/run2_gputests.log
{
for shape in 500x25 100x64; do
  for K in 1 3 9 27; do
    timeout 300 ./tools/dropin_concurrency threads $K $shape 2000
    timeout 300 ./tools/dropin_concurrency procs $K $shape 2000
  done
done
echo "# without cqi_epoch (every call ships and transposes its block)"
for K in 1 27; do timeout 300 ./tools/dropin_concurrency procs $K 500x25 2000 noepoch; done
echo "# RS_DROPIN_POLL=0 (hipStreamSynchronize instead of the polled word)"
for K in 1 27; do RS_DROPIN_POLL=0 timeout 300 ./tools/dropin_concurrency procs $K 500x25 2000; done
echo "# built-in kernels"
for K in 1 27; do timeout 300 ./tools/dropin_concurrency procs $K 500x25 2000 builtin; done
} > gpurun_out/r06/run2_concurrency.log 2>&1
cd /tmp
for K in 1 9; do
  rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r06/prof_threads_k$K" -o k$K --output-format csv -- "$GRAFT_REPO_ROOT/tools/dropin_concurrency" threads $K 500x25 1000 > "$GRAFT_REPO_ROOT/gpurun_out/r06/prof_threads_k$K.log" 2>&1
done
cd "$GRAFT_REPO_ROOT"
find gpurun_out/r06 -name "*kernel_trace.csv" -size +8M -delete
tail -3 gpurun_out/r06/run2_gputests.log
