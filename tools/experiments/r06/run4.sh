#!/bin/bash
# round 6, fourth lease: one-TTI kernels without state traffic; how many hardware queues a simulator process should ask for
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06/run4_gputests.log
( ./tools/dropin_latency 2000; RS_DROPIN_TIMING=1 ./tools/dropin_latency 1000 2>&1 | grep "rs_schedule_tti x" ) > gpurun_out/r06/run4_dropin_latency.log 2>&1
{
RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so RS_JIT_EXTRA=-DRS_STAMPS RS_STAMPS_JIT=1 RS_STAMPS_EPOCH=1 python tools/dropin_stamps.py
} > gpurun_out/r06/run4_dropin_stamps.log 2>&1
{
echo "# processes with ONE hardware queue each (GPU_MAX_HW_QUEUES=1): is it the queue count or the process count?"
for K in 6 9 12 27; do
  timeout 300 ./tools/dropin_concurrency procs $K 500x25 2000 hwq=1
done
for K in 9 27; do
  timeout 300 ./tools/dropin_concurrency procs $K 500x25 2000 hwq=2
done
echo "# paced, one queue per process"
for K in 9 27; do
  timeout 300 ./tools/dropin_concurrency procs $K 500x25 1000 think=1000 hwq=1
  timeout 300 ./tools/dropin_concurrency procs $K 500x25 1000 think=1000 hwq=2
done
echo "# threads, paced, 8 / 16 queues"
for Q in 8 16; do timeout 300 ./tools/dropin_concurrency threads 27 500x25 1000 think=1000 hwq=$Q; done
for Q in 8 16; do timeout 300 ./tools/dropin_concurrency threads 27 500x25 2000 hwq=$Q; done
} > gpurun_out/r06/run4_concurrency.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r06/prof_dropin_latency" -o dl --output-format csv -- "$GRAFT_REPO_ROOT/tools/dropin_latency" 1000 > "$GRAFT_REPO_ROOT/gpurun_out/r06/prof_dropin_latency.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/r06 -name "*kernel_trace.csv" -size +6M -delete
tail -3 gpurun_out/r06/run4_gputests.log
