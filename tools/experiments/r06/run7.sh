#!/bin/bash
# round 6, seventh lease: does pinning a drop-in context's stream to one compute unit keep its code warm?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
{
echo "# default"; timeout 120 ./tools/dropin_latency 1500 | grep -E "specialised" 
for m in auto auto+ 5 100; do
  echo "# RS_DROPIN_CU=$m"; RS_DROPIN_CU=$m timeout 120 ./tools/dropin_latency 1500 | grep -E "specialised"
done
echo "# default again"; timeout 120 ./tools/dropin_latency 1500 | grep -E "specialised"
} > gpurun_out/r06/run7_cu_mask.log 2>&1
cd /tmp
RS_DROPIN_CU=auto rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r06/prof_dropin_cu" -o dl --output-format csv -- "$GRAFT_REPO_ROOT/tools/dropin_latency" 1000 > "$GRAFT_REPO_ROOT/gpurun_out/r06/prof_dropin_cu.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_dropin_prof.py $(find gpurun_out/r06/prof_dropin_cu -name "*kernel_trace.csv" | head -1) > gpurun_out/r06/run7_cu_mask_kernel_times.md 2>&1
find gpurun_out/r06/prof_dropin_cu -name "*kernel_trace.csv" -delete
{
for K in 9 27; do RS_DROPIN_CU=auto timeout 300 ./tools/dropin_concurrency threads $K 500x25 2000 hwq=16; done
RS_DROPIN_CU=auto timeout 300 ./tools/dropin_concurrency procs 4 500x25 2000
} > gpurun_out/r06/run7_cu_mask_concurrency.log 2>&1
sed 's/ (checksum.*//' gpurun_out/r06/run7_cu_mask.log
