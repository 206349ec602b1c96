#!/bin/bash
# round 6, second session: the K-simulators table's key rows again on the final sources (the one-TTI kernel is ~12 % shorter)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
{
python3 -c "import radiosaber_amd as rs; print('device sources', rs.device_source_hash())"
for K in 1 3 9 27; do timeout 300 ./tools/dropin_concurrency threads $K 500x25 2000 hwq=16; done
for K in 9 27; do timeout 300 ./tools/dropin_concurrency threads $K 100x64 2000 hwq=16; done
timeout 300 ./tools/dropin_concurrency threads 27 500x25 1000 think=1000 hwq=16
for K in 3 9 12; do timeout 300 ./tools/dropin_concurrency procs $K 500x25 2000 hwq=1; done
timeout 300 ./tools/dropin_concurrency procs 9 500x25 1000 think=1000 hwq=1
} > gpurun_out/r06/run23_concurrency_final.log 2>&1
cat gpurun_out/r06/run23_concurrency_final.log
