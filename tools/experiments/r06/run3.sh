#!/bin/bash
# round 6, third lease: the tests that changed, hardware queues / process slots / paced callers
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r06/run3_tests.log
{
echo "# threads: one hardware queue per context (GPU_MAX_HW_QUEUES=32) against the runtime's default of 4"
for K in 3 9 27; do
  timeout 300 ./tools/dropin_concurrency threads $K 500x25 2000
  timeout 300 ./tools/dropin_concurrency threads $K 500x25 2000 hwq=32
done
for K in 9 27; do timeout 300 ./tools/dropin_concurrency threads $K 100x64 2000 hwq=32; done
echo "# processes: where the cliff is"
for K in 4 6 8 9 12; do timeout 300 ./tools/dropin_concurrency procs $K 500x25 2000; done
echo "# paced callers: every worker spends 1 000 us of host time between two calls (the reference's simulator spends more: SURVEY 6)"
for K in 1 9 27; do
  timeout 300 ./tools/dropin_concurrency procs $K 500x25 1000 think=1000
  timeout 300 ./tools/dropin_concurrency threads $K 500x25 1000 think=1000
  timeout 300 ./tools/dropin_concurrency threads $K 500x25 1000 think=1000 hwq=32
done
for K in 27; do
  timeout 300 ./tools/dropin_concurrency procs $K 100x64 1000 think=1000
  timeout 300 ./tools/dropin_concurrency threads $K 100x64 1000 think=1000 hwq=32
done
} > gpurun_out/r06/run3_concurrency.log 2>&1
tail -4 gpurun_out/r06/run3_tests.log
# where the one-TTI kernel's cycles go (stamped builds; load / store phases are slots of their own since round 6)
{
RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so python tools/dropin_stamps.py
RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so RS_JIT_EXTRA=-DRS_STAMPS RS_STAMPS_JIT=1 python tools/dropin_stamps.py
RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so RS_JIT_EXTRA=-DRS_STAMPS RS_STAMPS_JIT=1 RS_STAMPS_EPOCH=1 python tools/dropin_stamps.py
} > gpurun_out/r06/run3_dropin_stamps.log 2>&1
# which compiler builds the bench kernel, and does it matter?
{
python bench.py --no-cpu-baseline --no-streamed --no-cells1024 --steps 6
RS_BENCH_LOAD_LIBRARY_FIRST=1 python bench.py --no-cpu-baseline --no-streamed --no-cells1024 --steps 6
python bench.py --no-cpu-baseline --no-streamed --no-cells1024 --steps 6
RS_BENCH_LOAD_LIBRARY_FIRST=1 python bench.py --no-cpu-baseline --no-streamed --no-cells1024 --steps 6
} > gpurun_out/r06/run3_compiler_ab.log 2>&1
