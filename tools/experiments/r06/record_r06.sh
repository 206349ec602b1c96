#!/bin/bash
# Round 6: every record under profiles/ that carries the device sources' hash, on one lease.  The default bench line FIRST (a fresh lease is
# in its fast state, and the run leaves the self-check mark in the cache files: the profiled runs of the same command then launch exactly
# warm-up + steps times), then tools/record_round.sh's passes, the streamed-CQI passes, the drop-in logs.
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
python3 bench.py > gpurun_out/r06_bench_first.log 2> gpurun_out/r06_bench_first.err
tools/record_round.sh r06 > gpurun_out/r06_record.log 2>&1
tools/profile_streamed.sh > gpurun_out/r06_streamed.log 2>&1
cd $R
./tools/dropin_latency 2000 > gpurun_out/r06_dropin_latency.log 2>&1
RS_DROPIN_TIMING=1 ./tools/dropin_latency 1000 2>&1 | grep "rs_schedule_tti x" > gpurun_out/r06_dropin_timing.log
cd /tmp
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r06_prof_dropin" -o dl --output-format csv -- "$R/tools/dropin_latency" 1000 > "$R/gpurun_out/r06_prof_dropin.log" 2>&1
cd $R
python tools/summarize_dropin_prof.py $(find gpurun_out/r06_prof_dropin -name "*kernel_trace.csv" | head -1) > gpurun_out/r06_dropin_kernel_times.md 2>&1
find gpurun_out/r06_prof_dropin -name "*kernel_trace.csv" -delete
RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so RS_JIT_EXTRA=-DRS_STAMPS RS_STAMPS_JIT=1 RS_STAMPS_EPOCH=1 python tools/dropin_stamps.py > gpurun_out/r06_dropin_stamps.log 2>&1
tail -3 gpurun_out/r06_record.log; cut -c1-200 gpurun_out/r06_bench_first.log
