#!/bin/bash
# Round 5: every record under profiles/ that carries the device sources' hash, on one lease (tools/record_round.sh r05), plus the
# streamed-CQI passes and the drop-in call's logs.  Then HERE: tools/experiments/r05/finish_r05.sh
R=$GRAFT_REPO_ROOT; cd $R
tools/record_round.sh r05 > gpurun_out/r05_record.log 2>&1
tools/profile_streamed.sh > gpurun_out/r05_streamed.log 2>&1
cd $R
g++ -O2 -std=c++17 -Iinclude tools/dropin_latency.cpp -Lradiosaber_amd -lradiosaber_hip -Wl,-rpath,$R/radiosaber_amd -o /tmp/dropin_latency
/tmp/dropin_latency 2000 > gpurun_out/r05_dropin_latency.log 2>&1
RS_DROPIN_TIMING=1 /tmp/dropin_latency 1000 2>&1 | grep "rs_schedule_tti x" > gpurun_out/r05_dropin_timing.log
RS_DROPIN_POLL=0 RS_DROPIN_TIMING=1 /tmp/dropin_latency 1000 2>&1 | grep "rs_schedule_tti x" > gpurun_out/r05_dropin_timing_nopoll.log
tail -3 gpurun_out/r05_record.log; cut -c1-200 gpurun_out/r05_bench_default.log
