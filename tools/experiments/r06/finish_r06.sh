#!/bin/bash
# HERE, after tools/experiments/r06/final_r06.sh came back through gpurun: summaries and copies into profiles/.
# usage: tools/experiments/r06/finish_r06.sh '<phase-shares JSON of the headline shape>'
set -e
cd "$(dirname "$0")/../../.."
python tools/summarize_rocprof.py r06 | tail -3
python tools/summarize_shapes.py r06 --phase-shares-headline "$1" | tail -16
python tools/summarize_streamed.py r06 | tail -2
python tools/summarize_queue_prof.py r06q after | tail -3 || true
(cat gpurun_out/r06_dropin_latency.log; echo; echo "# RS_DROPIN_TIMING=1, the same contexts in the same order (six built-in, six specialised, then five + five with cqi_epoch), completion by the polled word:"; cat gpurun_out/r06_dropin_timing.log;
 echo; echo "# kernel durations of the same program under rocprofv3 --kernel-trace (tools/summarize_dropin_prof.py):"; cat gpurun_out/r06_dropin_kernel_times.md;
 echo; echo "# phase cycles of one call, specialised kernel with cqi_epoch, -DRS_STAMPS build (tools/dropin_stamps.py):"; grep sched gpurun_out/r06_dropin_stamps.log) > profiles/r06_dropin_latency.log
cp gpurun_out/r06_bench_first.log profiles/r06_bench_first.log
cp gpurun_out/r06_bench_default.log profiles/r06_bench_default.log
cp gpurun_out/r06_stamps_final.log profiles/r06_phase_stamps.log
cp gpurun_out/r06_gputests.log profiles/r06_gputests.log
find gpurun_out/r06_prof_dropin -name "*kernel_stats.csv" -exec cp {} profiles/r06_dropin_kernel_stats.csv \;
cat gpurun_out/r06_sweep.log
