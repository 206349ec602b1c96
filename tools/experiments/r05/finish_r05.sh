#!/bin/bash
# HERE, after tools/experiments/r05/final_r05.sh came back through gpurun: summaries and copies into profiles/.
# usage: tools/experiments/r05/finish_r05.sh '<phase-shares JSON of the headline shape>'
set -e
cd "$(dirname "$0")/../../.."
python tools/summarize_rocprof.py r05 | tail -3
python tools/summarize_shapes.py r05 --phase-shares-headline "$1" | tail -16
python tools/summarize_streamed.py r05 | tail -2
python tools/summarize_queue_prof.py r05q after | tail -3 || true
(cat gpurun_out/r05_dropin_latency.log; echo; echo "# RS_DROPIN_TIMING=1, the same six contexts (built-in, then specialised), completion by the polled word:"; cat gpurun_out/r05_dropin_timing.log;
 echo; echo "# RS_DROPIN_POLL=0 (hipStreamSynchronize):"; cat gpurun_out/r05_dropin_timing_nopoll.log) > profiles/r05_dropin_latency.log
cp gpurun_out/r05_bench_default.log profiles/r05_bench_default.log
cp gpurun_out/r05_stamps_final.log profiles/r05_phase_stamps.log
cp gpurun_out/r05_gputests.log profiles/r05_gputests.log
cat gpurun_out/r05_sweep.log
