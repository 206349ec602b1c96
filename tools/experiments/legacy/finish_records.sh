#!/bin/bash
# HERE, after tools/record_r04.sh came back through gpurun: summaries and copies into profiles/ (then tools/record_r04b.sh on the GPU box, then
# `python tools/summarize_queue_prof.py r04q after` and the copies of its three logs).  usage: tools/finish_records.sh '<phase-shares JSON>'
set -e
cd "$(dirname "$0")/.."
python tools/summarize_rocprof.py r04 | tail -3
python tools/summarize_shapes.py r04 --phase-shares-headline "$1" | tail -12
python tools/summarize_streamed.py r04 | tail -2
cp gpurun_out/r04_campaign.log profiles/r04_campaign_final.log
(cat gpurun_out/r04_dropin_latency.log; echo; echo "# RS_DROPIN_TIMING=1, the same six contexts (built-in, then specialised):"; cat gpurun_out/r04_dropin_timing.log) > profiles/r04_dropin_latency.log
cp gpurun_out/r04_stamps_final.log profiles/r04_phase_stamps.log
cat gpurun_out/r04_sweep.log
tail -4 gpurun_out/r04_gap.log
