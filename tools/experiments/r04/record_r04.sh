#!/bin/bash
# Round 4, final record on ONE lease (through gpurun): everything under profiles/ that carries the device sources' hash.
# Then HERE: python tools/summarize_rocprof.py r04; python tools/summarize_shapes.py r04 --phase-shares-headline '{...}';
#            python tools/summarize_streamed.py r04; python tools/summarize_queue_prof.py r04q after; cp gpurun_out/... profiles/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd ..
python -c "import radiosaber_amd as rs; print('device sources', rs.device_source_hash())" > $O/r04_source_hash.log
python -m pytest tests -m gpu -q > $O/r04_gputests.log 2>&1; echo "pytest rc $?" >> $O/r04_gputests.log; tail -2 $O/r04_gputests.log
bash tools/record_round.sh r04 > $O/r04_record_round.log 2>&1
bash tools/profile_streamed.sh > $O/r04_streamed.log 2>&1
bash tools/profile_gap.sh > $O/r04_gap.log 2>&1
cd ..
tools/dropin_latency 2000 > $O/r04_dropin_latency.log 2>&1
RS_DROPIN_TIMING=1 tools/dropin_latency 1000 2>&1 | grep -v "^sched" > $O/r04_dropin_timing.log
bash tools/campaign_r04.sh > $O/r04_campaign.log 2>&1
grep -c "SOAK OK" $O/r04_campaign.log; grep "SOAK MISMATCH\|Error\|Traceback" $O/r04_campaign.log | head
tail -3 $O/r04_gap.log; cut -c1-200 $O/r04_bench_default.log
