#!/bin/bash
# Round 6, final lease(s) on the frozen sources.  final_r06.sh records: the whole GPU suite, then every record
# (tools/experiments/r06/record_r06.sh).  final_r06.sh campaign: tools/campaign_r06.sh.
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
case "${1:-records}" in
records)
  ( time python -m pytest tests -m gpu -q ) > gpurun_out/r06_gputests.log 2>&1; tail -4 gpurun_out/r06_gputests.log
  bash tools/experiments/r06/record_r06.sh ;;
campaign)
  bash tools/campaign_r06.sh > gpurun_out/r06_campaign_final.log 2>&1; grep -c "bit-exact\|SOAK" gpurun_out/r06_campaign_final.log; tail -5 gpurun_out/r06_campaign_final.log ;;
esac
