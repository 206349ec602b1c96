#!/bin/bash
# Round 5, final lease: the whole GPU suite on the frozen sources, then every record (tools/experiments/r05/record_r05.sh).
R=$GRAFT_REPO_ROOT; cd $R
( time python -m pytest tests -m gpu -q ) > gpurun_out/r05_gputests.log 2>&1; tail -4 gpurun_out/r05_gputests.log
bash tools/experiments/r05/record_r05.sh
