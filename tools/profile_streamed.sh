#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the streamed-CQI mode (cqi_refresh = 1, a grid from HBM every TTI) under rocprofv3 -- kernel trace and the two
# HBM counters, each PMC in its own pass (no tracing beside --pmc).  Then HERE: python tools/summarize_streamed.py -> profiles/traffic.json[..._refresh1]
# (measurement passes: the summaries take per-launch means over every dispatch of the cell kernel, so the self-check's three short trial
#  launches -- on by default since round 6 for builds without the mark -- are switched off here; results are checked everywhere else)
export RS_JIT_SELFCHECK=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/streamed; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-r64 --no-streamed --no-cells1024 --cqi-refresh 1 --ttis 2000 --steps 4 --warmup 1"
python3 $R/bench.py $ARGS > $O/plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $ARGS > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
grep -h '^{' $O/plain.log | cut -c1-200
