#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the three rocprofv3 passes the numbers in profiles/ come from.
#   pass 1  --kernel-trace --stats          per-kernel durations (must agree with bench.py's HIP events)
#   pass 2  --pmc FETCH_SIZE                HBM read KiB per launch   (own pass, no tracing beside it)
#   pass 3  --pmc WRITE_SIZE                HBM write KiB per launch
# then `python tools/summarize_rocprof.py <tag> <workload key>` here copies the summaries into profiles/.
# usage: tools/profile_round.sh [bench args]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_write.log 2>&1
grep '^{' $R/gpurun_out/prof_kt.log | tail -1
