#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the three rocprofv3 passes the numbers in profiles/ come from.
#   pass 1  --kernel-trace --stats          per-kernel durations of the default bench.py workload (must agree with its HIP events)
#   pass 2  --pmc FETCH_SIZE                HBM read KiB per launch   (own pass, no tracing beside it)
#   pass 3  --pmc WRITE_SIZE                HBM write KiB per launch
# (--no-r64 --no-cpu-baseline: the extra 64-RBG batch would run under the same kernel name; the timed region is bench.py's default)
# then HERE `python tools/summarize_rocprof.py <tag>` copies the summaries into profiles/.
# usage: tools/profile_round.sh [bench args]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 $R/bench.py --no-cpu-baseline --no-r64 --no-streamed --no-cells1024 "$@" > $R/gpurun_out/prof_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --no-cpu-baseline --no-r64 --no-streamed --no-cells1024 --steps 6 --warmup 1 "$@" > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --no-cpu-baseline --no-r64 --no-streamed --no-cells1024 --steps 6 --warmup 1 "$@" > $R/gpurun_out/prof_write.log 2>&1
grep '^{' $R/gpurun_out/prof_kt.log | tail -1
