#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the queue-model measurements of profiles/r03_queue_mode.md.
#   1. tools/bench_queue_mode.py: TTIs/s of exp-customize-20slices x 512 cells under sched 9 / 7 / 1, with the backlogged twin
#   2. rocprofv3 --kernel-trace --stats of the same command (sched 9)
#   3. FETCH_SIZE / WRITE_SIZE in their own --pmc passes (sched 9, then 7, then 1)
# usage: tools/profile_queue_mode.sh <tag>
# (measurement passes: the summaries take per-launch means over every dispatch of the cell kernel, so the self-check's three short trial
#  launches -- on by default since round 6 for builds without the mark -- are switched off here; results are checked everywhere else)
export RS_JIT_SELFCHECK=0
R=$GRAFT_REPO_ROOT; TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_queue_mode.py --with-backlogged > $R/gpurun_out/${TAG}_bench.log 2> $R/gpurun_out/${TAG}_bench.err
rm -rf $R/gpurun_out/${TAG}_kt $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -- python3 $R/tools/bench_queue_mode.py --sched 9,7,1 > $R/gpurun_out/${TAG}_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/tools/bench_queue_mode.py --sched 9,7,1 --launches 2 > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/tools/bench_queue_mode.py --sched 9,7,1 --launches 2 > $R/gpurun_out/${TAG}_write.log 2>&1
cat $R/gpurun_out/${TAG}_bench.log | cut -c1-330
