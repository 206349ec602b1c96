#!/bin/bash
# Round 4, closing record (after tools/record_r04.sh + the summaries HERE, so that profiles/traffic.json and inst_counts.json carry the library's hash):
# the GPU suite, the queue model on the lean build of its kernels, and the default bench line (no stale PMC references).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd ..
python -m pytest tests -m gpu -q > $O/r04_gputests.log 2>&1; echo "pytest rc $?" >> $O/r04_gputests.log; tail -2 $O/r04_gputests.log
bash tools/profile_queue_mode.sh r04q > $O/prof_queue.log 2>&1
cd ..
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.log 2>&1; tail -1 $O/r04_smoke.log
python3 bench.py > $O/r04_bench_default.log 2> $O/r04_bench_default.err
cut -c1-200 $O/r04_bench_default.log
