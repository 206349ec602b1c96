#!/bin/bash
# Round 5, GPU run 1: the full GPU suite, the default bench line, the counting sort A/B (round-4 form against the rewritten one, same
# box), the drop-in call's split (polled completion word on / off; kernel time by rocprofv3 --kernel-trace).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run1; mkdir -p $O; cd $R
( time python -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1; tail -3 $O/gputests.log
python bench.py > $O/bench_default.log 2> $O/bench_default.err; cut -c1-300 $O/bench_default.log
for v in "" "-DRS_COUNTING_SORT_V1" "" "-DRS_COUNTING_SORT_V1"; do
  RS_JIT_EXTRA="$v" python bench.py --allow-variant --no-cpu-baseline --no-streamed --steps 6 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('counting sort [%s]: %.2f M at 25 RBGs, r64 %.2f M' % (sys.argv[1], d['value'] / 1e6, d['value_r64'] / 1e6))" "$v" | tee -a $O/counting_ab.log
done
g++ -O2 -std=c++17 -Iinclude tools/dropin_latency.cpp -Lradiosaber_amd -lradiosaber_hip -Wl,-rpath,$R/radiosaber_amd -o /tmp/dropin_latency
for poll in 1 0; do
  echo "== RS_DROPIN_POLL=$poll" >> $O/dropin.log
  RS_DROPIN_POLL=$poll RS_DROPIN_TIMING=1 /tmp/dropin_latency 2000 >> $O/dropin.log 2>&1
done
tail -30 $O/dropin.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dropin_prof -- /tmp/dropin_latency 1000 > $O/dropin_prof.log 2>&1
find $O/dropin_prof -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -20
