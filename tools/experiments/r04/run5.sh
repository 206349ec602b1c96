#!/bin/bash
# Round 4, GPU run 5: exact EWMA updates handed to wave 1, drop-in prefetch: full parity suite, then rates and latencies
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run5; mkdir -p $O; cd ..
python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
tools/dropin_latency 2000 > $O/dropin_latency.log 2>&1; cat $O/dropin_latency.log
RS_DROPIN_TIMING=1 tools/dropin_latency 1000 2>&1 | grep -v "^sched" > $O/dropin_latency_timing.log; tail -6 $O/dropin_latency_timing.log
ab() { # tag, extra, bench args
  local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 200 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 --ttis 4000 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-20s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do
ab s7_r25_$rep "" --sched 7
ab s7_r25_base_$rep "-DRS_NO_EARLY17" --sched 7
ab s7_u1000_$rep "" --sched 7 --ues-per-slice 50
ab s7_r64_$rep "" --sched 7 --rbgs 64 --rbg-size 8
ab s1_r25_$rep "" --sched 1
ab s1_u1000_$rep "" --sched 1 --ues-per-slice 50
ab s8_r25_$rep "" --sched 8
ab s8_r25_noearly_$rep "-DRS_HOLD_NO_EARLY" --sched 8
ab s8_u1000_$rep "" --sched 8 --ues-per-slice 50
ab s9_r25_$rep "" --sched 9
done
