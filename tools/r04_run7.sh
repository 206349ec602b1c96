#!/bin/bash
# Round 4, GPU run 7: after deleting the losing variants -- full parity suite, drop-in prefetch A/B (same box), rates of the sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run7; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
for rep in 1 2; do
echo "--- prefetch"; tools/dropin_latency 1500 | grep specialised
echo "--- no prefetch"; RS_JIT_EXTRA="-DRS_NO_PREFETCH" tools/dropin_latency 1500 | grep specialised
done 2>&1 | tee $O/dropin_ab.log
ab() { # tag, extra, bench args
  local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 4 --warmup 1 --ttis 4000 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-20s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do
ab s9_r25_$rep "" --sched 9
ab s9_r64_$rep "" --sched 9 --rbgs 64 --rbg-size 8
ab s7_r25_$rep "" --sched 7
ab s7_r25_base_$rep "-DRS_NO_EARLY17" --sched 7
ab s8_r25_$rep "" --sched 8
ab s1_r25_$rep "" --sched 1
ab s103_r25_$rep "" --sched 103
ab s103_r64_$rep "" --sched 103 --rbgs 64 --rbg-size 8
done
