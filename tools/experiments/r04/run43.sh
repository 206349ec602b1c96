#!/bin/bash
# Round 4, GPU run 43: LLVM's machine scheduler strategy per kernel, re-measured on the lean builds (RS_JIT_SCHED_STRATEGY)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run43; mkdir -p $O; cd ..
ab() { local tag=$1 strat=$2; shift 2
  if [ -n "$strat" ]; then export RS_JIT_SCHED_STRATEGY=$strat; else unset RS_JIT_SCHED_STRATEGY; fi
  timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-20s %-16s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$strat]" || tail -3 $O/ab_$tag.log
}
for st in "" default iterative-ilp max-ilp iterative-minreg max-memory-clause; do
ab s9_r25 "$st" --sched 9 --ttis 8000
ab s9_r64 "$st" --sched 9 --ttis 4000 --rbgs 64 --rbg-size 8
ab s8_r25 "$st" --sched 8 --ttis 4000
ab s7_r25 "$st" --sched 7 --ttis 4000
ab s1_r25 "$st" --sched 1 --ttis 4000
ab s8_u1000 "$st" --sched 8 --ttis 4000 --ues-per-slice 50
done
