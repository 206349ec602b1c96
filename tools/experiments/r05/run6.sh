#!/bin/bash
# Round 5, GPU run 6: the fair issue priority on the short-TTI schedulers at the reference's small shapes: off / period 16 / period 128 / 512.
R=$GRAFT_REPO_ROOT; cd $R
one() { RS_PRIO_BALANCE=$1 RS_JIT_EXTRA="$2" python bench.py --allow-variant --no-cpu-baseline --no-streamed --no-r64 --steps 5 ${@:3} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('balance %s [%s] %s: %.2f M  cells %.2f / %.2f / %.2f ms' % (sys.argv[1], sys.argv[2], ' '.join(sys.argv[3:]), d['value'] / 1e6, d['cell_ms_min'], d['cell_ms_mean'], d['cell_ms_max']))" "$@"; }
for key in exp-fixranues/10slices-ip/config-pf.json exp-fix20slices/10ues-ip/config-pf.json; do
  for s in 7 1 8; do
    one 0 "" --sched $s --config-key $key
    one 2 "" --sched $s --config-key $key
    one 2 "-DRS_PRIO_PERIOD=128" --sched $s --config-key $key
    one 2 "-DRS_PRIO_PERIOD=512" --sched $s --config-key $key
  done
done
for s in 7 1 8; do
  one 0 "" --sched $s
  one 2 "" --sched $s
  one 2 "-DRS_PRIO_PERIOD=128" --sched $s
done
