#!/bin/bash
# Round 5, GPU run 4: why did the runs with RS_JIT_EXTRA set and --no-r64 of run 3 come out 2 % slower?  A 2 x 2 on one lease.
R=$GRAFT_REPO_ROOT; cd $R
one() { RS_JIT_EXTRA="$1" python bench.py --allow-variant --no-cpu-baseline --no-streamed --steps 5 ${@:2} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('[%s] %s: %.2f M  cells %.1f / %.1f / %.1f ms  %.0f MHz' % (sys.argv[1], ' '.join(sys.argv[2:]), d['value'] / 1e6, d['cell_ms_min'], d['cell_ms_mean'], d['cell_ms_max'], d['shader_mhz']))" "$@"; }
for rep in 1 2; do
one ""
one "" --no-r64
one "-DRS_DUMMY=1"
one "-DRS_DUMMY=1" --no-r64
done
