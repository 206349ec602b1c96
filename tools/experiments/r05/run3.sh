#!/bin/bash
# Round 5, GPU run 3: co-resident cells taking turns at the issue priority (RS_SETPRIO) -- same-lease A/B per scheduler and grid, the
# per-cell run-time spread with and without, window lengths.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run3; mkdir -p $O; cd $R
for pb in 0 1 2; do echo "== RS_PRIO_BALANCE=$pb"; RS_PRIO_BALANCE=$pb python tools/cell_spread.py | head -4; done 2>&1 | tee $O/spread.log
one() { RS_PRIO_BALANCE=$1 RS_JIT_EXTRA="$2" python bench.py --allow-variant --no-cpu-baseline --no-streamed --steps 5 ${@:3} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('balance %s [%s] %s: %.2f M  r64 %s  cells %.1f / %.1f / %.1f ms' % (sys.argv[1], sys.argv[2], ' '.join(sys.argv[3:]), d['value'] / 1e6, ('%.2f M' % (d['value_r64'] / 1e6)) if 'value_r64' in d else '-', d['cell_ms_min'], d['cell_ms_mean'], d['cell_ms_max']))" "$@"; }
for rep in 1 2; do for pb in 0 1 2; do one $pb ""; done; done 2>&1 | tee $O/ab_sched9.log
for w in 4 8 32 64; do one 2 "-DRS_PRIO_PERIOD=$w" --no-r64; done; one 2 "-DRS_PRIO_PERIOD=16" --no-r64; one 1 "-DRS_PRIO_WINDOW_LOG2=15" --no-r64 2>&1 | tee $O/ab_window.log
for s in 8 7 1 10; do for pb in 0 2; do one $pb "" --sched $s; done; done 2>&1 | tee $O/ab_scheds.log
for a in "--ues-per-slice 50" "--cells 384" "--cells 1024"; do for pb in 0 2; do one $pb "" --no-r64 $a; done; done 2>&1 | tee $O/ab_shapes.log
