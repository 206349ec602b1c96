#!/bin/bash
# Run ON THE GPU BOX: the scheduler x grid sweep of profiles/r01_sched_sweep.md (BASELINE.json configs[2]).
one() { timeout 150 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('| %s | %.1f M TTIs/s, %.1f us, %.0f GB/s (%.1f %%) |' % (' '.join(sys.argv[1:]), d['value']/1e6, d['us_per_tti_per_cell'], r['achieved'], 100*r['frac']))" "$@"; }
for s in 1 7 8 9 10 11 101 103; do
  one --sched $s --ues-per-slice 50
  one --sched $s
  one --sched $s --rbgs 64 --rbg-size 8
done
