#!/bin/bash
# Run ON THE GPU BOX: the scheduler x grid sweep of profiles/r0N_sched_sweep.md (BASELINE.json configs[2]).
# The fraction is bench.py's nominal one: algorithmic_bytes_per_tti(U, R, S, sched) -- per scheduler since round 6 -- x the rate / 8 TB/s.
one() { timeout 150 python bench.py --no-cpu-baseline --no-r64 --no-streamed --no-cells1024 --no-cells1024 --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
assert r['frac'] <= 1.0, 'a nominal fraction above 1: algorithmic_bytes_per_tti is wrong for this scheduler'
print('| %s | %.1f M TTIs/s, %.1f us, %.0f GB/s (%.1f %%) |' % (' '.join(sys.argv[1:]), d['value']/1e6, d['us_per_tti_per_cell'], r['achieved'], 100*r['frac']))" "$@"; }
for s in 1 7 8 9 10 11 101 103; do
  one --sched $s --ues-per-slice 50
  one --sched $s
  one --sched $s --rbgs 64 --rbg-size 8
done
