#!/bin/bash
# round 6, ninth lease: the JIT's rule table was tuned through bench.py, i.e. under the wheel's clang 20; the deployed product (C++ hosts,
# and bench.py from now on) compiles with the system's clang 22.  Same-lease A/B of the rules that matter, under clang 22.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
one() { python3 bench.py --no-cpu-baseline --no-streamed --no-cells1024 --no-r64 --steps 8 --allow-variant "$@" 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %-40s %.2f M TTIs/s  %.3f ms  %s' % (' '.join(sys.argv[1:]) or '(headline)', 'EXTRA=' + os.environ.get('RS_JIT_EXTRA', '') + ' SS=' + os.environ.get('RS_JIT_SCHED_STRATEGY', ''),
      d['value'] / 1e6, sum(d['kernel_ms_per_launch']) / len(d['kernel_ms_per_launch']), d['compiler'].split(' clang ')[1][:9]))" "$@"; }
{
for rep in 1 2; do
  for args in "--sched 8" "--sched 8 --rbgs 64 --rbg-size 8" "--sched 8 --ues-per-slice 50" "" "--rbgs 64 --rbg-size 8" "--ues-per-slice 50" "--sched 101" "--sched 103"; do
    one $args
    RS_JIT_SCHED_STRATEGY=default one $args
    RS_JIT_SCHED_STRATEGY=iterative-ilp one $args
  done
done
} > gpurun_out/r06/run9_rules_clang22.log 2>&1
cat gpurun_out/r06/run9_rules_clang22.log
