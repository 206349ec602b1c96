#!/bin/bash
# round 6, second session: scheduler 11 (the NVS sampler) lost a quarter between the first session's sweep and the final one -- working tree
# against the library of the session's first commit (scratch_prev/ = git archive f98b685), same lease, alternating; then variants
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
out=gpurun_out/r06/run19_sched11.log
one() { python3 bench.py --no-cpu-baseline --no-streamed --no-cells1024 --no-r64 --steps 6 --allow-variant "$@" 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-36s %-6s %-30s %.2f M TTIs/s  %.3f ms' % (' '.join(sys.argv[1:]) or '(headline)', 'prev' if os.environ.get('RS_HIP_LIB') else 'tree', os.environ.get('RS_JIT_EXTRA', ''),
      d['value'] / 1e6, sum(d['kernel_ms_per_launch']) / len(d['kernel_ms_per_launch'])))" "$@"; }
{
for rep in 1 2; do
  for args in "--sched 11" "--sched 11 --rbgs 64 --rbg-size 8"; do
    one $args
    RS_HIP_LIB=$GRAFT_REPO_ROOT/scratch_prev/radiosaber_amd/libradiosaber_hip.so one $args
    RS_JIT_EXTRA=-DRS_P5_NO_PAIR_SKIP one $args
  done
done
} > $out 2>&1
cat $out
