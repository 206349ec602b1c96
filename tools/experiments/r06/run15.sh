#!/bin/bash
# round 6, second session: working tree against the previous commit's library (scratch_prev/, see run12.sh): parity tests of the
# transport schedulers + drop-in calls, then alternating bench runs on one lease.  usage: run15.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-ab}
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
out=gpurun_out/r06/run15_$tag.log
{
echo "== parity tests"
timeout 900 python3 -m pytest tests/test_sort_killers.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
one() { python3 bench.py --no-cpu-baseline --no-streamed --no-cells1024 --no-r64 --steps 8 --allow-variant "$@" 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-30s %-6s %-40s %.2f M TTIs/s  %.3f ms' % (' '.join(sys.argv[1:]) or '(headline)', 'prev' if os.environ.get('RS_HIP_LIB') else 'tree', os.environ.get('RS_JIT_EXTRA', ''),
      d['value'] / 1e6, sum(d['kernel_ms_per_launch']) / len(d['kernel_ms_per_launch'])))" "$@"; }
for rep in 1 2 3; do
  for args in "" "--rbgs 64 --rbg-size 8" "--sched 8" "--sched 7" "--sched 1"; do
    one $args
    RS_HIP_LIB=$GRAFT_REPO_ROOT/scratch_prev/radiosaber_amd/libradiosaber_hip.so one $args
  done
done
} > $out 2>&1
cat $out
