#!/bin/bash
# round 6, second session: RS_JIT_EXTRA variants against the product on one lease, alternating; no tests (variants that only move work)
# usage: run17.sh <tag> "<bench args>" <variant> [<variant> ...]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; args=$2; shift 2
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
out=gpurun_out/r06/run17_$tag.log
one() { python3 bench.py --no-cpu-baseline --no-streamed --no-cells1024 --no-r64 --steps 8 --allow-variant "$@" 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-36s %-44s %.2f M TTIs/s  %.3f ms' % (' '.join(sys.argv[1:]) or '(headline)', os.environ.get('RS_JIT_EXTRA', ''),
      d['value'] / 1e6, sum(d['kernel_ms_per_launch']) / len(d['kernel_ms_per_launch'])))" "$@"; }
{
for rep in 1 2 3; do
  one $args
  for v in "$@"; do RS_JIT_EXTRA="$v" one $args; done
done
} > $out 2>&1
cat $out
