#!/bin/bash
# round 6, second session: one RS_JIT_EXTRA variant against the product on one lease (alternating), after the sort parity tests
# usage: run16.sh <tag> <variant flags> [bench args ...;-separated list in $3]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-ab}; var=$2
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
out=gpurun_out/r06/run16_$tag.log
{
cd tools/microbench
for n in 1280; do
  for v in "" "$var"; do
    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DMB_N=$n $v -I. -I../../radiosaber_amd/csrc -I../../include -o /tmp/mb_sort_$n mb_sort.hip 2>/dev/null
    echo "== mb_sort N=$n ${v:-product}"
    timeout 120 /tmp/mb_sort_$n keys_r64.bin | grep "workgroup levels"
  done
done
cd "$GRAFT_REPO_ROOT"
echo "== parity tests"
timeout 900 python3 -m pytest tests/test_sort_killers.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
one() { python3 bench.py --no-cpu-baseline --no-streamed --no-cells1024 --no-r64 --steps 8 --allow-variant "$@" 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-36s %-40s %.2f M TTIs/s  %.3f ms' % (' '.join(sys.argv[1:]) or '(headline)', os.environ.get('RS_JIT_EXTRA', ''),
      d['value'] / 1e6, sum(d['kernel_ms_per_launch']) / len(d['kernel_ms_per_launch'])))" "$@"; }
for rep in 1 2 3; do
  for args in "--rbgs 64 --rbg-size 8" "--rbgs 64 --rbg-size 8 --sched 10" "--config-key exp-fixranues/20slices"; do
    one $args
    RS_JIT_EXTRA="$var" one $args
  done
done
} > $out 2>&1
cat $out
